#!/bin/bash
# On the GPU box from the repo root: SQ_INSTS_VALU of the dominant kernel of every VALU-issue-bound bench configuration (the static
# layout, whole documents, the 40-token query), one rocprofv3 pass each (--kernel-trace + --pmc only), and the kernel statistics of
# the same commands.  Writes gpurun_out/pmc_valu_<tag>/valu.json (copy to profiles/valu.json: bench.py quotes it as the achieved
# VALU rate of those entries) and <tag>_kernel_stats_<config>.csv.   Usage: tools/pmc_valu.sh r04 [configs...]
set -e
tag=${1:-r04}
shift || true
configs=${@:-"2static 2q40 4static docs docslin"}
root=$(pwd)
out=$root/gpurun_out/pmc_valu_$tag
mkdir -p "$out"
export TMPDIR=/tmp
cd /tmp
for c in $configs; do
	VK_BENCH_FULL="$out/full_$c.json" rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES --output-format csv -d "$out/pmc_$c" -- python3 "$root/bench.py" --config $c --steps 4 --warmup 2 --no-extra --no-cpu-baseline > "$out/pmc_$c.log" 2>&1
	rocprofv3 --kernel-trace --stats --output-format csv -d "$out/stats_$c" -- python3 "$root/bench.py" --config $c --steps 12 --warmup 4 --no-extra --no-cpu-baseline > "$out/bench_$c.log" 2>&1
	f=$(find "$out/stats_$c" -name "*kernel_stats.csv" | head -n 1)
	head -n 8 "$f" > "$out/${tag}_kernel_stats_config$c.csv"
	tail -n 1 "$out/bench_$c.log" | cut -c1-400
done
cd "$root"
python3 - "$out" "$tag" $configs <<'PY'
import csv, glob, json, os, sys, collections
out, tag, configs = sys.argv[1], sys.argv[2], sys.argv[3:]
pattern = {"4static": "vk_rwmd_static32_kernel", "2static": "vk_score_kernel<2", "2q40": "vk_score32_kernel", "docs": "vk_doc_kernel<false", "docslin": "vk_doc_kernel<false"}
res = {}
for c in configs:
	full = json.load(open(os.path.join(out, f"full_{c}.json")))
	name = full["config"]["name"]
	per = collections.defaultdict(lambda: collections.defaultdict(float))
	for f in glob.glob(os.path.join(out, f"pmc_{c}", "**", "*counter_collection.csv"), recursive=True):
		for r in csv.DictReader(open(f)):
			if pattern[c] in r["Kernel_Name"]:
				per[(r["Kernel_Name"], r["Dispatch_Id"])][r["Counter_Name"]] += float(r["Counter_Value"])
	if not per:
		continue
	# the dominant kernel: the one with the most VALU instructions per launch among the matches
	by_kernel = collections.defaultdict(list)
	for (k, d), cs in per.items():
		by_kernel[k].append(cs)
	k = max(by_kernel, key=lambda k: sum(cs["SQ_INSTS_VALU"] for cs in by_kernel[k]) / len(by_kernel[k]))
	n = len(by_kernel[k])
	avg = {cn: sum(cs[cn] for cs in by_kernel[k]) / n for cn in by_kernel[k][0]}
	res[name] = {"kernel": k[:90], "launches_measured": n, "valu_insts_per_launch": avg["SQ_INSTS_VALU"], "salu_insts_per_launch": avg.get("SQ_INSTS_SALU"),
		"wave_quad_cycles_per_launch": avg.get("SQ_WAVE_CYCLES"), "busy_cycles_per_launch": avg.get("SQ_BUSY_CYCLES"),
		"tokens_per_launch": full.get("tokens_per_gpu"),
		"source": f"rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU .. -- bench.py --config {c} --steps 4 --warmup 2 --no-extra, round {tag}"}
json.dump(res, open(os.path.join(out, "valu.json"), "w"), indent=1)
print(json.dumps(res, indent=1))
PY
