#!/bin/bash
# On the GPU box from the repo root: instruction-issue counters of the config-4 GEMM kernel and of the four-block general-gap
# kernel (40-token query), each in passes of its own (--kernel-trace + --pmc only).  Usage: tools/pmc_issue.sh r03
set -e
tag=${1:-r03}
root=$(pwd)
out=$root/gpurun_out/pmc_issue_$tag
mkdir -p "$out"
export TMPDIR=/tmp
cd /tmp
for set in "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_MISC SQ_WAVE_CYCLES" "SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES"; do
	name=$(echo $set | tr ' ' '_' | cut -c1-40)
	rocprofv3 --kernel-trace --pmc $set --output-format csv -d "$out/c4_$name" -- python3 "$root/bench.py" --config 4 --steps 3 --warmup 2 --no-extra --no-cpu-baseline > "$out/c4_$name.log" 2>&1
	rocprofv3 --kernel-trace --pmc $set --output-format csv -d "$out/q40_$name" -- python3 "$root/tools/bench_configs.py" --alg align --gap exp5 --len-t 40 --sentences 1000000 --steps 3 --warmup 2 > "$out/q40_$name.log" 2>&1
done
cd "$root"
python3 - "$out" "$tag" <<'PY'
import csv, glob, os, sys, collections
out, tag = sys.argv[1], sys.argv[2]
rows = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(os.path.join(out, "**", "*counter_collection.csv"), recursive=True):
	which = os.path.relpath(f, out).split("_")[0]
	per = collections.defaultdict(float)
	for r in csv.DictReader(open(f)):
		k = r["Kernel_Name"]
		if ("vk_rwmd_batch32d" in k and which == "c4") or ("vk_score32_kernel" in k and which == "q40"):
			per[(k, r["Counter_Name"], r["Dispatch_Id"])] += float(r["Counter_Value"])
	for (k, cn, d), v in per.items():
		rows[(which, k)][cn].append(v)
with open(os.path.join(out, f"{tag}_pmc_issue.csv"), "w") as fh:
	fh.write("workload,kernel,counter,launches,avg_per_launch\n")
	for (which, k), cs in sorted(rows.items()):
		for cn, vals in sorted(cs.items()):
			fh.write('%s,"%s",%s,%d,%.4g\n' % (which, k[:60], cn, len(vals), sum(vals) / len(vals)))
print(open(os.path.join(out, f"{tag}_pmc_issue.csv")).read())
PY
