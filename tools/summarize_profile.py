"""Condense the rocprofv3 output of tools/profile_round.sh into the small files kept under profiles/.

python tools/summarize_profile.py gpurun_out/prof_r01 r01
  -> <dir>/summary/{tag}_pmc_score_kernel.csv, traffic.json (the kernel statistics are cut by the shell script)
"""
import csv
import glob
import json
import os
import sys


def find(d, suffix):
	hits = glob.glob(os.path.join(d, "**", "*" + suffix), recursive=True)
	return hits[0] if hits else None


def main():
	src, tag = sys.argv[1], sys.argv[2]
	dst = os.path.join(src, "summary")
	os.makedirs(dst, exist_ok=True)
	traffic = {}
	rows_out = []
	per = {}
	for name, counter in (("pmc_fetch", "FETCH_SIZE"), ("pmc_write", "WRITE_SIZE")):
		f = find(os.path.join(src, name), "counter_collection.csv")
		if not f:
			continue
		with open(f) as fh:
			for r in csv.DictReader(fh):
				if r.get("Counter_Name") != counter or "vk_score_kernel" not in r.get("Kernel_Name", ""):
					continue
				key = (r["Kernel_Name"], counter)
				per.setdefault(key, {}).setdefault(r["Dispatch_Id"], 0.0)
				per[key][r["Dispatch_Id"]] += float(r["Counter_Value"])
	kernel = None
	for (kname, counter), d in per.items():
		vals = list(d.values())
		rows_out.append((kname, counter, len(vals), sum(vals) / len(vals), min(vals), max(vals)))
		kernel = kname
		traffic[counter + "_KiB_avg"] = sum(vals) / len(vals)
		traffic["launches_measured"] = len(vals)
	with open(os.path.join(dst, f"{tag}_pmc_score_kernel.csv"), "w") as out:
		out.write("kernel,counter,launches,avg_KiB,min_KiB,max_KiB\n")
		for r in rows_out:
			out.write('"%s",%s,%d,%.3f,%.3f,%.3f\n' % r)
	if "FETCH_SIZE_KiB_avg" in traffic and "WRITE_SIZE_KiB_avg" in traffic:
		traffic["kernel"] = kernel
		traffic["hbm_bytes_per_launch"] = (2 * traffic["FETCH_SIZE_KiB_avg"] + traffic["WRITE_SIZE_KiB_avg"]) * 1024
		traffic["algorithmic_bytes_per_launch"] = 19200000000
		traffic["source"] = f"rocprofv3 --kernel-trace --pmc FETCH_SIZE and --pmc WRITE_SIZE, separate passes of bench.py --steps 4 --warmup 2 --no-extra, round {tag}"
		traffic["correction"] = ("2 x FETCH_SIZE (gfx950 wide-stream under-count, MI355X_MICROARCH.md HBM section)"
			" + WRITE_SIZE, x 1024")
		with open(os.path.join(dst, "traffic.json"), "w") as out:
			json.dump({"config2:exp5": traffic}, out, indent=1)
	print(json.dumps(traffic))


if __name__ == "__main__":
	main()
