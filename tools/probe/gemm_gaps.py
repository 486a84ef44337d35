"""probe: gaps between consecutive launches of one kernel in a rocprofv3 --kernel-trace csv, and what ran inside them
  python tools/probe/gemm_gaps.py <dir with *_kernel_trace.csv> vk_rwmd_batch32d"""
import csv, glob, os, sys
d, name = sys.argv[1], sys.argv[2]
f = glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True)[0]
rows = [r for r in csv.DictReader(open(f))]
ks = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in rows), key=lambda x: x[0])
g = [k for k in ks if name in k[2]]
for a, b in zip(g[-8:-1], g[-7:]):
	inside = [k for k in ks if k[0] >= a[1] and k[1] <= b[0]]
	print("gemm %.3f ms, gap to next %.3f ms; kernels inside the gap: %s" % ((a[1] - a[0]) / 1e6, (b[0] - a[1]) / 1e6,
		", ".join("%s %.3f" % (k[2][:28], (k[1] - k[0]) / 1e6) for k in inside)))
