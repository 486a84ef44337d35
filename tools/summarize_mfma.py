"""MFMA-busy share of the config-4 GEMM kernel from a rocprofv3 PMC pass (tools/profile_configs.sh):
MFMA busy = sum(SQ_VALU_MFMA_BUSY_CYCLES) / (active cycles x SIMDs) (the MfmaUtil formula rocprofv3 -L lists).
rocprofv3 hands back ONE GRBM_GUI_ACTIVE value per dispatch, the sum over the 8 XCDs (value / duration = 16 GHz):
active cycles = value / 8, which also gives the clock the chip actually ran at (it lowers it under MFMA load).
SQ_VALU_MFMA_BUSY_CYCLES counts 32 per v_mfma_f32_32x32x16_bf16 (MI355X_MICROARCH.md), SQ_INSTS_VALU_MFMA_MOPS_BF16 x 512 = flop.

python tools/summarize_mfma.py gpurun_out/prof_r01/pmc_mfma out.csv
"""
import csv
import glob
import os
import sys

SIMDS = 256 * 4
XCDS = 8


def main():
	src, dst = sys.argv[1], sys.argv[2]
	f = glob.glob(os.path.join(src, "**", "*counter_collection.csv"), recursive=True)[0]
	per = {}
	with open(f) as fh:
		for r in csv.DictReader(fh):
			if "vk_rwmd_batch" not in r.get("Kernel_Name", ""):
				continue
			d = per.setdefault((r["Kernel_Name"], r["Dispatch_Id"]), {})
			name, val = r["Counter_Name"], float(r["Counter_Value"])
			# GRBM_GUI_ACTIVE: one value per dispatch (max over its rows); the SQ counters add up over their rows
			d[name] = d.get(name, 0.0) + val
			d["dur"] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-9
	with open(dst, "w") as out:
		out.write("kernel,dispatch,duration_ms,SQ_VALU_MFMA_BUSY_CYCLES,GRBM_GUI_ACTIVE,SQ_INSTS_VALU_MFMA_MOPS_BF16,clock_GHz,mfma_busy_percent,mfma_flop,issued_PFLOPs\n")
		for (k, disp), d in sorted(per.items(), key=lambda x: int(x[0][1])):
			busy, act, mops = d.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0), d.get("GRBM_GUI_ACTIVE", 0.0), d.get("SQ_INSTS_VALU_MFMA_MOPS_BF16", 0.0)
			cyc = act / XCDS
			util = 100.0 * busy / (cyc * SIMDS) if act else 0.0
			out.write('"%s",%s,%.3f,%.0f,%.0f,%.0f,%.3f,%.2f,%.4g,%.3f\n' % (k, disp, d["dur"] * 1e3, busy, act, mops, cyc / d["dur"] / 1e9, util, mops * 512, mops * 512 / d["dur"] / 1e15))
			print(k[:40], disp, "%.2f ms" % (d["dur"] * 1e3), "clock %.2f GHz" % (cyc / d["dur"] / 1e9), "MFMA busy %.1f %%" % util, "flop %.3g" % (mops * 512))


if __name__ == "__main__":
	main()
