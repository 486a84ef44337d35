#!/bin/bash
# Run on the GPU box from the repo root: rocprofv3 kernel statistics of every bench.py configuration (one process per
# configuration, so that each kernel_stats.csv holds one workload), the two HBM counter passes of every HBM-bound workload and
# the MFMA-busy pass of config 4.  Counters are collected in passes of their own (--kernel-trace + --pmc only).
# Usage: tools/profile_round.sh r02 [configs...]  ->  gpurun_out/prof_r02/summary/{r02_kernel_stats_<config>.csv, traffic.json, ...}
set -e
tag=${1:-r02}
shift || true
configs=${@:-"2 3 4 5 5wrd 5rwmd 2f32"}
root=$(pwd)
out=$root/gpurun_out/prof_$tag
mkdir -p "$out/summary"
export TMPDIR=/tmp
cd /tmp
for c in $configs; do
	rocprofv3 --kernel-trace --stats --output-format csv -d "$out/stats_$c" -- python3 "$root/bench.py" --config $c --steps 12 --warmup 4 --no-extra --no-cpu-baseline > "$out/bench_$c.log" 2>&1
	tail -n 1 "$out/bench_$c.log" | cut -c1-600
	f=$(find "$out/stats_$c" -name "*kernel_stats.csv" | head -n 1)
	head -n 8 "$f" > "$out/summary/${tag}_kernel_stats_config$c.csv"
	grep "^{\"metric\"" "$out/bench_$c.log" | tail -n 1 > "$out/summary/${tag}_bench_config$c.json"
done
# HBM traffic of the scoring kernel, every HBM-bound configuration: FETCH_SIZE and WRITE_SIZE in passes of their own
for c in $configs; do
	case $c in 4|2static|5rwmd) continue;; esac
	rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d "$out/pmc_fetch_$c" -- python3 "$root/bench.py" --config $c --steps 4 --warmup 2 --no-extra --no-cpu-baseline > "$out/pmc_fetch_$c.log" 2>&1
	rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d "$out/pmc_write_$c" -- python3 "$root/bench.py" --config $c --steps 4 --warmup 2 --no-extra --no-cpu-baseline > "$out/pmc_write_$c.log" 2>&1
done
if echo " $configs " | grep -q " 4 "; then
	rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_BF16 --output-format csv -d "$out/pmc_mfma" -- python3 "$root/bench.py" --config 4 --steps 3 --warmup 2 --no-extra --no-cpu-baseline > "$out/pmc_mfma.log" 2>&1
	python3 "$root/tools/summarize_mfma.py" "$out/pmc_mfma" "$out/summary/${tag}_pmc_config4_mfma.csv"
fi
cd "$root"
python3 tools/summarize_profile.py "$out" "$tag"
