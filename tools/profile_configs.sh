#!/bin/bash
# rocprofv3 kernel statistics of the other BASELINE shapes (run on the GPU box from the repo root):
# tools/profile_configs.sh r01 -> gpurun_out/prof_r01/summary/r01_kernel_stats_{config4,wrd,span,static,d768}.csv
set -e
tag=${1:-r01}
root=$(pwd)
out=$root/gpurun_out/prof_$tag
mkdir -p "$out/summary"
export TMPDIR=/tmp
cd /tmp
run() {
	name=$1; shift
	rocprofv3 --kernel-trace --stats --output-format csv -d "$out/stats_$name" -- python3 "$root/tools/bench_configs.py" "$@" > "$out/bench_$name.log" 2>&1
	tail -n 1 "$out/bench_$name.log" | cut -c1-400
	f=$(find "$out/stats_$name" -name "*kernel_stats.csv" | head -n 1)
	head -n 8 "$f" > "$out/summary/${tag}_kernel_stats_$name.csv"
}
run config4 --alg rwmd --batch 256 --steps 4 --warmup 1
# MFMA-busy counters of the same kernel (their own pass: no --stats, no other trace domain)
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_BF16 --output-format csv -d "$out/pmc_mfma" -- python3 "$root/tools/bench_configs.py" --alg rwmd --batch 256 --steps 2 --warmup 1 > "$out/pmc_mfma.log" 2>&1
python3 "$root/tools/summarize_mfma.py" "$out/pmc_mfma" "$out/summary/${tag}_pmc_config4_mfma.csv"
run wrd --alg wrd --steps 6 --warmup 1
run d768 --d 768 --min-len 8 --max-len 64 --sentences 400000 --steps 10 --warmup 2
run span --min-len 1 --max-len 1 --len-t 1 --d 768 --sentences 8000000 --gap linear --steps 10 --warmup 2
run static --layout static --sentences 4000000 --steps 10 --warmup 2
run q20 --len-t 20 --gap linear --steps 10 --warmup 2
run q20wsb --len-t 20 --gap exp5 --steps 10 --warmup 2
