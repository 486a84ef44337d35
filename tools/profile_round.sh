#!/bin/bash
# Run on the GPU box from the repo root: rocprofv3 kernel statistics and the two HBM counter passes for bench.py.
# Usage: tools/profile_round.sh r01   ->  gpurun_out/prof_r01/{stats_exp5,stats_linear,pmc_fetch,pmc_write}
set -e
tag=${1:-r01}
root=$(pwd)
out=$root/gpurun_out/prof_$tag
mkdir -p "$out"
export TMPDIR=/tmp
cd /tmp
for gap in exp5 linear; do
	rocprofv3 --kernel-trace --stats --output-format csv -d "$out/stats_$gap" -- python3 "$root/bench.py" --steps 20 --warmup 3 --gap $gap --no-cpu-baseline > "$out/bench_$gap.log" 2>&1
	tail -n 1 "$out/bench_$gap.log"
done
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d "$out/pmc_fetch" -- python3 "$root/bench.py" --steps 4 --warmup 1 --gap exp5 --no-cpu-baseline > "$out/pmc_fetch.log" 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d "$out/pmc_write" -- python3 "$root/bench.py" --steps 4 --warmup 1 --gap exp5 --no-cpu-baseline > "$out/pmc_write.log" 2>&1
cd "$root"
python3 tools/summarize_profile.py "$out" "$tag"
