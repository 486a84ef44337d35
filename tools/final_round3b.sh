#!/bin/bash
# Run on the GPU box from the repo root: the GPU test tier, the default bench line, the document benches (tools/bench_documents.py)
# and the kernel statistics of the document path.  Usage: tools/final_round3b.sh  ->  gpurun_out/final/*
set -e
root=$(pwd)
out=$root/gpurun_out/final
mkdir -p "$out"
timeout -k 10 900 python -m pytest tests -m gpu -x -q > "$out/gputest.log" 2>&1
tail -n 2 "$out/gputest.log"
VK_BENCH_FULL=$out/bench_default_full.json timeout -k 10 400 python bench.py > "$out/bench_default.json" 2> "$out/bench_default.err"
tail -c 600 "$out/bench_default.json"
timeout -k 10 300 python tools/bench_documents.py > "$out/bench_documents.jsonl" 2> "$out/bench_documents.err"
timeout -k 10 300 python tools/bench_documents.py --oracle-docs 0 --docs 20 --short 400000 --strategies linear,wsb,rwmd >> "$out/bench_documents.jsonl" 2>> "$out/bench_documents.err"
timeout -k 10 300 python tools/bench_documents.py --oracle-docs 0 --docs 8000 --min-len 300 --max-len 512 --strategies linear,wsb >> "$out/bench_documents.jsonl" 2>> "$out/bench_documents.err"
timeout -k 10 300 python tools/bench_documents.py --oracle-docs 0 --docs 20000 --min-len 100 --max-len 200 --strategies linear,wsb >> "$out/bench_documents.jsonl" 2>> "$out/bench_documents.err"
cut -c1-240 "$out/bench_documents.jsonl"
export TMPDIR=/tmp
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$out/stats_documents" -- python3 "$root/tools/bench_documents.py" --oracle-docs 0 --reps 8 > "$out/prof_documents.log" 2>&1
f=$(find "$out/stats_documents" -name "*kernel_stats.csv" | head -n 1)
head -n 14 "$f" > "$out/r03_kernel_stats_documents.csv"
cat "$out/r03_kernel_stats_documents.csv" | cut -c1-200
