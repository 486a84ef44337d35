#!/bin/bash
# On the GPU box: queries of 65 .. 512 tokens (vk_longq_kernel) over 1 M x 32 x 300-d (200 k slices for general gaps), per query.
# Usage: tools/bench_long_query.sh  ->  gpurun_out/long_query.log
out=gpurun_out/long_query.log
: > $out
for lt in 65 100 200 512; do
	for gap in linear affine; do
		python tools/bench_configs.py --alg align --gap $gap --len-t $lt --sentences 1000000 --steps 4 --warmup 1 2>/dev/null | tail -n 1 | cut -c1-400 >> $out
	done
	python tools/bench_configs.py --alg align --gap exp5 --len-t $lt --sentences 200000 --steps 2 --warmup 1 2>/dev/null | tail -n 1 | cut -c1-400 >> $out
done
python tools/bench_configs.py --alg align --gap linear --len-t 100 --sentences 1000000 --layout static --steps 4 --warmup 1 2>/dev/null | tail -n 1 | cut -c1-400 >> $out
cat $out
