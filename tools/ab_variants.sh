#!/bin/bash
# on the GPU box: times one bench.py configuration for the regular library and for each variant under vectorian_amd/lib/variants
# usage: tools/ab_variants.sh "<bench args>" name1 name2 ...   (name "base" = the regular build; NAME@ENV=V sets an environment variable)
args=$1; shift
for rnd in 1 2; do
for v in "$@"; do
	name=${v%%@*}; envs=""
	if [ "$name" != "$v" ]; then envs=${v#*@}; fi
	lib=""
	if [ "$name" != "base" ]; then lib="VECTORIAN_HIP_LIB=$(pwd)/vectorian_amd/lib/variants/$name.so"; fi
	env $lib $envs python bench.py $args --no-extra --no-cpu-baseline 2>/dev/null | python -c "import sys,json; o=json.loads(sys.stdin.read()); print('$v', round(o['value']/1e6,1), 'M/s', round(o['ms_per_step'],3), 'ms/step kernel', round(o['roofline']['kernel_ms'],3), 'frac', o['roofline']['frac'])"
done; done
