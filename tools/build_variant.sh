#!/bin/bash
# A/B builds of one translation unit: tools/build_variant.sh NAME UNIT "-DFLAG=1 ..."  ->  vectorian_amd/lib/variants/NAME.so
# (the other objects are those of the regular build; select the library with VECTORIAN_HIP_LIB=... on the GPU box)
set -e
name=$1; unit=$2; flags=$3
root=$(cd "$(dirname "$0")/.." && pwd)
cs=$root/vectorian_amd/csrc
obj=$root/vectorian_amd/lib/obj
mkdir -p "$root/vectorian_amd/lib/variants"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -Wall -Wno-pass-failed -Wno-unused-function $flags -c -o "$obj/variant_${name}.o" "$cs/$unit.hip"
others=$(ls $obj/vk_*.o | grep -v "/$unit.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o "$root/vectorian_amd/lib/variants/$name.so" "$obj/variant_${name}.o" $others
echo "built variants/$name.so"
