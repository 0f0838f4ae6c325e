#!/bin/bash
# Builds a measurement variant of libbeluga_mcl.so with extra compiler flags into build/variants/<name>/ (never the product
# library): tools/build_variant.sh <name> "<flags>";  use with BELUGA_MCL_LIB=build/variants/<name>/libbeluga_mcl.so
set -eu
name=$1; flags=${2:-}
root=$(cd "$(dirname "$0")/.." && pwd)
out=$root/build/variants/$name
mkdir -p $out
common="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Wno-unused-result -I$root/include -I$root/beluga_amd/csrc $flags"
/opt/rocm/bin/hipcc $common -x hip -c $root/beluga_amd/csrc/kernels.hip -o $out/kernels.o
[ -f $root/beluga_amd/lib/context.o ] || python -m beluga_amd.build
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $out/libbeluga_mcl.so $out/kernels.o $root/beluga_amd/lib/context.o $root/beluga_amd/lib/map_build.o
echo built $out/libbeluga_mcl.so
