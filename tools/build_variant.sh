#!/bin/bash
# Builds a measurement variant of libbeluga_mcl.so with extra compiler flags into build/variants/<name>/ (never the product
# library): tools/build_variant.sh <name> "<flags>" [r04 | <git revision>];  use with BELUGA_MCL_LIB=build/variants/<name>/libbeluga_mcl.so
#
# Third argument: a git revision whose sources the variant is built from (A/B against an earlier tree on the same box).
# "r04": the variant is built from ROUND 4's sources (git commit 83dc34d - kernels.hip, context.hip, kernels.h with the
# ablation / timing / statistics blocks the measurement scripts of rounds 2 - 4 drive: -DMCL_ABLATE, -DMCL_LF_TIMING, -DMCL_PIPE_ABLATE,
# -DMCL_DRAW_ABLATE, -DMCL_BEAM_STATS, -DMCL_BEAM_ABLATE, -DMCL_PATCH_LDS_PAD, k_reweight_lf_pipe, the LDS-direct lf_producer = 0 form).
# Round 5 took those blocks out of the product's kernels.hip; the numbers in profiles/r0[2-4]_* were measured on that tree.
# Every variant is tagged -DMCL_MEASUREMENT_BUILD: beluga_amd/capi.py refuses to load it as the product library.
set -eu
name=$1; flags=${2:-}; tree=${3:-}
root=$(cd "$(dirname "$0")/.." && pwd)
out=$root/build/variants/$name
mkdir -p $out
src=$root/beluga_amd/csrc
if [ -n "$tree" ]; then
  rev=$tree; [ "$tree" = "r04" ] && rev=83dc34d
  src=$out/src_$tree
  mkdir -p $src
  for f in kernels.hip context.hip kernels.h se2.h rng.h map_build.h map_build.cpp; do git -C $root show $rev:beluga_amd/csrc/$f > $src/$f; done
  for f in beam_kernels.hip device_common.hpp; do git -C $root show $rev:beluga_amd/csrc/$f > $src/$f 2>/dev/null || rm -f $src/$f; done
  git -C $root show $rev:include/beluga_mcl.h > $src/beluga_mcl.h
fi
common="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Wno-unused-result -I$src -I$root/include -DMCL_MEASUREMENT_BUILD $flags"
/opt/rocm/bin/hipcc $common -x hip -c $src/kernels.hip -o $out/kernels.o
beam=""   # (the beam model's kernels are a translation unit of their own from round 5 on; older revisions have them inside kernels.hip)
if [ -f $src/beam_kernels.hip ]; then /opt/rocm/bin/hipcc $common -x hip -c $src/beam_kernels.hip -o $out/beam_kernels.o; beam=$out/beam_kernels.o; fi
if [ -n "$tree" ]; then
  /opt/rocm/bin/hipcc $common -x hip -c $src/context.hip -o $out/context.o
  /opt/rocm/bin/hipcc $common -c $src/map_build.cpp -o $out/map_build.o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $out/libbeluga_mcl.so $out/kernels.o $beam $out/context.o $out/map_build.o
else
  [ -f $root/beluga_amd/lib/context.o ] || python -m beluga_amd.build
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $out/libbeluga_mcl.so $out/kernels.o $beam $root/beluga_amd/lib/context.o $root/beluga_amd/lib/map_build.o
fi
echo built $out/libbeluga_mcl.so
