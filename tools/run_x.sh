#!/bin/bash
set -u
for v in pt2 pt4 "" prev pt2 pt4 "" prev; do
  if [ -z "$v" ]; then unset BELUGA_MCL_LIB; echo "== product"; else export BELUGA_MCL_LIB=$PWD/build/variants/$v/libbeluga_mcl.so; echo "== $v"; fi
  python tools/exp_lf_fixed.py 2>&1 | grep sigma | head -n 1
done
