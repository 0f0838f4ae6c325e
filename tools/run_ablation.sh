export BELUGA_MCL_ALLOW_MEASUREMENT_BUILD=1
for v in "" $VARIANTS; do
  if [ -z "$v" ]; then unset BELUGA_MCL_LIB; echo "== product"; else export BELUGA_MCL_LIB=$PWD/build/variants/$v/libbeluga_mcl.so; echo "== $v"; fi
  python tools/exp_lf_fixed.py 2>&1 | grep sigma
done
