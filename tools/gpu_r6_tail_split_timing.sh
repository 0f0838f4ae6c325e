mkdir -p gpurun_out/r06
O=gpurun_out/r06/lf_tail_split_timing.txt
: > $O
for round in 1 2; do
for lib in product ts384 ts768 ts1536; do
  echo "== $lib" >> $O
  if [ $lib = product ]; then timeout 300 python tools/exp_lf_fixed.py 2>&1 | grep "^sigma" >> $O
  else BELUGA_MCL_ALLOW_MEASUREMENT_BUILD=1 BELUGA_MCL_LIB=build/variants/$lib/libbeluga_mcl.so timeout 300 python tools/exp_lf_fixed.py 2>&1 | grep "^sigma" >> $O; fi
done; done
cat $O
