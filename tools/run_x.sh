python tools/exp_patch_debug.py 2>&1 | tail -3
BELUGA_MCL_LF_PRODUCER=1 python tools/exp_patch_debug.py 2>&1 | tail -3
python tools/exp_lf_fixed.py 2>&1 | grep sigma
BELUGA_MCL_LF_PRODUCER=1 python tools/exp_lf_fixed.py 2>&1 | grep sigma
ONLY="all (defaults)" python tools/exp_lf_ab.py 25 2>&1 | grep -v "^   "
BELUGA_MCL_LF_PRODUCER=1 ONLY="all (defaults)" python tools/exp_lf_ab.py 25 2>&1 | grep -v "^   "
