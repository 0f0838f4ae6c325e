python tools/exp_beam_table_debug.py 2>&1 | head -2
timeout 600 python -m pytest tests/test_gpu_parity.py -q -x -k "beam" 2>&1 | tail -3
python tools/bench_configs.py 5 --steps 3 2>&1 | tail -2
