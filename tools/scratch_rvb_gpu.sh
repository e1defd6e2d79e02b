set -e
timeout -k 10 900 python -m pytest tests/test_gpu_tempering.py -x -q -m gpu 2>&1 | tail -5
timeout -k 10 300 python tools/bench_tempering.py
