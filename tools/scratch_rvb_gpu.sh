set -e
timeout -k 10 1000 python -m pytest tests -x -q -m gpu 2>&1 | tail -5
timeout -k 10 120 python bench.py --rvb --steps 4 --warmup 1 --equilibrate 60 --no-cpu-baseline
