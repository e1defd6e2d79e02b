set -e
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "rvb" 2>&1 | tail -3
for w in 4 8; do ISINGMC_RVB_MAIN_W=$w timeout -k 10 120 python tools/rvb_phases.py; done
ISINGMC_HIP_LIB=isingmontecarlo_amd/csrc/build/exp/lib_rvbtiming.so timeout -k 10 120 python tools/rvb_phases.py
