#!/bin/bash
# SQ counter passes over bench.py for the timed kernels (dev tool; raw output under gpurun_out/$1)
set -e -o pipefail
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/${1:-pmc_r3}
mkdir -p $O
cd /tmp
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --kernel-trace --output-format csv -d $O/sq1 -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --steps 6 --warmup 2 > /dev/null
rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_BRANCH SQ_WAIT_INST_LDS --kernel-trace --output-format csv -d $O/sq2 -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --steps 6 --warmup 2 > /dev/null
cd $GRAFT_REPO_ROOT
python3 tools/pmc_summary.py $O
