#!/bin/bash
# Regenerates everything under profiles/ for round $R on an MI355X box (run from the repository root, e.g. through
# `gpurun --timeout 1100 -- 'bash tools/collect_profiles.sh'`); raw output goes to gpurun_out/, the summaries that
# profiles/README.md describes are then copied by tools/collect_profiles.py.
set -e -o pipefail
export TMPDIR=/tmp
R=${ROUND:-r03}
O=gpurun_out/profiles_raw
mkdir -p $O
python bench.py > $O/${R}_bench.json
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 bench.py --no-cpu-baseline > $O/${R}_bench_under_rocprof.json
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_fetch -- python3 bench.py --no-cpu-baseline --steps 10 --warmup 2 > /dev/null
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_write -- python3 bench.py --no-cpu-baseline --steps 10 --warmup 2 > /dev/null
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --kernel-trace --output-format csv -d $O/pmc_sq1 -- python3 bench.py --no-cpu-baseline --steps 6 --warmup 2 > /dev/null
rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_BRANCH SQ_WAIT_INST_LDS --kernel-trace --output-format csv -d $O/pmc_sq2 -- python3 bench.py --no-cpu-baseline --steps 6 --warmup 2 > /dev/null
python bench.py --pmj3d 32 --beta 4 --replicas 512 --equilibrate 60 --steps 10 --warmup 2 > $O/${R}_bench_pmj3d32.json
python bench.py --pmj3d 16 --beta 4 --replicas 512 --equilibrate 60 --steps 20 --warmup 3 --no-cpu-baseline > $O/${R}_bench_pmj3d16.json
python bench.py --rvb --steps 4 --warmup 1 --equilibrate 60 > $O/${R}_bench_config2_rvb.json
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_rvb -- python3 bench.py --rvb --steps 4 --warmup 1 --equilibrate 60 --no-cpu-baseline > /dev/null
python tools/rvb_phases.py > $O/${R}_rvb_sweep.txt
python tools/rvb_phases.py --cfg 2048 >> $O/${R}_rvb_sweep.txt  # the fused kernel, for comparison
if [ -f isingmontecarlo_amd/csrc/build/exp/lib_rvbtiming.so ]; then  # diagnostic build (EXP_TU=sweep_rvb tools/experiment_build.py rvbtiming=-DSSE_PHASE_TIMING)
  ISINGMC_HIP_LIB=isingmontecarlo_amd/csrc/build/exp/lib_rvbtiming.so python tools/rvb_phases.py > $O/${R}_rvb_phases_diagnostic_build.txt
fi
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM --kernel-trace --output-format csv -d $O/pmc_rvb -- python3 tools/rvb_phases.py --equilibrate 40 > /dev/null
python tools/bench_tempering.py > $O/${R}_bench_tempering_64x64.json
python tools/bench_tempering.py --window 2.0 1.05 > $O/${R}_bench_tempering_64x64_window.json
python tools/pass_split.py > $O/${R}_pass_split.txt
if [ -f isingmontecarlo_amd/csrc/build/exp/lib_timing.so ]; then  # diagnostic build (EXP_TU=sweep_cluster tools/experiment_build.py timing=-DSSE_PHASE_TIMING)
  ISINGMC_HIP_LIB=isingmontecarlo_amd/csrc/build/exp/lib_timing.so python tools/attribute.py --pass cluster 0 2 16 > $O/${R}_cluster_attribution.txt
fi
echo "raw profiles in $O"
