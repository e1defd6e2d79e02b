#!/bin/bash
# dev tool: recompile ONE translation unit ($1, e.g. sweep_cluster) and relink the product library from the object files of the last
# full build (the other units must be unchanged); refreshes the build stamp
set -e
cd "$(dirname "$0")/../isingmontecarlo_amd/csrc"
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -c $1.hip -o build/$1.o
hipcc --offload-arch=gfx950 -shared -fPIC -o ../libisingmc_hip.so build/isingmc_hip.o build/sweep_fast.o build/sweep_cluster.o build/sweep_rvb.o build/sweep_w1.o build/sweep_w4.o build/sweep_w6.o build/sweep_w8.o build/sweep_w16.o -ldl
cd ../.. && python -c "
from isingmontecarlo_amd import _build
open(_build.STAMP,'w').write(_build.source_hash()+'\n')"
