#!/bin/bash
# development aid: a build of the library with K8's per-phase clocks (-DLH_RFA_PROF), next to the product build: bash tools/prof_rfa.sh && python tools/c4_stats.py --lib lariat_amd/_build/liblariat_hip_prof.so
cd "$(dirname "$0")/../lariat_amd/csrc" || exit 1
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -Wno-unused-value -ffp-contract=off -pthread -DLH_RFA_PROF -DLH_K1_TRACE ${LH_PROF_EXTRA} -o ../_build/liblariat_hip_prof.so lariat_hip.hip index_build.cpp ingest.cpp records.cpp bamfile.cpp synth.cpp -lz
