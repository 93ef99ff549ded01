#!/bin/bash
# per-kernel VGPRs / scratch / occupancy / LDS of the gfx950 code object (device-only compile; no GPU needed): bash tools/kernel_resources.sh [filter]
cd "$(dirname "$0")/../lariat_amd/csrc" || exit 1
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-value -ffp-contract=off --cuda-device-only -c -Rpass-analysis=kernel-resource-usage -o /dev/null lariat_hip.hip 2>&1 |
  sed 's/\[-Rpass-analysis=kernel-resource-usage\]//g' |
  awk '/Function Name:/ {name=$NF} / VGPRs:/ {v=$NF} /ScratchSize/ {s=$NF} /Occupancy/ {o=$NF} /LDS Size/ {l=$NF; printf "vgpr %-4s scratch %-5s waves/SIMD %-3s lds %-7s %s\n", v, s, o, l, name}' |
  { if [ -n "$1" ]; then grep -E "$1"; else cat; fi; }
