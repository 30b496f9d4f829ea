#!/bin/bash
# usage: tools/kres.sh "<extra flags>" [kernel-substring]   -- VGPR / spill / LDS / occupancy of the device kernels (no GPU needed)
flags="$1"; pat="${2:-fft512b_kernelILi13ELi0}"
hipcc --offload-arch=gfx950 -O3 -std=c++17 -DHIPFEAT_BUILD $flags -Iinclude --cuda-device-only -c -Rpass-analysis=kernel-resource-usage \
  lhotse_amd/csrc/hipfeat.hip -o /dev/null 2>&1 | grep -A12 "Function Name: .*$pat" | grep -E "Function Name|VGPRs:|Spill|Occupancy|LDS Size|SGPRs:" | sed 's/.*remark: //' | tr '\n' ' '; echo
