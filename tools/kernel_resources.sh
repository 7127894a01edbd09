#!/bin/bash
# Register / LDS / scratch / occupancy per kernel, from the compiler's own remarks (no GPU needed).
# usage: bash tools/kernel_resources.sh <file-stem> [extra flags]     e.g.  bash tools/kernel_resources.sh blend
cd "$(dirname "$0")/../gscream_amd/csrc" || exit 1
F=$1; shift
case $F in
  blend) X="-munsafe-fp-atomics -mllvm -amdgpu-atomic-optimizer-strategy=None -mllvm -amdgpu-sched-strategy=iterative-ilp";;
  preprocess|gauss_bwd) X="-ffp-contract=off";;
  *) X="";;
esac
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 $X "$@" --cuda-device-only -c $F.hip -o /dev/null \
  -Rpass-analysis=kernel-resource-usage 2>&1 | grep -E "Function Name|VGPRs:|AGPRs|ScratchSize|Occupancy|LDS Size|SGPRs:" |
  sed -E 's/.*remark: [^ ]+ //' | paste - - - - - - - | sed -E 's/\[-Rpass-analysis=kernel-resource-usage\]//g; s/ +/ /g'
