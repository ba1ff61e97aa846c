#!/bin/bash
# Registers / spills / scratch / occupancy of the kernels of one source file, as the compiler reports them:
#   tools/kernel_resources.sh msda_backward_onepass.hip [extra -D flags] | grep <kernel>
cd "$(dirname "$0")/../mvdetr_amd/csrc" || exit 1
f=$1; shift
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -munsafe-fp-atomics -ffp-contract=fast "$@" \
    -Rpass-analysis=kernel-resource-usage -c "$f" -o /dev/null 2>&1 | grep "remark:" |
  sed 's/.*remark: *//; s/ *\[-Rpass-analysis=kernel-resource-usage\]//' |
  awk '/^Function Name/ {if (line) print line; line=$3; next} /^(VGPRs|ScratchSize|Occupancy|SGPRs Spill|VGPRs Spill)/ {line=line " | " $0} END {print line}' |
  while read -r l; do n=${l%% *}; echo "$(echo "$n" | c++filt | cut -c1-110) ${l#* }"; done
