#!/bin/bash
# Register / spill / occupancy summary of every kernel of one .hip file (hipcc -Rpass-analysis=kernel-resource-usage).
# usage: tools/kernel_resources.sh mrgingham_amd/csrc/cc.hip [extra hipcc flags]
f=$1; shift
/opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -fPIC -ffp-contract=off "$@" -c "$f" -o /dev/null \
    -Rpass-analysis=kernel-resource-usage 2>&1 | grep "remark:" | sed 's/.*remark: //; s/ \[-Rpass.*//' |
  awk '/Function Name:/ {name=$3} /^ *VGPRs:/ {v=$2} /VGPRs Spill|VGPR Spill/ {sp=$NF} /ScratchSize/ {sc=$NF} /Occupancy/ {occ=$NF} /LDS Size/ {print name, "VGPRs", v, "spill", sp, "scratch", sc, "occupancy", occ, "lds", $(NF)}' | c++filt | cut -c1-220
