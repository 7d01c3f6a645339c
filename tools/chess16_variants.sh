#!/bin/bash
# builds ab/v16_<name>.so for a list of "name:flags" variants of chess16.hip (here, no GPU needed)
R=$(cd $(dirname $0)/.. && pwd)
mkdir -p $R/ab
for spec in "$@"; do
  name=${spec%%:*}; flags=${spec#*:}; [ "$flags" = "$spec" ] && flags=""
  bash $R/tools/build_variant.sh chess16 $R/mrgingham_amd/csrc/chess16.hip $R/ab/v16_$name.so -mllvm -amdgpu-sched-strategy=max-ilp $flags -UMRG_EXPERIMENT 2>&1 | grep -v warning | tail -1
done
