#!/bin/bash
# compact kernel resource usage of one translation unit: name, SGPRs, VGPRs, scratch, occupancy, LDS
# usage: tools/kusage.sh motion_planning_amd/csrc/rollout_pk.hip [extra hipcc flags]
src=$1; shift
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC "$@" -Rpass-analysis=kernel-resource-usage -c "$src" -o /dev/null 2>&1 |
  awk '/Function Name:/ {n=$0; sub(/.*Function Name: /,"",n); sub(/ \[-Rpass.*/,"",n)}
       /TotalSGPRs:/ {s=$(NF-1)} / VGPRs:/ {v=$(NF-1)} /AGPRs:/ {a=$(NF-1)} /ScratchSize/ {sc=$(NF-1)} /Occupancy/ {o=$(NF-1)}
       /LDS Size/ {l=$(NF-1); cmd="echo " n " | c++filt"; cmd | getline d; close(cmd); printf "sgpr %3s vgpr %3s agpr %3s scratch %4s occ %s lds %6s  %s\n", s, v, a, sc, o, l, substr(d,1,150)}'
