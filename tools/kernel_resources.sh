#!/bin/bash
# Register / LDS / spill table of every k_eval_z instance (cross-compiles, no GPU needed):  tools/kernel_resources.sh [regex]
mkdir -p /tmp/kres && cd /tmp/kres && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=on -mllvm -simplifycfg-sink-common=false -I/root/repo/include -I/root/repo/super-resolution_amd/csrc -c /root/repo/super-resolution_amd/csrc/kernels_ztile.hip -save-temps -o kz.o 2>/dev/null
S=kernels_ztile-hip-amdgcn-amd-amdhsa-gfx950.s
echo "instance<T,S,B,REGK,R>  lds sgpr sgpr_spill vgpr vgpr_spill"
grep -E "^\s+\.(vgpr_count|sgpr_count|group_segment_fixed_size|vgpr_spill_count|sgpr_spill_count):|^\s+\.name:" $S | paste - - - - - - | awk '{print $4, $2, $6, $8, $10, $12}' | grep k_eval_z | sed 's/_ZN5srmap12_GLOBAL__N_18k_eval_z//; s/EEEvNS0_5ZArgs[^ ]*//; s/ELi/,/g; s/^I\(.\)Li/\1,/' | grep "${1:-.}"
