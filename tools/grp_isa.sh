#!/bin/bash
# (development) compile lsd_regions.hip with -save-temps and print registers / scratch of the lsd_rg_grp kernels; keeps the ISA of one of them in /tmp/gs
cd /root/repo/cube_slam_amd/csrc || exit 1
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fopenmp -Wall -Wno-unused-function -Wno-unused-result -Wno-unused-value -Wno-unused-variable $EXTRA -c lsd_regions.hip -o lsd_regions.o -save-temps=obj 2>&1 | head -20
S=lsd_regions-hip-amdgcn-amd-amdhsa-gfx950.s
grep -A25 "amdhsa_kernel.*lsd_rg_grp\|amdhsa_kernel.*lsd_rg_lpf" $S | grep "kernel\|next_free_vgpr\|private_segment_fixed"
mkdir -p /tmp/gs
for k in ILi1ELi256; do
  a=$(grep -n "^_ZN12_GLOBAL__N_110lsd_rg_grp$k" $S | cut -d: -f1); b=$(awk -v a=$a 'NR>a && /s_endpgm/{print NR; exit}' $S); sed -n "${a},${b}p" $S > /tmp/gs/grp_$k.s; wc -l /tmp/gs/grp_$k.s
done
rm -f lsd_regions-h* lsd_regions.hip-hip*
