#!/bin/bash
# developer tool (no GPU needed): VGPR / SGPR / scratch / LDS of every kernel in pt_wavefront.hip as compiled for gfx950. usage: tools/regs.sh [extra hipcc flags]
set -e
D=/tmp/regs_$$; rm -rf $D; mkdir -p $D; cd $D
SRC=${REGS_SRC:-pt_wavefront}
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -fhip-fp32-correctly-rounded-divide-sqrt "$@" --save-temps -c /root/repo/rtxpt_amd/csrc/$SRC.hip -o $SRC.o 2>/dev/null
S=$SRC-hip-amdgcn-amd-amdhsa-gfx950.s
awk '/^\s*\.amdhsa_kernel /{k=$2} /amdhsa_next_free_vgpr/{v=$2} /amdhsa_next_free_sgpr/{s=$2} /amdhsa_private_segment_fixed_size/{p=$2} /amdhsa_group_segment_fixed_size/{l=$2} /^\s*\.end_amdhsa_kernel/{printf "%-110s vgpr %4s sgpr %4s scratch %6s lds %6s\n", substr(k,1,110), v, s, p, l}' $S | grep -E "${REGS_FILTER:-k_shade|k_extend|k_shadow|k_nee|k_surface|k_tail}"
rm -rf $D
