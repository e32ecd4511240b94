#!/bin/bash
# developer tool (no GPU needed): VGPR / SGPR / scratch / LDS of every kernel in pt_wavefront.hip as compiled for gfx950. usage: tools/regs.sh [extra hipcc flags]
set -e
D=/tmp/regs_$$; rm -rf $D; mkdir -p $D; cd $D
SRC=${REGS_SRC:-pt_wavefront}
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -fhip-fp32-correctly-rounded-divide-sqrt "$@" --save-temps -c /root/repo/rtxpt_amd/csrc/$SRC.hip -o $SRC.o 2>/dev/null
S=$SRC-hip-amdgcn-amd-amdhsa-gfx950.s
awk '/^ *- \.agpr_count:/{a=$3} /^ *\.name:/{k=$2} /^ *\.private_segment_fixed_size:/{p=$2} /^ *\.group_segment_fixed_size:/{l=$2} /^ *\.sgpr_count:/{s=$2} /^ *\.vgpr_count:/{v=$2} /^ *\.vgpr_spill_count:/{printf "%-110s vgpr %4s sgpr %4s scratch %6s lds %6s spilled %s\n", substr(k,1,110), v, s, p, l, $2}' $S | grep -E "${REGS_FILTER:-k_shade|k_extend|k_shadow|k_nee|k_surface|k_tail}"
rm -rf $D
