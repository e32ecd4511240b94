#!/bin/bash
# developer tool: static instruction mix of the traversal kernels (no GPU needed): compiles pt_wavefront.hip with --save-temps into /tmp/isa_stats and
# prints, per kernel, VALU / v_mov / SALU / memory instruction counts of the whole kernel. usage: tools/isa_stats.sh [extra hipcc flags]
set -e
D=/tmp/isa_stats; rm -rf $D; mkdir -p $D; cd $D
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -fhip-fp32-correctly-rounded-divide-sqrt "$@" --save-temps -c /root/repo/rtxpt_amd/csrc/pt_wavefront.hip -o pt_wavefront.o 2>/dev/null
S=pt_wavefront-hip-amdgcn-amd-amdhsa-gfx950.s
for k in _ZN3ptk8k_extendILb0EEE _ZN3ptk8k_shadowILb0ELb0EEE _ZN3ptk14k_extend_tasksILi0ELb0EEE _ZN3ptk7k_shadeILb0ENS_18PathKernelContextTILb1EEEEE; do
  n=$(grep -n "^$k" $S | head -1 | cut -d: -f1)
  sed -n "${n},\$p" $S | awk '{print} /s_endpgm/{exit}' > $k.s
  echo "$k: lines $(wc -l < $k.s) VALU $(grep -cE '^\s+v_' $k.s) mov $(grep -cE '^\s+v_mov_b(32|64)_e32' $k.s) SALU $(grep -cE '^\s+s_' $k.s) MEM $(grep -cE '^\s+(global_|ds_|buffer_|flat_|scratch_)' $k.s) vgpr $(grep -A30 "^\s*.amdhsa_kernel $k" $S | grep -m1 next_free_vgpr | awk '{print $2}')"
done
