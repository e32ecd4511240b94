#!/bin/bash
# developer A/B: runs bench.py against every variant library under gpurun_ab/ (usage: tools/ab.sh [steps])
STEPS=${1:-2}
for lib in gpurun_ab/lib_*.so; do
  MI355PT_LIB=$PWD/$lib python bench.py --steps $STEPS --warmup 1 --no-cpu-baseline $AB_ARGS 2>/dev/null | python -c "
import json,sys
d=json.load(sys.stdin); r=d['roofline']; k=r['kernel_ms_per_step']
print('%-28s %7.1f Mrays/s %7.1f ms  ext %6.1f shade %5.1f shadow %5.1f | nodes %.1f tris %.1f leaves %.1f it/ray %.2f util %.2f sh_nodes %.1f sh_tris %.1f | phases %s cyc/it %.0f leafblk %.2f ev %s' % ('$lib'.split('/')[-1], d['value'], d['ms_per_step'], k['k_extend'], k['k_shade'], k['k_shadow'], r['node_visits_per_ray'], r['tri_tests_per_ray'], r['leaf_visits_per_ray'], r['wave_iterations_per_ray'], r['work_slots_per_quad_iteration'], r['shadow_node_visits_per_ray'], r['shadow_tri_tests_per_ray'], ' '.join('%.2f' % x for x in r['phase_cycle_share']), r['cycles_per_wave_iteration'], r['leaf_block_share'], ' '.join('%s=%.3f' % kv for kv in r['block_runs_per_wave_iteration'].items())))"
done
