#!/bin/bash
# Regenerates every fixture that holds output of the REFERENCE'S OWN TEXT (build container only: needs /root/reference; about two hours of CPU time on eight cores, most of it the
# light baker's text run thread by thread at 4K). The small fixtures come first; the at-scale ones (DESIGN.md §5 "Full-size frames against the reference's own text") follow.
# usage: tools/regen_reference_text_fixtures.sh [small|scale]
set -e
cd "$(dirname "$0")/.."
G=tests/golden
if [ "${1:-all}" != "scale" ]; then
  for s in make_refpin_golden make_refpin_hlsl_golden make_reference_integrator_golden make_device_path_golden make_env_cube_golden make_neeat_golden make_neeat_loop_golden \
           make_stable_planes_golden make_realtime_golden; do echo "== $s"; python $G/$s.py; done
fi
if [ "${1:-all}" != "small" ]; then
  for s in make_bench_frame_golden make_config_frames_golden make_pin_cases_hd_golden make_fuzz_hd_golden make_stable_planes_hd_golden make_fuzz_sp_hd_golden make_env_cube_2048_golden \
           make_neeat_4k_golden make_realtime_4k_golden make_neeat_loop_hd_golden make_realtime_hd_golden make_realtime_coupled_4k_golden make_neeat_loop_4k_golden make_display_4k_golden; do
    echo "== $s"; python $G/$s.py
  done
fi
