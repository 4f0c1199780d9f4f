#!/bin/bash
# bench.py's headline + G1/G2 lines for the in-tree library and for every variants/*.so (kernel-tuning experiments)
for l in "" variants/*.so; do
  echo "== ${l:-in-tree}"
  GPBC_LIB_PATH=${l:+$PWD/$l} python bench.py --steps 2 --warmup 1 --no-configs --no-cpu 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); s=d['secondary']; k=d['roofline']['kernels']
print('pairings/s %.3fM  fexp %.2f acc %.2f lines %.2f | g1 %.2fM g2 %.2fM fb %.1fM msm %.1fM h2c1 %.1fM gtexp %.2fM' % (d['value']/1e6, k['k_final_exp']['ms_per_step'], k['k_miller_accumulate']['ms_per_step'], k['k_miller_lines']['ms_per_step'], s['g1_scalar_mults_per_s']/1e6, s['g2_scalar_mults_per_s']/1e6, s['g1_fixed_base_mults_per_s']/1e6, s['g1_msm256_terms_per_s']/1e6, s['g1_map_to_curve_per_s']/1e6, s['gt_exp_per_s']/1e6))"
done
