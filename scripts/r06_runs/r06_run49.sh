#!/bin/bash
# round 6 run 49: the step-level parity block of the bench line with per-tensor values and the hinge-flip count, three default runs
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
for i in 1 2 3; do
  timeout 900 python bench.py --no-cpu-baseline --no-exact-cycle 2>/dev/null | tail -1 > gpurun_out/r06_49_bench_$i.json
  python - <<PY
import json
d=json.load(open('gpurun_out/r06_49_bench_$i.json')); p=d['parity_gan_steps']['product']
print($i, round(d['value'],1), d['parity_ok'], p['ok'], 'd_step', round(p['d_step']['grad_cos_min'],5), round(p['d_step']['grad_rel_l2_max'],4), 'flips', p['d_step']['hinge_flips'], {k:(round(v['cos'],4), round(v['rel_l2'],4)) for k,v in p['per_tensor']['d_step'].items()})
PY
done 2>&1 | tee gpurun_out/r06_49_parity.txt
