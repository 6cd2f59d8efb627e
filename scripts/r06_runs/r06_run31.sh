#!/bin/bash
# round 6 run 31: non-temporal loads in the streaming elementwise kernels (M355_ELEM_NT bits: 1 affine x, 4 bwd_apply, 8 bwd_reduce, 16 lrelu_bwd,
# 32 chan_stats): bench line per setting, same box, alternated; then the GAN module / elementwise tests on the default
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
one() { # label, env
  env $2 timeout 600 python bench.py --no-cpu-baseline --no-step-parity 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); k=d['kernels_ms_per_step']; g=lambda n: round(k.get(n,0),3); print('$1', round(d['value'],1), round(d['ms_per_step'],3), round(d['gan_ms_per_cycle'],3), 'affine', g('affine_act_fwd'), 'apply', g('affine_act_bwd_apply'), 'reduce', g('affine_act_bwd_partial'), 'lrelu', g('lrelu_bwd'), 'stats', g('bn_stats_partial'), d.get('parity_ok'))"
}
for rep in 1 2; do
  one nt0 "M355_ELEM_NT=0"
  one nt1 "M355_ELEM_NT=1"
  one nt5_apply "M355_ELEM_NT=5"
  one nt9_reduce "M355_ELEM_NT=9"
  one nt13 "M355_ELEM_NT=13"
  one nt61_all "M355_ELEM_NT=61"
done 2>&1 | tee gpurun_out/r06_31_nt_ab.txt
timeout 900 python -m pytest tests/test_gan_elem_gpu.py tests/test_gan_modules.py -m gpu -q -x 2>&1 | tail -2
