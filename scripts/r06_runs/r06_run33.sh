#!/bin/bash
# round 6 run 33: do the HBM-bound elementwise kernels find the tail of what their producer just wrote / their predecessor just read in the
# memory-side cache if they walk the samples in REVERSE?  builds -DM355_ELEM_REV=1 (k_affine_act), 2 (k_act_bwd_apply), 3 (both)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
one() { # label, env
  env $2 timeout 600 python bench.py --no-cpu-baseline --no-step-parity 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); k=d['kernels_ms_per_step']; g=lambda n: round(k.get(n,0),3); print('$1', round(d['value'],1), round(d['ms_per_step'],3), round(d['gan_ms_per_cycle'],3), 'affine', g('affine_act_fwd'), 'apply', g('affine_act_bwd_apply'), 'reduce', g('affine_act_bwd_partial'), d.get('parity_ok'))"
}
for rep in 1 2; do
  one base "M355_LIB=libm355.so"
  one rev_affine "M355_LIB=libm355_rev1.so"
  one rev_apply "M355_LIB=libm355_rev2.so"
  one rev_both "M355_LIB=libm355_rev3.so"
done 2>&1 | tee gpurun_out/r06_33_rev.txt
