#!/bin/bash
# round 6 run 30: non-temporal loads (1) / stores (2) / both (3) in k_affine_act (M355_ELEM_NT) and non-temporal output stores in k_conv_c8
# (M355_C8_NT=1): the bench line's per-kernel times, same box, alternated
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
one() { # label, env
  env $2 timeout 600 python bench.py --no-cpu-baseline --no-step-parity 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); k=d['kernels_ms_per_step']; print('$1', round(d['value'],1), round(d['ms_per_step'],3), round(d['gan_ms_per_cycle'],3), 'affine', round(k['affine_act_fwd'],3), 'c8', round(k['k_conv_c8'],3), 'halo', round(k['k_conv_halo'],3), d.get('parity_ok'))"
}
for rep in 1 2; do
  one base "A=1"
  one nt_load "M355_ELEM_NT=1"
  one nt_store "M355_ELEM_NT=2"
  one nt_both "M355_ELEM_NT=3"
  one c8_nt "M355_C8_NT=1"
done 2>&1 | tee gpurun_out/r06_30_nt_ab.txt
