#!/bin/bash
# round 6 run 47: the spectral-norm prefetch by SITE (M355_SN_PREFETCH_SITES: 1 = D's chain under the G step's backward, 2 = G's chain under the D
# step, 4 = D's chain after its optimiser step; 7 = all, the default; 0 = none) at batch 64 (eager) and 16 (graph)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
one() { # label, env, args
  env $2 timeout 600 python bench.py --no-cpu-baseline --no-step-parity $3 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$1', round(d['value'],1), round(d['ms_per_step'],3), round(d['gan_ms_per_cycle'],3))"
}
for rep in 1 2; do
  for m in 7 0 1 2 4 3 5 6; do one b64_sites$m "M355_SN_PREFETCH_SITES=$m" ""; done
  for m in 7 0 1 2 4 6; do one b16_sites$m "M355_SN_PREFETCH_SITES=$m" "--batch 16 --workload gan"; done
done 2>&1 | tee gpurun_out/r06_47_sn_sites.txt
