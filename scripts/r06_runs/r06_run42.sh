#!/bin/bash
# round 6 run 42: bounding experiment -- what would folding bn_finalize (42 launches of ~5 us per cycle, each between a conv and its affine pass) buy?
# needs the temporary M355_DBG_SKIP_BNFIN switch in gan_ops (coefficients of the 4th call reused, no launch: timing only; not in the tree)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
one() { # label, env, args
  env $2 timeout 600 python bench.py --no-cpu-baseline --no-step-parity $3 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$1', round(d['value'],1), round(d['ms_per_step'],3), round(d['gan_ms_per_cycle'],3))"
}
for rep in 1 2 3; do
  one b64 "A=1" ""
  one b64_skip_bnfin "M355_DBG_SKIP_BNFIN=1" ""
  one b16 "A=1" "--batch 16 --workload gan"
  one b16_skip_bnfin "M355_DBG_SKIP_BNFIN=1" "--batch 16 --workload gan"
done 2>&1 | tee gpurun_out/r06_42_bnfin_bound.txt
