#!/bin/bash
# round 6 run 35: bounding experiment for the spectral-norm chain (needs the temporary M355_DBG_SKIP_SN switch in gan_ops.SpectralNormGroup._launch:
# "if os.environ.get('M355_DBG_SKIP_SN') and slot launched twice: return" -- not in the tree) -> profiles/r06_sn_chain_bound.txt
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
one() { # label, env, args
  env $2 timeout 600 python bench.py --no-cpu-baseline --no-step-parity $3 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$1', round(d['value'],1), round(d['ms_per_step'],3), round(d['gan_ms_per_cycle'],3))"
}
for rep in 1 2; do
  one b64 "A=1" ""
  one b64_skip_sn "M355_DBG_SKIP_SN=1" ""
  one b16 "A=1" "--batch 16 --workload gan"
  one b16_skip_sn "M355_DBG_SKIP_SN=1" "--batch 16 --workload gan"
done 2>&1 | tee gpurun_out/r06_35_sn_bound.txt
