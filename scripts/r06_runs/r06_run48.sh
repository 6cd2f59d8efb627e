#!/bin/bash
# round 6 run 48: separate thresholds for the two forked branches -- the generator's mesh head (M355_FORK_MAX_BATCH, default 96) and the mesh
# discriminator (M355_FORK_MAX_BATCH_D, A/B switch): at batch 64 the G step's D pass sees 64 samples, the D steps' 128
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
one() { # label, env, args
  env $2 timeout 600 python bench.py --no-cpu-baseline --no-step-parity $3 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$1', round(d['value'],1), round(d['ms_per_step'],3), round(d['gan_ms_per_cycle'],3))"
}
for rep in 1 2 3; do
  one b64_head96_meshD96 "A=1" ""
  one b64_head96_meshD32 "M355_FORK_MAX_BATCH_D=32" ""
  one b64_head128_meshD32 "M355_FORK_MAX_BATCH=128 M355_FORK_MAX_BATCH_D=32" ""
  one b64_head32 "M355_FORK_MAX_BATCH=32" ""
done 2>&1 | tee gpurun_out/r06_48_fork_branches.txt
