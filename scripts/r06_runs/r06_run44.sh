#!/bin/bash
# round 6 run 44: the fork threshold (samples per LAUNCH: a D step sees 2 x the per-GPU batch) -- 32 (default), 64, 96, 128, all -- at batch 64 / 48 / 40
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
one() { # label, env, args
  env $2 timeout 600 python bench.py --no-cpu-baseline --no-step-parity $3 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$1', round(d['value'],1), round(d['ms_per_step'],3), round(d['gan_ms_per_cycle'],3))"
}
for rep in 1 2; do
  for t in 32 64 96 128; do one b64_fork$t "M355_FORK_MAX_BATCH=$t" ""; done
  for t in 32 48 96; do one b48_fork$t "M355_FORK_MAX_BATCH=$t" "--batch 48 --workload gan --no-graph"; done
  for t in 32 80; do one b40_fork$t "M355_FORK_MAX_BATCH=$t" "--batch 40 --workload gan --no-graph"; done
done 2>&1 | tee gpurun_out/r06_44_fork_threshold.txt
