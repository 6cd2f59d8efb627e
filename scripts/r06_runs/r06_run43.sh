#!/bin/bash
# round 6 run 43: the second stream for the small branches (mesh head / mesh discriminator; auto at <= 32 samples per GPU) forced on at batch 64 and 48
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
one() { # label, env, args
  env $2 timeout 600 python bench.py --no-cpu-baseline --no-step-parity $3 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$1', round(d['value'],1), round(d['ms_per_step'],3), round(d['gan_ms_per_cycle'],3), d['config']['gan_streams'])"
}
for rep in 1 2 3; do
  one b64_auto "A=1" ""
  one b64_streams "M355_STREAMS=1" ""
  one b48_auto "A=1" "--batch 48 --workload gan --no-graph"
  one b48_streams "M355_STREAMS=1" "--batch 48 --workload gan --no-graph"
done 2>&1 | tee gpurun_out/r06_43_streams.txt
