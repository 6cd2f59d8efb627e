#!/bin/bash
# round 6 run 52: queue priority of the second streams (M355_SIDE_PRIORITY; the range the runtime offers is printed first): does a LOW-priority side
# stream stop the mesh discriminator from delaying the persistent class kernels beside the 128-sample D step (threshold 128 = fork everywhere)?
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
python -c "import torch; print('priority range', torch.cuda.Stream.priority_range())" 2>/dev/null | tee gpurun_out/r06_52_priority.txt
one() { # label, env, args
  env $2 timeout 600 python bench.py --no-cpu-baseline --no-step-parity $3 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$1', round(d['value'],1), round(d['ms_per_step'],3), round(d['gan_ms_per_cycle'],3))"
}
for rep in 1 2; do
  one b64_fork96_prio0 "A=1" ""
  one b64_fork96_low "M355_SIDE_PRIORITY=1" ""
  one b64_fork128_prio0 "M355_FORK_MAX_BATCH=128" ""
  one b64_fork128_low "M355_FORK_MAX_BATCH=128 M355_SIDE_PRIORITY=1" ""
  one b64_fork96_high "M355_SIDE_PRIORITY=-1" ""
  one b16_prio0 "A=1" "--batch 16 --workload gan"
  one b16_low "M355_SIDE_PRIORITY=1" "--batch 16 --workload gan"
done 2>&1 | tee -a gpurun_out/r06_52_priority.txt
