#!/bin/bash
# round 6 run 46: the blocks' 1x1 shortcut convs on the second stream (M355_FORK_SHORTCUT=1, experiment) at batch 64 / 48 / 16
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
one() { # label, env, args
  env $2 timeout 600 python bench.py --no-cpu-baseline --no-step-parity $3 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$1', round(d['value'],1), round(d['ms_per_step'],3), round(d['gan_ms_per_cycle'],3), d.get('parity_ok'))"
}
for rep in 1 2 3; do
  one b64 "A=1" ""
  one b64_fork_sc "M355_FORK_SHORTCUT=1" ""
  one b16 "A=1" "--batch 16 --workload gan"
  one b16_fork_sc "M355_FORK_SHORTCUT=1" "--batch 16 --workload gan"
done 2>&1 | tee gpurun_out/r06_46_fork_shortcut.txt
M355_FORK_SHORTCUT=1 timeout 900 python -m pytest tests/test_gan_modules.py -m gpu -q -x 2>&1 | tail -2
