#!/bin/bash
# round 5 run 19: same-box comparison of the launch modes on the final tree: eager (default), --graph, M355_STREAMS=1, both
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
one() { # label, env, args
  env $2 timeout 600 python bench.py --no-cpu-baseline --no-step-parity $3 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$1', round(d['value'],1), round(d['ms_per_step'],3), d['config'].get('gan_launch'), d['config'].get('gan_streams'), d.get('parity_ok'))"
}
for rep in 1 2; do
  one eager "X=1" ""
  one graph "X=1" "--graph"
  one eager_streams "M355_STREAMS=1" ""
  one graph_streams "M355_STREAMS=1" "--graph"
done
