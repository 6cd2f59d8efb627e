#!/bin/bash
# round 6 run 39: where does graph replay stop paying?  bench.py --workload gan at batch 16 / 24 / 32 / 48, replay (default <= 32) against --no-graph
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
one() { # label, args
  timeout 600 python bench.py --no-cpu-baseline --no-step-parity --workload gan $2 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$1', round(d['value'],1), round(d['ms_per_step'],3), round(d['gan_ms_per_cycle'],3), d['config']['gan_launch'][:12])"
}
for rep in 1 2; do
  for b in 16 24 32 48; do
    one b${b}_graph "--batch $b --graph"
    one b${b}_eager "--batch $b --no-graph"
  done
done 2>&1 | tee gpurun_out/r06_39_graph_threshold.txt
