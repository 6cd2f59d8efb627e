#!/bin/bash
# round 5 run 13: same-box A/B of the round-4 tree (_r04_tree: commit 19ea610 built beside this one) and the final tree: headline batch,
# and the host-bound small batch
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
run() {  # tree, args
  ( cd $1 && timeout 600 python bench.py $2 --no-cpu-baseline $3 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('%8.1f samples/s  %7.3f ms/step  gan %s ms/cycle' % (d['value'], d['ms_per_step'], d.get('gan_ms_per_cycle')))" )
}
( for rep in 1 2; do
    echo "r04 tree  batch 64:  $(run _r04_tree '' '')"
    echo "r05 tree  batch 64:  $(run . '' '--no-step-parity')"
  done
  for rep in 1 2; do
    echo "r04 tree  gan batch 16: $(run _r04_tree '--workload gan --batch 16 --steps 20 --warmup 5' '')"
    echo "r05 tree  gan batch 16: $(run . '--workload gan --batch 16 --steps 20 --warmup 5' '--no-step-parity')"
  done ) > gpurun_out/r05_ab_r04_vs_r05.txt 2>&1
cat gpurun_out/r05_ab_r04_vs_r05.txt
