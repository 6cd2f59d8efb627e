#!/bin/bash
# round 6 run 37: where is the idle time of the eager batch-64 cycle (5 % of the span)?  rocprofv3 kernel trace of scripts/graph_trace.py 64 --eager,
# gap histogram + the kernel pairs around the gaps >= 20 us (scripts/rocpd_gaps.py); the same for the batch-16 replay
cd /tmp; export TMPDIR=/tmp; mkdir -p $GRAFT_REPO_ROOT/gpurun_out
for cfg in "64 30 --eager" "16 40"; do
  tag=$(echo $cfg | tr ' ' '_' | tr -d '-')
  rm -rf /tmp/prof_$tag
  timeout 600 rocprofv3 --kernel-trace -d /tmp/prof_$tag -o t -- python $GRAFT_REPO_ROOT/scripts/graph_trace.py $cfg > $GRAFT_REPO_ROOT/gpurun_out/r06_37_trace_$tag.log 2>&1
  python $GRAFT_REPO_ROOT/scripts/rocpd_gaps.py /tmp/prof_$tag/t_results.db 0.5 > $GRAFT_REPO_ROOT/gpurun_out/r06_37_gaps_$tag.txt 2>&1
  head -32 $GRAFT_REPO_ROOT/gpurun_out/r06_37_gaps_$tag.txt | cut -c1-200
done
