#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/trace
for p in 1 2 3 4 5 6 7 8 9 10 11 12; do
  timeout 300 python scripts/r04_det_trace.py 16 256 3 gpurun_out/trace/s$p.txt 2>&1 | grep "^trace\|Error"
done
for p in 1 2 3 4 5 6 7 8; do
  TRACE_SYNC=0 timeout 300 python scripts/r04_det_trace.py 16 256 3 gpurun_out/trace/a$p.txt 2>&1 | grep "^trace\|Error"
done
