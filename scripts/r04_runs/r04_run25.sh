#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/trace2
echo "--- default build"
for p in 1 2 3 4 5 6 7 8 9 10 11 12 13 14; do
  TRACE_SYNC=0 timeout 300 python scripts/r04_det_trace.py 16 256 2 gpurun_out/trace2/d$p.txt 2>&1 | grep "^trace\|Error"
done
echo "--- no epilogue credit"
for p in 1 2 3 4 5 6 7 8 9 10 11 12 13 14; do
  M355_LIB=$GRAFT_REPO_ROOT/2dimageto3dmodel_amd/lib/libm355_nocredit.so TRACE_SYNC=0 timeout 300 python scripts/r04_det_trace.py 16 256 2 gpurun_out/trace2/n$p.txt 2>&1 | grep "^trace\|Error"
done
