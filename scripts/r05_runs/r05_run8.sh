#!/bin/bash
# round 5 run 8: whole GPU suite (product + EXACT), incl. the data-parallel mesh-estimation step on two ranks sharing the GPU
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; rm -f gpurun_out/exact_report.jsonl
timeout 1700 python -m pytest tests -m gpu -q -x > gpurun_out/r05_8_all.log 2>&1; echo "all rc=$?" >> gpurun_out/r05_8_all.log
tail -25 gpurun_out/r05_8_all.log | cut -c1-300
