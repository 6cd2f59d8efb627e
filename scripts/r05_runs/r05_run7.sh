#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; rm -f gpurun_out/exact_report.jsonl
timeout 1500 python -m pytest tests/test_exact_mode_gpu.py -q --timeout 1200 > gpurun_out/r05_7_exact.log 2>&1; echo "exact rc=$?" >> gpurun_out/r05_7_exact.log
tail -12 gpurun_out/r05_7_exact.log | cut -c1-250
