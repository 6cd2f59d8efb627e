#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; rm -f gpurun_out/exact_report.jsonl
timeout 1500 python -m pytest tests/test_exact_mode_gpu.py -q --timeout 1200 > gpurun_out/r05_6_exact.log 2>&1; echo "exact rc=$?" >> gpurun_out/r05_6_exact.log
tail -12 gpurun_out/r05_6_exact.log | cut -c1-250
M355_NO_CLASS_STATS=1 M355_TOP=70 timeout 300 python scripts/layer_times.py 64 > gpurun_out/r05_6_layers_nocstats.txt 2>&1
M355_TOP=90 timeout 300 python scripts/layer_times.py 64 > gpurun_out/r05_6_layers.txt 2>&1
grep -E "up1|bn_stats_partial|total" gpurun_out/r05_6_layers_nocstats.txt | head; echo; grep -E "up1|bn_stats_partial|total" gpurun_out/r05_6_layers.txt | head -12
