#!/bin/bash
# round 6 run 38: CPU issue time of the eager cycle against its GPU time, batch 64 / 32 / 16 (scripts/probes/cpu_issue_time.py)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
for b in 64 32 16; do timeout 300 python scripts/probes/cpu_issue_time.py $b 10 2>/dev/null; done | tee gpurun_out/r06_38_cpu_issue.txt
