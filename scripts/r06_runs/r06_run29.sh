#!/bin/bash
# round 6 run 29: k_conv_c8 (D.conv1 forward, 1.07 GB of output per launch): plain against non-temporal 16-byte output stores (M355_C8_NT=1)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
for rep in 1 2; do for nt in 0 1; do echo "M355_C8_NT=$nt"; M355_C8_NT=$nt timeout 300 python scripts/c8_rate.py 1.0 2>/dev/null | grep "fwd\|dgrad"; done; done 2>&1 | tee gpurun_out/r06_29_c8_nt.txt
