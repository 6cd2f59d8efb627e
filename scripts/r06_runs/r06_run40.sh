#!/bin/bash
# round 6 run 40: the new full-size test (partial rows against the exact integer sums at the timed batch), both modes
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_headline_batch_gpu.py -m gpu -q -x 2>&1 | tail -3
M355_DETERMINISTIC=1 timeout 900 python -m pytest tests/test_headline_batch_gpu.py -m gpu -q -x 2>&1 | tail -3
