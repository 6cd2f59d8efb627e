#!/bin/bash
# round 6 run 41: long soaks on the final tree (a rare race has more launches to show up): 48 cycles at batch 64 / 256^2, 100 cycles at batch 8 /
# 128^2 (the script's default shape), 24 cycles at batch 16 / 512^2 nd 3; the repeat-under-memory-pressure tests five times over
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
( timeout 900 python scripts/soak_determinism.py 48 64 256; timeout 600 python scripts/soak_determinism.py 100 8 128; timeout 900 python scripts/soak_determinism.py 24 16 512 3 ) > gpurun_out/r06_41_soak_long.txt 2>&1
grep -c "SOAK OK" gpurun_out/r06_41_soak_long.txt; grep -a "SOAK\|differ" gpurun_out/r06_41_soak_long.txt | cut -c1-230 | tail -14
for i in 1 2 3 4 5; do timeout 600 python -m pytest tests/test_conv_gpu.py -m gpu -q -k "memory_pressure" 2>&1 | tail -1; done | tee gpurun_out/r06_41_pressure.txt
