#!/bin/bash
# round 6 run 27: the final tree under global switches -- the whole GPU suite in deterministic mode, with the second stream forced on, with
# the weight gradients' partial rows off (atomics everywhere) -- and the soaks of the other configurations (batch 32 at 256^2; 512^2 nd 3
# batch 16: BASELINE configs[4]/[5] shapes), the 512^2 stress line
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
M355_DETERMINISTIC=1 timeout 1700 python -m pytest tests -m gpu -q > gpurun_out/r06_27_det.log 2>&1; echo "det rc=$?" >> gpurun_out/r06_27_det.log; tail -3 gpurun_out/r06_27_det.log | cut -c1-300
M355_STREAMS=1 timeout 1700 python -m pytest tests -m gpu -q > gpurun_out/r06_27_streams.log 2>&1; echo "streams rc=$?" >> gpurun_out/r06_27_streams.log; tail -3 gpurun_out/r06_27_streams.log | cut -c1-300
M355_WGRAD_HALO_PART=0 timeout 1700 python -m pytest tests/test_gan_modules.py tests/test_conv_gpu.py tests/test_headline_batch_gpu.py -m gpu -q > gpurun_out/r06_27_norows.log 2>&1; echo "norows rc=$?" >> gpurun_out/r06_27_norows.log; tail -3 gpurun_out/r06_27_norows.log | cut -c1-300
( timeout 500 python scripts/soak_determinism.py 16 32 256; timeout 500 python scripts/soak_determinism.py 6 16 512 3 ) > gpurun_out/r06_27_soak.txt 2>&1
grep -c "SOAK OK" gpurun_out/r06_27_soak.txt; grep -a "SOAK\|differ" gpurun_out/r06_27_soak.txt | cut -c1-200 | tail -8
timeout 300 python scripts/stress_cfg5.py > gpurun_out/r06_27_cfg5_stress.json 2> gpurun_out/r06_27_cfg5.err; tail -c 500 gpurun_out/r06_27_cfg5_stress.json
