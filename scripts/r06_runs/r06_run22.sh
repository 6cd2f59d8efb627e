#!/bin/bash
# round 6 run 22: sub-pixel partial rows + the batch-dependent rule for the 3x3 layers: whole GPU suite (default), model-level tests in
# deterministic mode, bench lines at batch 64 / 16 by M355_WGRAD_HALO_PART (1 default, 3 = no 3x3 rows)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 1700 python -m pytest tests -m gpu -q -x > gpurun_out/r06_22_all.log 2>&1; echo "all rc=$?" >> gpurun_out/r06_22_all.log; tail -3 gpurun_out/r06_22_all.log | cut -c1-300
M355_DETERMINISTIC=1 timeout 1700 python -m pytest tests/test_conv_gpu.py tests/test_gan_modules.py tests/test_headline_batch_gpu.py tests/test_exact_mode_gpu.py -m gpu -q > gpurun_out/r06_22_det.log 2>&1; echo "det rc=$?" >> gpurun_out/r06_22_det.log; tail -3 gpurun_out/r06_22_det.log | cut -c1-300
one() { # label, env, args
  env $2 timeout 600 python bench.py --no-cpu-baseline --no-step-parity $3 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$1', round(d['value'],1), round(d['ms_per_step'],3), d.get('gan_ms_per_cycle'), round(d['kernels_ms_per_step']['k_wgrad_halo'],3), d.get('parity_ok'))"
}
for rep in 1 2; do
  one b64_default "A=1" ""
  one b16_default "A=1" "--batch 16 --workload gan"
  one b16_no3x3rows "M355_WGRAD_HALO_PART=3" "--batch 16 --workload gan"
  one b32_default "A=1" "--batch 32 --workload gan"
  one b32_no3x3rows "M355_WGRAD_HALO_PART=3" "--batch 32 --workload gan"
done 2>&1 | tee gpurun_out/r06_22_bench_ab.txt
