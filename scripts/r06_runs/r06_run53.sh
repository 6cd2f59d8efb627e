#!/bin/bash
# round 6 run 53 (run twice: the second time with RG = 1 below 16 rows): the ordered partial-row sum with four row groups per output (k_wgrad_part_sum) + the unrolled 16 -> 9 fold, against the previous
# build (lib/libm355_prev.so): conv tests, per-launch constants of the layers that end in it, bench lines at batch 64 / 16
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_conv_gpu.py tests/test_headline_batch_gpu.py -m gpu -q -x 2>&1 | tail -2
for l in libm355_prev.so libm355.so; do echo "M355_LIB=$l"; M355_LIB=$l timeout 300 python scripts/probes/wgrad_fixed_cost.py 2>/dev/null; M355_LIB=$l timeout 300 python scripts/probes/launch_constants.py 2>/dev/null | grep "wgrad" | grep "D.conv1\|blk6.conv1\|blk5.conv1"; done 2>&1 | tee gpurun_out/r06_53_part_sum.txt
one() { # label, env, args
  env $2 timeout 600 python bench.py --no-cpu-baseline --no-step-parity $3 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); k=d['kernels_ms_per_step']; print('$1', round(d['value'],1), round(d['ms_per_step'],3), round(d['gan_ms_per_cycle'],3), 'wgrad_halo', round(k.get('k_wgrad_halo',0),3), 'wgrad_c8', round(k.get('k_wgrad_c8',0),3))"
}
for rep in 1 2; do
  one b64_prev "M355_LIB=libm355_prev.so" ""
  one b64_new "M355_LIB=libm355.so" ""
  one b16_prev "M355_LIB=libm355_prev.so" "--batch 16 --workload gan"
  one b16_new "M355_LIB=libm355.so" "--batch 16 --workload gan"
done 2>&1 | tee -a gpurun_out/r06_53_part_sum.txt
