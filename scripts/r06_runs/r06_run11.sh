#!/bin/bash
# round 6 run 11: the small upsample + 3x3 stages in the sub-pixel form on the generic kernel (forward + dgrad): conv tests, model-level
# tests, then same-box A/B against the previous library (lib/libm355_base.so) at batch 64 and batch 16
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_conv_gpu.py tests/test_gan_modules.py tests/test_headline_batch_gpu.py tests/test_exact_mode_gpu.py -m gpu -q -x > gpurun_out/r06_11_tests.log 2>&1; echo "rc=$?" >> gpurun_out/r06_11_tests.log
tail -4 gpurun_out/r06_11_tests.log | cut -c1-400
one() { # label, env, args
  env $2 timeout 600 python bench.py --no-cpu-baseline --no-step-parity $3 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$1', round(d['value'],1), round(d['ms_per_step'],3), d.get('gan_ms_per_cycle'), round(d['kernels_ms_per_step'].get('k_conv_glds',0),3), d.get('parity_ok'))"
}
for rep in 1 2; do
  one base64 "M355_LIB=libm355_base.so" ""
  one new64 "M355_LIB=libm355.so" ""
  one base16 "M355_LIB=libm355_base.so" "--batch 16 --workload gan"
  one new16 "M355_LIB=libm355.so" "--batch 16 --workload gan"
done 2>&1 | tee gpurun_out/r06_11_ab.txt
M355_TOP=150 timeout 300 python scripts/layer_times.py 64 2>/dev/null | grep "up1" > gpurun_out/r06_11_up1_new.txt
M355_LIB=libm355_base.so M355_TOP=150 timeout 300 python scripts/layer_times.py 64 2>/dev/null | grep "up1" > gpurun_out/r06_11_up1_base.txt
echo new; cat gpurun_out/r06_11_up1_new.txt | cut -c1-150; echo base; cat gpurun_out/r06_11_up1_base.txt | cut -c1-150
