#!/bin/bash
# round 5 run 17: the (hi, lo) cell pairs of the deterministic weight gradient -- the scale test on the new and on the previous
# library (single 2^-36 grid, kept as lib/libm355_fix1.so for this A/B only), the deterministic-mode tests, and the
# deterministic-mode step time with both libraries on the same box
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_conv_gpu.py -q -x -k "scale_free or bit_identically" > gpurun_out/r05_17_new.log 2>&1; tail -3 gpurun_out/r05_17_new.log | cut -c1-300
M355_LIB=libm355_fix1.so timeout 600 python -m pytest tests/test_conv_gpu.py -q -k "scale_free" > gpurun_out/r05_17_old.log 2>&1; tail -12 gpurun_out/r05_17_old.log | cut -c1-300
timeout 900 python -m pytest tests/test_gan_modules.py tests/test_headline_batch_gpu.py tests/test_exact_mode_gpu.py -q -x > gpurun_out/r05_17_gan.log 2>&1; tail -3 gpurun_out/r05_17_gan.log | cut -c1-300
for lib in libm355_fix1.so libm355.so libm355_fix1.so libm355.so; do
  M355_LIB=$lib M355_DETERMINISTIC=1 timeout 600 python bench.py --no-cpu-baseline --no-step-parity 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$lib', d['config'].get('deterministic'), round(d['value'],1), round(d['ms_per_step'],3))"
done
