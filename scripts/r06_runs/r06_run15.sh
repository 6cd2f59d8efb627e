#!/bin/bash
# round 6 run 15: A/B -- a bias-less tile's first MFMAs take the constant 0 as their C operand instead of 64 registers cleared in the epilogue
# (-DM355_ZINIT, lib/libm355_zinit.so) against the current library: per layer with output hashes, the bench line, the conv tests
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
for rep in 1 2; do
  for lib in libm355.so libm355_zinit.so; do M355_LIB=$lib timeout 300 python scripts/dconv_ab.py 128 2>/dev/null; done
done 2>&1 | tee gpurun_out/r06_15_dconv_ab.txt
one() { # label, env, args
  env $2 timeout 600 python bench.py --no-cpu-baseline --no-step-parity $3 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$1', round(d['value'],1), round(d['ms_per_step'],3), d.get('gan_ms_per_cycle'), round(d['kernels_ms_per_step']['k_conv_halo'],3), d.get('parity_ok'))"
}
for rep in 1 2 3; do
  one base "M355_LIB=libm355.so" ""
  one zinit "M355_LIB=libm355_zinit.so" ""
done 2>&1 | tee gpurun_out/r06_15_bench_ab.txt
M355_LIB=libm355_zinit.so timeout 900 python -m pytest tests/test_conv_gpu.py -m gpu -q -x 2>&1 | tail -3
