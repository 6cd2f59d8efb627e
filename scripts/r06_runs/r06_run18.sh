#!/bin/bash
# round 6 run 18: k_conv_c8 at THREE workgroups per CU (168 registers, 5 spilled, 768 persistent workgroups) against two (215 registers,
# 512 workgroups): the two users of the kernel at batch 128 / 64, then the bench line
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
for rep in 1 2; do
  M355_LIB=libm355.so timeout 300 python scripts/bench_c8.py 128 2>/dev/null
  M355_LIB=libm355_c8o3.so M355_C8_WGS=768 timeout 300 python scripts/bench_c8.py 128 2>/dev/null
  M355_LIB=libm355_c8o3.so M355_C8_WGS=512 timeout 300 python scripts/bench_c8.py 128 2>/dev/null
done 2>&1 | tee gpurun_out/r06_18_c8_ab.txt
one() { # label, env, args
  env $2 timeout 600 python bench.py --no-cpu-baseline --no-step-parity $3 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$1', round(d['value'],1), round(d['ms_per_step'],3), d.get('gan_ms_per_cycle'), round(d['kernels_ms_per_step']['k_conv_c8'],3), d.get('parity_ok'))"
}
for rep in 1 2; do
  one base "M355_LIB=libm355.so" ""
  one c8o3 "M355_LIB=libm355_c8o3.so M355_C8_WGS=768" ""
done 2>&1 | tee gpurun_out/r06_18_bench_ab.txt
