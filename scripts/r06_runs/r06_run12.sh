#!/bin/bash
# round 6 run 12: the small sub-pixel form with its size threshold (>= 4096 stored pixels per launch): whole GPU suite, same-box A/B
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 1700 python -m pytest tests -m gpu -q -x > gpurun_out/r06_12_all.log 2>&1; echo "all rc=$?" >> gpurun_out/r06_12_all.log
tail -4 gpurun_out/r06_12_all.log | cut -c1-300
one() { # label, env, args
  env $2 timeout 600 python bench.py --no-cpu-baseline --no-step-parity $3 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$1', round(d['value'],1), round(d['ms_per_step'],3), d.get('gan_ms_per_cycle'), round(d['kernels_ms_per_step'].get('k_conv_glds',0),3), d.get('parity_ok'))"
}
for rep in 1 2 3; do
  one base64 "M355_LIB=libm355_base.so" ""
  one new64 "M355_LIB=libm355.so" ""
done 2>&1 | tee gpurun_out/r06_12_ab.txt
one base16 "M355_LIB=libm355_base.so" "--batch 16 --workload gan" | tee -a gpurun_out/r06_12_ab.txt
one new16 "M355_LIB=libm355.so" "--batch 16 --workload gan" | tee -a gpurun_out/r06_12_ab.txt
