#!/bin/bash
# round 6 run 28: deterministic mode -- what would rows for EVERY stride-1 3x3 halo layer (M355_WGRAD_HALO_PART=2) buy over the integer cells?
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
one() { # label, env, args
  env $2 timeout 600 python bench.py --no-cpu-baseline --no-step-parity $3 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$1', round(d['value'],1), round(d['ms_per_step'],3), d.get('gan_ms_per_cycle'), round(d['kernels_ms_per_step']['k_wgrad_halo'],3), d.get('parity_ok'))"
}
for rep in 1 2; do
  one det_default "M355_DETERMINISTIC=1" ""
  one det_rows_all3x3 "M355_DETERMINISTIC=1 M355_WGRAD_HALO_PART=2" ""
done 2>&1 | tee gpurun_out/r06_28_det_rows.txt
( time timeout 900 python bench.py > /dev/null 2>&1 ) 2>&1 | grep real | tee -a gpurun_out/r06_28_det_rows.txt
