#!/bin/bash
# round 6 run 21: the sub-pixel (upsample + 3x3) weight gradients as partial rows + an ordered 16 -> 9 fold: conv tests in both modes,
# per-launch constants by switch (M355_WGRAD_UP_PART=0: zeroed cells + atomics), bench line by switch in both modes
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_conv_gpu.py -m gpu -q -x 2>&1 | tail -3
M355_DETERMINISTIC=1 timeout 900 python -m pytest tests/test_conv_gpu.py -m gpu -q -x 2>&1 | tail -3
for m in 0 1; do echo "M355_WGRAD_UP_PART=$m"; M355_WGRAD_UP_PART=$m timeout 300 python scripts/probes/launch_constants.py 2>/dev/null | grep "conv1 *wgrad"; done 2>&1 | tee gpurun_out/r06_21_up_probe.txt
one() { # label, env, args
  env $2 timeout 600 python bench.py --no-cpu-baseline --no-step-parity $3 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$1', round(d['value'],1), round(d['ms_per_step'],3), d.get('gan_ms_per_cycle'), round(d['kernels_ms_per_step']['k_wgrad_halo'],3), d.get('parity_ok'))"
}
for rep in 1 2; do
  one atomics "M355_WGRAD_UP_PART=0" ""
  one rows "M355_WGRAD_UP_PART=1" ""
done 2>&1 | tee gpurun_out/r06_21_bench_ab.txt
one atomics16 "M355_WGRAD_UP_PART=0" "--batch 16 --workload gan" | tee -a gpurun_out/r06_21_bench_ab.txt
one rows16 "M355_WGRAD_UP_PART=1" "--batch 16 --workload gan" | tee -a gpurun_out/r06_21_bench_ab.txt
one det_cells "M355_WGRAD_UP_PART=0 M355_DETERMINISTIC=1" "" | tee -a gpurun_out/r06_21_bench_ab.txt
one det_rows "M355_WGRAD_UP_PART=1 M355_DETERMINISTIC=1" "" | tee -a gpurun_out/r06_21_bench_ab.txt
one rows16_all3x3 "M355_WGRAD_HALO_PART=2" "--batch 16 --workload gan" | tee -a gpurun_out/r06_21_bench_ab.txt
