#!/bin/bash
# round 6 run 19: the stride-2 class weight gradients as partial rows + an ordered sum (no same-address atomics): conv tests (weight gradients
# against fp32 torch at the unchanged 2e-4), fixed-cost probe by switch (M355_WGRAD_HALO_PART=0 atomics / 1 classes / 2 also the 3x3 layers),
# default and deterministic mode; then the bench line by switch
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_conv_gpu.py -m gpu -q -x 2>&1 | tail -3
for m in 0 1 2; do echo "M355_WGRAD_HALO_PART=$m"; M355_WGRAD_HALO_PART=$m timeout 300 python scripts/probes/wgrad_fixed_cost.py 2>/dev/null; done 2>&1 | tee gpurun_out/r06_19_wgrad_probe.txt
echo deterministic | tee -a gpurun_out/r06_19_wgrad_probe.txt
for m in 0 1; do echo "M355_WGRAD_HALO_PART=$m"; M355_DETERMINISTIC=1 M355_WGRAD_HALO_PART=$m timeout 300 python scripts/probes/wgrad_fixed_cost.py 2>/dev/null; done 2>&1 | tee -a gpurun_out/r06_19_wgrad_probe.txt
one() { # label, env, args
  env $2 timeout 600 python bench.py --no-cpu-baseline --no-step-parity $3 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$1', round(d['value'],1), round(d['ms_per_step'],3), d.get('gan_ms_per_cycle'), round(d['kernels_ms_per_step']['k_wgrad_halo'],3), d.get('parity_ok'))"
}
for rep in 1 2; do
  one atomics "M355_WGRAD_HALO_PART=0" ""
  one rows "M355_WGRAD_HALO_PART=1" ""
  one rows_all "M355_WGRAD_HALO_PART=2" ""
done 2>&1 | tee gpurun_out/r06_19_bench_ab.txt
one atomics16 "M355_WGRAD_HALO_PART=0" "--batch 16 --workload gan" | tee -a gpurun_out/r06_19_bench_ab.txt
one rows16 "M355_WGRAD_HALO_PART=1" "--batch 16 --workload gan" | tee -a gpurun_out/r06_19_bench_ab.txt
one det_atomics "M355_WGRAD_HALO_PART=0 M355_DETERMINISTIC=1" "" | tee -a gpurun_out/r06_19_bench_ab.txt
one det_rows "M355_WGRAD_HALO_PART=1 M355_DETERMINISTIC=1" "" | tee -a gpurun_out/r06_19_bench_ab.txt
