#!/bin/bash
# round 6 run 50: re-measure rules of earlier rounds on the final kernels: k_wgrad_halo's stride-2 shape (M355_WGRAD_HALO_VARIANT: twin (default) /
# narrow / wide), the small-tile rule of k_conv_glds (M355_NO_SMALL_TILE), the split-K forward (M355_NO_SPLITK), in the bench line
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
for v in twin narrow wide; do echo "M355_WGRAD_HALO_VARIANT=$v"; M355_WGRAD_HALO_VARIANT=$v timeout 300 python scripts/dconv_ab.py 128 2>/dev/null | grep conv; done 2>&1 | tee gpurun_out/r06_50_rules.txt
one() { # label, env, args
  env $2 timeout 600 python bench.py --no-cpu-baseline --no-step-parity $3 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); k=d['kernels_ms_per_step']; print('$1', round(d['value'],1), round(d['ms_per_step'],3), round(d['gan_ms_per_cycle'],3), 'glds', round(k.get('k_conv_glds',0),3), 'wgrad_halo', round(k.get('k_wgrad_halo',0),3))"
}
for rep in 1 2; do
  one base "A=1" ""
  one no_small_tile "M355_NO_SMALL_TILE=1" ""
  one no_splitk "M355_NO_SPLITK=1" ""
  one b16_base "A=1" "--batch 16 --workload gan"
  one b16_no_small_tile "M355_NO_SMALL_TILE=1" "--batch 16 --workload gan"
  one b16_no_splitk "M355_NO_SPLITK=1" "--batch 16 --workload gan"
done 2>&1 | tee -a gpurun_out/r06_50_rules.txt
