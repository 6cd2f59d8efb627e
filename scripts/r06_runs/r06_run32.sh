#!/bin/bash
# round 6 run 32: cache policy of the ACTIVATION tiles' LDS-DMA in the halo kernels (k_conv_halo / k_wgrad_halo): default against
# non-temporal (build -DM355_DMA_X_AUX=2 -> lib/libm355_nt.so): the discriminator's big layers alone (with output hashes), then the bench line
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
for rep in 1 2; do for l in libm355.so libm355_nt.so; do echo "M355_LIB=$l"; M355_LIB=$l timeout 300 python scripts/dconv_ab.py 128 2>/dev/null; done; done 2>&1 | tee gpurun_out/r06_32_dconv.txt
one() { # label, env
  env $2 timeout 600 python bench.py --no-cpu-baseline --no-step-parity 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); k=d['kernels_ms_per_step']; g=lambda n: round(k.get(n,0),3); print('$1', round(d['value'],1), round(d['ms_per_step'],3), round(d['gan_ms_per_cycle'],3), 'halo', g('k_conv_halo'), 'wgrad_halo', g('k_wgrad_halo'), d.get('parity_ok'))"
}
for rep in 1 2 3; do
  one base "M355_LIB=libm355.so"
  one nt "M355_LIB=libm355_nt.so"
done 2>&1 | tee gpurun_out/r06_32_bench.txt
