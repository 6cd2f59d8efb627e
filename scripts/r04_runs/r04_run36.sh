#!/bin/bash
# the model-level GPU tests on the FALLBACK kernels (every A/B switch that removes a specialised kernel): the generic paths that
# unusual shapes take must give the same answers
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
T="tests/test_gan_modules.py tests/test_headline_batch_gpu.py tests/test_reconstruction.py tests/test_recon_step.py tests/test_gan_elem_gpu.py tests/test_gan_io_gpu.py tests/test_mesh.py"
echo "--- A: no c8 / head5 / pack1x4 / split-K / tiled weight prep / deferred finish / fused Adam"
M355_NO_C8=1 M355_NO_C8_DGRAD=1 M355_NO_HEAD5=1 M355_NO_PACK1X4=1 M355_NO_SPLITK=1 M355_NO_WPREP_TILED=1 M355_NO_DEFER_FINISH=1 M355_NO_FUSED_ADAM=1 \
  timeout 1500 python -m pytest $T -m gpu -q 2>&1 | grep -v amdgpu.ids | tail -8 | cut -c1-300
echo "--- B: no halo family (forward / dgrad / wgrad), no fused statistics, no mask bits, no small tiles"
M355_CONV_HALO=0 M355_NO_WGRAD_HALO=1 M355_NO_CONV_STATS=1 M355_NO_MASKBITS=1 M355_NO_SMALL_TILE=1 M355_NO_DIRECT_REPLICATE=1 \
  timeout 1500 python -m pytest $T -m gpu -q 2>&1 | grep -v amdgpu.ids | tail -8 | cut -c1-300
echo "--- C: halo variants off one by one (stride-2 classes, resident weights, class pairs, 8-wave only)"
M355_NO_HALO_S2=1 M355_NO_HALO_RES=1 M355_NO_HALO_PAIR=1 M355_HALO_8W_ONLY=1 \
  timeout 1500 python -m pytest $T -m gpu -q 2>&1 | grep -v amdgpu.ids | tail -8 | cut -c1-300
