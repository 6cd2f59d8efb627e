#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
echo "--- A (which switch breaks the spectral-norm group test?)"
for sw in M355_NO_WPREP_TILED M355_NO_DEFER_FINISH M355_NO_SPLITK M355_NO_C8; do
  echo "$sw: $(env $sw=1 timeout 600 python -m pytest tests/test_gan_elem_gpu.py -m gpu -q -k spectral_norm_group 2>&1 | grep -v amdgpu.ids | tail -1)"
done
M355_NO_WPREP_TILED=1 timeout 600 python -m pytest tests/test_gan_elem_gpu.py -m gpu -q -k spectral_norm_group 2>&1 | grep -v amdgpu.ids | grep "^E" | head -12 | cut -c1-400
echo "--- C one switch at a time"
for sw in M355_NO_HALO_S2 M355_NO_HALO_RES M355_NO_HALO_PAIR M355_HALO_8W_ONLY; do
  echo "$sw: $(env $sw=1 timeout 900 python -m pytest tests/test_gan_modules.py tests/test_headline_batch_gpu.py tests/test_gan_elem_gpu.py -m gpu -q -x 2>&1 | grep -v amdgpu.ids | grep '^E  \|passed\|failed' | head -4 | cut -c1-500)"
done
