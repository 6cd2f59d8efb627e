#!/bin/bash
# round 4, GPU call 12: why the second stream changes bits in deterministic mode + the fused mask_cat loaders
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
timeout 600 python scripts/r04_fork_diag.py 2>&1 | grep -v amdgpu.ids | tail -30
timeout 900 python -m pytest tests/test_gan_io_gpu.py tests/test_gan_modules.py tests/test_headline_batch_gpu.py -m gpu -q -x --deselect tests/test_gan_modules.py::test_second_stream_branches_change_no_bit 2>&1 | tail -5
timeout 600 python bench.py --steps 20 --warmup 5 > gpurun_out/r04_v8_bench.json 2> gpurun_out/r04_v8_bench.err
python - <<'P'
import json
d = json.load(open("gpurun_out/r04_v8_bench.json"))
k = d["kernels_ms_per_step"]
print("r04_v8", round(d["value"], 1), round(d["ms_per_step"], 3), d["parity_ok"], round(d["gan_ms_per_cycle"], 3))
print({n: round(k[n], 3) for n in ("pool_pack_fwd", "mask_cat_fwd", "mask_cat_bwd", "pool_unpack_bwd") if n in k})
P
