#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
python -c "import __graft_entry__ as g; g.build(); g.smoke()" 2>&1 | tail -1
timeout 1800 python -m pytest tests -m gpu -x -q 2>&1 | grep -v amdgpu.ids | tail -4
timeout 900 python bench.py > gpurun_out/r04_v13_bench.json 2> gpurun_out/r04_v13_bench.err
python - <<'P'
import json
d = json.load(open("gpurun_out/r04_v13_bench.json"))
print("r04_v13", round(d["value"], 1), round(d["ms_per_step"], 3), "parity_ok", d["parity_ok"], d["parity_gan"], "conv", round(d["roofline"]["all_conv_tflops"]), d["roofline"]["frac"])
P
timeout 600 python scripts/soak_determinism.py 3 128 256 2>&1 | grep -v amdgpu.ids | tail -5 | cut -c1-250
