#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
timeout 900 python bench.py > gpurun_out/r04_v11_bench.json 2> gpurun_out/r04_v11_bench.err; tail -3 gpurun_out/r04_v11_bench.err | cut -c1-300
python - <<'P'
import json
d = json.load(open("gpurun_out/r04_v11_bench.json"))
print("r04_v11", round(d["value"], 1), round(d["ms_per_step"], 3), "parity_ok", d["parity_ok"])
print(d["parity"]); print(d["parity_gan"]); print({k: d["cpu_baseline"][k] for k in ("value", "unit", "cores", "kind")})
P
