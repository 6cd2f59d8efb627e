#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
for i in 1 2; do
timeout 900 python bench.py --steps 6 --warmup 2 --no-cpu-baseline > /tmp/b.json 2> /tmp/b.err || tail -3 /tmp/b.err | cut -c1-400
python - <<'P'
import json
d = json.load(open("/tmp/b.json"))
print(round(d["value"], 1), d["parity_ok"], d["parity_gan"])
P
done
