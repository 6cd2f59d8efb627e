#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
timeout 1500 python -m pytest tests/test_gan_modules.py tests/test_headline_batch_gpu.py -m gpu -q 2>&1 | grep -v amdgpu.ids | tail -4 | cut -c1-300
for b in 64 256; do
  timeout 900 python bench.py --batch $b --steps 6 --warmup 2 --no-cpu-baseline > /tmp/b.json 2> /tmp/b.err || tail -3 /tmp/b.err | cut -c1-400
  python - $b <<'P'
import json, sys
try:
    d = json.load(open("/tmp/b.json"))
    print(f"batch {sys.argv[1]}: {d['value']:.1f} samples/s, {d['ms_per_step']:.2f} ms per step, parity_ok {d['parity_ok']}, gan parity {d['parity_gan']['ok']}, conv aggregate {d['roofline']['all_conv_tflops']:.0f} TF")
except Exception as e:
    print("batch", sys.argv[1], "no result:", e)
P
done
