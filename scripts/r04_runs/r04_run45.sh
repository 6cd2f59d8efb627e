#!/bin/bash
# the benchmarked step at 2x / 4x the batch per GPU (288 GB of HBM: what does not fit 64 samples should not be the limit)
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
for b in 128 256; do
  timeout 900 python bench.py --batch $b --steps 6 --warmup 2 --no-cpu-baseline > /tmp/b.json 2> /tmp/b.err || tail -3 /tmp/b.err | cut -c1-300
  python - $b <<'P'
import json, sys, torch
try:
    d = json.load(open("/tmp/b.json"))
    print(f"batch {sys.argv[1]}: {d['value']:.1f} samples/s, {d['ms_per_step']:.2f} ms per step, parity_ok {d['parity_ok']}, conv aggregate {d['roofline']['all_conv_tflops']:.0f} TF, "
          f"dominant {d['roofline']['kernel']} {d['roofline']['frac']:.3f}")
except Exception as e:
    print("batch", sys.argv[1], "no result:", e)
P
done
