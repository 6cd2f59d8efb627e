#!/bin/bash
# Single-GPU lines from which the 1 -> 8 GPU curve can be predicted: the per-GPU share of SURVEY 8d cfg 4
# (16 clouds of 4096 points + GAN batch 16) and the per-GPU batches of a strong-scaling run of the global batch 64 (32 / 16 / 8),
# each with the GAN cycle launched eagerly and replayed from one hipGraph.  -> gpurun_out/r03_scaling_inputs.json
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/gpurun_out/r03_scaling_inputs.json
mkdir -p $R/gpurun_out; : > $OUT.tmp
run() { timeout 300 python $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline "$@" 2>/dev/null | tail -1 >> $OUT.tmp; }
run --batch 64
run --batch 64 --graph
run --batch 16 --points 4096
run --batch 16 --points 4096 --graph
for b in 32 16 8; do run --batch $b; run --batch $b --graph; done
python - "$OUT.tmp" "$OUT" <<'PY'
import json, sys
rows = []
for ln in open(sys.argv[1]):
    if not ln.startswith("{"): continue
    d = json.loads(ln)
    rows.append({"per_gpu_batch": d["config"]["per_gpu_batch"], "points": d["config"]["points"], "gan_launch": d["config"]["gan_launch"],
                 "ms_per_step": d["ms_per_step"], "samples_per_s": d["value"], "proj_ms_per_step": d["proj_ms_per_step"],
                 "gan_ms_per_cycle": d["gan_ms_per_cycle"], "kernel_ms_per_step": sum(d["kernels_ms_per_step"].values()),
                 "all_conv_tflops": d["roofline"]["all_conv_tflops"]})
json.dump({"note": "one MI355X, bench.py --steps 20 --warmup 5; kernel_ms_per_step = sum of HIP-event kernel times of the eager step "
                   "(what a perfectly overlapped host would reach); predict N GPUs: step(N) ~ step(1 GPU at the per-GPU batch) + allreduce_ms "
                   "(75 MB flat fp32 per cycle) + 56 SyncBN all-reduces x latency", "rows": rows}, open(sys.argv[2], "w"), indent=1)
for r in rows: print(r)
PY
rm -f $OUT.tmp
