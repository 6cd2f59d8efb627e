#!/bin/bash
# round 6 run 51: the N > 1 flow of bench.py at the HEADLINE per-GPU batch (64) with the new second-stream rule: two ranks sharing cuda:0 over gloo
# (M355_SHARE_GPU=1; the numbers mean nothing -- one GPU, a CPU transport -- the flow, the counters and the finiteness do)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
M355_SHARE_GPU=1 OMP_NUM_THREADS=1 timeout 900 python bench.py --gpus 2 --backend gloo --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/r06_51_two_ranks.json 2> gpurun_out/r06_51_two_ranks.err; echo "rc=$?"
python - <<'PY'
import json
d=json.loads([l for l in open('gpurun_out/r06_51_two_ranks.json') if l.startswith('{')][-1])
print(d['n_gpus'], d['config']['global_batch'], d['config']['parallelism'], d['config']['gan_streams'], 'grad allreduces', d['grad_allreduces_per_step'], 'syncbn', d['syncbn_collectives_per_step'], d['syncbn_transport'], 'MB', round(d['grad_allreduce_mb_per_step'],1), 'losses', d['config']['losses'], 'parity_ok', d.get('parity_ok'))
PY
