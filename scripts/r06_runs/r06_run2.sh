#!/bin/bash
# round 6 run 2: spectral-norm prefetch (gan_ops.SpectralNormGroup.prefetch): its bit-identity test + the model-level tests, then a same-box
# A/B by switch (the switch only moves launches between streams: no code difference in any kernel) at batch 64 eager and batch 16 graph;
# rocprofv3 kernel trace of the batch-16 graph replay (true per-kernel durations: HIP-event timers of a host-bound eager run are inflated)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gan_modules.py tests/test_distributed_gpu.py -m gpu -q -x > gpurun_out/r06_2_tests.log 2>&1; echo "rc=$?" >> gpurun_out/r06_2_tests.log
tail -3 gpurun_out/r06_2_tests.log | cut -c1-300
one() { # label, env, args
  env $2 timeout 600 python bench.py --no-cpu-baseline --no-step-parity $3 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$1', round(d['value'],1), round(d['ms_per_step'],3), d.get('gan_ms_per_cycle'), d['config'].get('gan_launch'), d.get('parity_ok'))"
}
for rep in 1 2; do
  one b64_prefetch_off "M355_SN_PREFETCH=0" ""
  one b64_prefetch_on "M355_SN_PREFETCH=1" ""
  one b16g_prefetch_off "M355_SN_PREFETCH=0" "--batch 16 --workload gan --graph"
  one b16g_prefetch_on "M355_SN_PREFETCH=1" "--batch 16 --workload gan --graph"
done 2>&1 | tee gpurun_out/r06_2_ab.txt
cd /tmp; export TMPDIR=/tmp; rm -rf /tmp/prof_b16
timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof_b16 -o b16 -- python $GRAFT_REPO_ROOT/bench.py --batch 16 --workload gan --graph --steps 20 --warmup 3 --no-cpu-baseline --no-step-parity > $GRAFT_REPO_ROOT/gpurun_out/r06_2_rocprof_b16.log 2>&1
python $GRAFT_REPO_ROOT/scripts/rocpd_stats.py /tmp/prof_b16/b16_results.db $GRAFT_REPO_ROOT/gpurun_out/r06_2_b16_graph_kernel_stats.csv | tail -3
ls /tmp/prof_b16 | head
python $GRAFT_REPO_ROOT/scripts/rocpd_gaps.py /tmp/prof_b16/b16_results.db 0.4 > $GRAFT_REPO_ROOT/gpurun_out/r06_2_b16_graph_gaps.txt 2>&1; head -3 $GRAFT_REPO_ROOT/gpurun_out/r06_2_b16_graph_gaps.txt
