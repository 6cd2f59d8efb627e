#!/bin/bash
# round 5 run 12: the other BASELINE configurations on the final tree (config 5 stress; configs 1..3 and 512^2 through bench.py)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 300 python scripts/stress_cfg5.py > gpurun_out/r05_cfg5_stress.json 2> gpurun_out/r05_12_cfg5.err; tail -c 600 gpurun_out/r05_cfg5_stress.json
( for args in "--workload proj --batch 32 --points 2048 --grid 128" "--workload gan --batch 16 --res 256" "--workload gan --batch 64 --res 512" "--workload proj --batch 16 --points 4096 --grid 128" "--workload recon --batch 50"; do
    echo "### bench.py $args"; timeout 600 python bench.py $args --no-cpu-baseline --no-step-parity --steps 10 --warmup 3 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read())
keep={k:d.get(k) for k in ('metric','value','unit','ms_per_step','parity_ok','proj_ms_per_step','gan_ms_per_cycle','proj_samples_per_s','gan_samples_per_s')}
keep['workload']=d['config']['workload']; r=d.get('roofline') or {}
keep['roofline']={k:r.get(k) for k in ('kernel','achieved','frac','all_conv_tflops','executed')}
print(json.dumps(keep))"
  done ) > gpurun_out/r05_other_configs.txt 2>&1
cat gpurun_out/r05_other_configs.txt | cut -c1-400
