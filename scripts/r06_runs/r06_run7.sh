#!/bin/bash
# round 6 run 7: profile of this tree (bench line + rocprofv3 kernel stats + FETCH / WRITE passes), per-layer table, the other BASELINE
# configurations, and two SQ counter passes over the discriminator's big layers (what are k_wgrad_halo's waits on)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 1500 bash scripts/make_profile.sh r06_v1 --steps 20 --warmup 5
M355_TOP=150 timeout 300 python scripts/layer_times.py 64 > gpurun_out/r06_7_layers_b64.txt 2>&1; tail -1 gpurun_out/r06_7_layers_b64.txt
cd /tmp; export TMPDIR=/tmp
rocprofv3 -L 2>/dev/null | grep -o "SQ_[A-Z0-9_]*" | sort -u > $GRAFT_REPO_ROOT/gpurun_out/r06_7_sq_counters.txt; wc -l $GRAFT_REPO_ROOT/gpurun_out/r06_7_sq_counters.txt
p=A
for set in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES" "SQ_WAVE_CYCLES SQ_INSTS_VMEM SQ_INSTS_LDS SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"; do
  rm -rf /tmp/pmc_$p
  timeout 600 rocprofv3 --kernel-trace --pmc $set -d /tmp/pmc_$p -o pmc -- python $GRAFT_REPO_ROOT/scripts/pmc_dconv_r06.py 128 > $GRAFT_REPO_ROOT/gpurun_out/r06_7_pmc_$p.log 2>&1
  python $GRAFT_REPO_ROOT/scripts/pmc_sq_mfma.py /tmp/pmc_$p/pmc_results.db > $GRAFT_REPO_ROOT/gpurun_out/r06_7_pmc_sq_dconv_$p.txt 2>> $GRAFT_REPO_ROOT/gpurun_out/r06_7_pmc_$p.log
  tail -2 $GRAFT_REPO_ROOT/gpurun_out/r06_7_pmc_$p.log | cut -c1-200
  p=B
done
cd $GRAFT_REPO_ROOT
( for args in "--workload proj --batch 32 --points 2048 --grid 128" "--workload gan --batch 64 --res 512" "--workload proj --batch 16 --points 4096 --grid 128" "--workload recon --batch 50"; do
    echo "### bench.py $args"; timeout 600 python bench.py $args --no-cpu-baseline --no-step-parity --steps 10 --warmup 3 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read())
keep={k:d.get(k) for k in ('metric','value','unit','ms_per_step','parity_ok','proj_ms_per_step','gan_ms_per_cycle','proj_samples_per_s','gan_samples_per_s')}
keep['workload']=d['config']['workload']; keep['gan_launch']=d['config'].get('gan_launch'); r=d.get('roofline') or {}
keep['roofline']={k:r.get(k) for k in ('kernel','achieved','frac','all_conv_tflops','executed')}
print(json.dumps(keep))"
  done ) > gpurun_out/r06_other_configs.txt 2>&1
cut -c1-300 gpurun_out/r06_other_configs.txt
timeout 300 python scripts/stress_cfg5.py > gpurun_out/r06_cfg5_stress.json 2> gpurun_out/r06_7_cfg5.err; tail -c 400 gpurun_out/r06_cfg5_stress.json
