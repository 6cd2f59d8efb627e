#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out
cd $R
timeout 1800 python -m pytest tests -m gpu -q 2>&1 | tail -12 > $OUT/r04_run7_pytest.txt; cat $OUT/r04_run7_pytest.txt
python scripts/chamfer_rate.py > $OUT/r04_chamfer_rate2.txt 2>&1; tail -6 $OUT/r04_chamfer_rate2.txt
(MASTER_PORT=29617 HSA_ENABLE_IPC_MODE_LEGACY=0 timeout 400 python tests/_rccl_single_rank.py --graph 2> $OUT/r04_rccl_graph_probe.err | tail -1 > $OUT/r04_rccl_graph_probe.json; echo "probe rc=$?"; python -c "
import json; print(json.load(open('$OUT/r04_rccl_graph_probe.json'))['graph'])"; tail -5 $OUT/r04_rccl_graph_probe.err)
timeout 400 python bench.py --no-cpu-baseline 2> $OUT/r04_v5_bench.err | tail -1 > $OUT/r04_v5_bench.json
python - <<P
import json
j=json.load(open("$OUT/r04_v5_bench.json")); k=j["kernels_ms_per_step"]
print("r04_v5", round(j["value"],1), round(j["ms_per_step"],3), round(j["roofline"]["all_conv_tflops"],1), j.get("parity_ok"), round(j["gan_ms_per_cycle"],3))
P
