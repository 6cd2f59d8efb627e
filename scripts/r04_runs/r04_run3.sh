#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out
mkdir -p $OUT
cd $R
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -25 > $OUT/r04_run3_pytest.txt
cat $OUT/r04_run3_pytest.txt
python scripts/thin_rate.py 0.8 > $OUT/r04_thin_rate_planar.txt 2>&1
M355_WGC8_V1=1 python scripts/thin_rate.py 0.8 "D.conv1 8->64 5x5 256^2 wgrad" > $OUT/r04_thin_rate_v1.txt 2>&1
cat $OUT/r04_thin_rate_planar.txt $OUT/r04_thin_rate_v1.txt
for s in 0 1 0 1; do
  M355_STREAMS=$s timeout 300 python bench.py --no-cpu-baseline --batch 16 --graph --workload gan 2> $OUT/r04_b16_s$s.err | tail -1 > $OUT/r04_b16_s$s.json
  python -c "
import json; j=json.load(open('$OUT/r04_b16_s$s.json')); print('batch16 graph streams=$s', round(j['ms_per_step'],3), j['config'].get('gan_streams'))"
done
