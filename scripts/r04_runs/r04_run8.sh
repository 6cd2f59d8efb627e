#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out
cd $R
for t in 8 16 32; do echo "MAXTS=$t"; M355_CHAMFER_MAXTS=$t python scripts/chamfer_rate.py 2>/dev/null | tail -5; done
for i in 1 2 3; do
  MASTER_PORT=$((29620+i)) HSA_ENABLE_IPC_MODE_LEGACY=0 timeout 300 python tests/_rccl_single_rank.py --graph > $OUT/r04_rccl_probe_$i.out 2> $OUT/r04_rccl_probe_$i.err
  echo "probe $i rc=$?"; grep '^{' $OUT/r04_rccl_probe_$i.out | python -c "
import json,sys
for l in sys.stdin: print(json.loads(l)['graph'])"
  grep -i "what()\|terminate\|abort" $OUT/r04_rccl_probe_$i.err | head -3
done
