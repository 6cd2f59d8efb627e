#!/bin/bash
# RCCL collectives inside the captured cycle after the capture went thread-local: 16 stand-alone probes
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
ok=0
for i in $(seq 1 16); do
  MASTER_PORT=$((29700 + RANDOM % 200)) timeout 240 python tests/_rccl_single_rank.py --graph > /tmp/p.out 2> /tmp/p.err
  rc=$?
  res=$(grep '^{' /tmp/p.out | tail -1 | python -c "import sys, json; l = sys.stdin.read().strip(); print(json.loads(l)['graph'] if l else None)")
  echo "trial $i rc=$rc $res"
  if [ $rc -ne 0 ]; then grep "what()\|Error" /tmp/p.err | head -3 | cut -c1-300; fi
done
