#!/bin/bash
# RCCL collectives inside the captured cycle: how often does the single-rank probe die, and does (a) letting ProcessGroupNCCL's
# watchdog drain before the capture or (b) a thread-local capture mode change that?
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
probe() {   # label, env...
    local label=$1; shift
    for i in 1 2 3 4 5; do
        env "$@" MASTER_PORT=$((29700 + RANDOM % 200)) timeout 240 python tests/_rccl_single_rank.py --graph > /tmp/p.out 2> /tmp/p.err
        rc=$?
        res=$(grep '^{' /tmp/p.out | tail -1 | python -c "import sys, json; l = sys.stdin.read().strip(); print(json.loads(l)['graph'] if l else None)")
        echo "$label trial $i rc=$rc $res"
        if [ $rc -ne 0 ]; then grep -v "amdgpu.ids\|Librccl" /tmp/p.err | tail -12 | cut -c1-300; fi
    done
}
probe "global           " M355_X=0
probe "global+settle500 " M355_CAPTURE_SETTLE_MS=500
probe "thread_local     " M355_CAPTURE_MODE=thread_local
