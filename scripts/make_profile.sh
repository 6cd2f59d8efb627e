#!/bin/bash
# Run ON THE GPU BOX (gpurun): the bench line, the rocprofv3 kernel-trace summary of the same command and the two
# PMC passes (FETCH_SIZE / WRITE_SIZE, each in its own run) -> gpurun_out/<tag>_*; copy what is to be judged into profiles/.
#   usage: scripts/make_profile.sh <tag> [bench args...]
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
TAG=${1:-r01}; shift || true
ARGS=${*:-"--steps 5 --warmup 2"}
OUT=$R/gpurun_out
mkdir -p $OUT
cd /tmp; export TMPDIR=/tmp
timeout 600 python $R/bench.py $ARGS 2> $OUT/${TAG}_bench.err | tail -1 > $OUT/${TAG}_bench.json
rm -rf /tmp/prof_$TAG
timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof_$TAG -o bench -- python $R/bench.py $ARGS --no-cpu-baseline --no-step-parity > $OUT/${TAG}_rocprof_bench.log 2>&1
python $R/scripts/rocpd_stats.py /tmp/prof_$TAG/bench_results.db $OUT/${TAG}_kernel_stats.csv
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pmc_$c
  timeout 600 rocprofv3 --kernel-trace --pmc $c -d /tmp/pmc_$c -o pmc -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-step-parity > $OUT/${TAG}_pmc_$c.log 2>&1
  python $R/scripts/rocpd_pmc.py /tmp/pmc_$c/pmc_results.db $OUT/${TAG}_pmc_$c.csv > /dev/null
done
python $R/scripts/pmc_traffic.py $OUT/${TAG}_pmc_FETCH_SIZE.csv $OUT/${TAG}_pmc_WRITE_SIZE.csv $OUT/${TAG}_kernel_stats.csv $OUT/${TAG}_pmc_traffic.json
cat $OUT/${TAG}_bench.json | cut -c1-600
