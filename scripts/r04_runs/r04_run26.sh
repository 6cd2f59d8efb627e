#!/bin/bash
# the epilogue credit of k_conv_halo's counted wait: old (one step too many for the L = 1 variants: a race), fixed, none.
# (1) determinism: 16 asynchronous traces of the fixed build; (2) same-box bench of the three builds, alternating
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/trace3
L=$GRAFT_REPO_ROOT/2dimageto3dmodel_amd/lib
echo "--- fixed credit: async traces"
for p in 1 2 3 4 5 6 7 8 9 10 11 12 13 14 15 16; do
  TRACE_SYNC=0 timeout 300 python scripts/r04_det_trace.py 16 256 2 gpurun_out/trace3/f$p.txt 2>&1 | grep "^trace\|Error" | cut -c1-120
done
echo "--- bench"
for rep in 1 2; do
for v in libm355_oldcredit.so libm355.so libm355_nocredit.so; do
  M355_LIB=$L/$v timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > /tmp/b.json 2> /tmp/b.err
  python - "$v" <<'P'
import json, sys
d = json.load(open("/tmp/b.json")); k = d["kernels_ms_per_step"]
print(f"{sys.argv[1]:24s} {d['value']:8.1f} samples/s {d['ms_per_step']:7.3f} ms  gan {d['gan_ms_per_cycle']:7.3f}  k_conv_halo {k['k_conv_halo']:.3f}  k_wgrad_halo {k['k_wgrad_halo']:.3f}  parity {d['parity_ok']}")
P
done
done
