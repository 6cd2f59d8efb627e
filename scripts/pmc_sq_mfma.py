"""SQ wave-state shares and MFMA-pipe busy share per conv kernel instantiation from ONE rocprofv3 pass
(--kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES
SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS): python scripts/pmc_sq_mfma.py pmc_results.db > profiles/r05_pmc_sq_conv.txt
MI355X_MICROARCH.md "rocprofv3 PMC slots": SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* count quad-cycles, SQ_VALU_MFMA_BUSY_CYCLES
counts cycles (32 per v_mfma_f32_32x32x16_bf16).  mfma-busy = MFMA_BUSY / (4 SIMDs x 256 CUs x kernel duration x f): the share of
the chip's MFMA issue time the kernel used, at the NOMINAL 2.4 GHz (the socket's power cap holds the conv kernels near 1.5-1.9 GHz,
DESIGN.md 5 'Round 3': against the effective clock the share is proportionally higher)."""
import sqlite3, sys
db = sys.argv[1]
cur = sqlite3.connect(db).cursor()
dur = {}
rows = cur.execute("select * from kernels")
cols = [c[0] for c in rows.description]
print("# kernels view columns:", cols)
ix = {c: i for i, c in enumerate(cols)}
def grid_of(r):
    for trip in (("grid_size",), ("grid_size_x", "grid_size_y", "grid_size_z"), ("grid_x", "grid_y", "grid_z")):
        if all(t in ix for t in trip):
            g = 1
            for t in trip:
                g *= int(r[ix[t]] or 1)
            return g
    return 0
byname = {}
for r in rows:
    name, dt = r[ix["name"]], r[ix["end"]] - r[ix["start"]]
    dur.setdefault((name, grid_of(r)), []).append(dt)
    byname.setdefault(name, []).append(dt)
tab = {}
for k, g, c, v in cur.execute("select kernel_name, grid_size, counter_name, avg(value) from counters_collection group by kernel_name, grid_size, counter_name"):
    tab.setdefault((k, g), {})[c] = v
print("# " + __doc__.replace("\n", "\n# "))
for (k, g), d in sorted(tab.items(), key=lambda kv: -sum(dur.get(kv[0], [0]))):
    if "m355::k_conv" not in k and "m355::k_wgrad" not in k and "k_head5" not in k:
        continue
    wc = d.get("SQ_WAVE_CYCLES")
    pct = lambda c: ("%5.1f%%" % (100.0 * d[c] / wc)) if (wc and c in d) else "   n/a"
    t = dur.get((k, g)) or dur.get((k, int(g))) or byname.get(k)   # (falls back to the instantiation's mean over all its shapes)
    avg_us = sum(t) / len(t) / 1e3 if t else float("nan")
    mf = d.get("SQ_VALU_MFMA_BUSY_CYCLES")
    busy = 100.0 * mf / (4 * 256 * avg_us * 2400.0) if (mf and t) else float("nan")
    print(k[:150])
    print(f"    grid {g}  avg {avg_us:8.1f} us  mfma-busy {busy:5.1f}% of the chip at 2.4 GHz  ({(mf or 0) / 32:.4g} MFMA 32x32x16-equivalents per launch)")
    print(f"    wait {pct('SQ_WAIT_ANY')}  issue-stall {pct('SQ_WAIT_INST_ANY')}  active {pct('SQ_ACTIVE_INST_ANY')}  valu-active {pct('SQ_ACTIVE_INST_VALU')}  lds-active {pct('SQ_ACTIVE_INST_LDS')}")
    print("    " + "  ".join(f"{c}={v:.4g}" for c, v in sorted(d.items())))
