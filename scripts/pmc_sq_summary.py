"""Merge the counters of several rocprofv3 --pmc passes (rocpd databases) per kernel instantiation and print the SQ
ratios the guide defines (MI355X_MICROARCH.md, rocprofv3 PMC slots): wait / issue-stall / active shares of the wave
cycles, LDS and VALU instruction counts, LDS bank conflicts.
    python scripts/pmc_sq_summary.py <filter substring> a.db [b.db ...]"""
import sqlite3
import sys

flt, dbs = sys.argv[1], sys.argv[2:]
tab = {}
for db in dbs:
    cur = sqlite3.connect(db).cursor()
    q = ("select kernel_name, grid_size, counter_name, avg(value), count(*) from counters_collection "
         "where kernel_name like ? group by kernel_name, grid_size, counter_name")
    for k, g, c, v, n in cur.execute(q, ("%" + flt + "%",)):
        tab.setdefault((k, g), {})[c] = v
print("# per (kernel instantiation, grid): counter averages per dispatch; shares are of SQ_WAVE_CYCLES (quad-cycles);")
print("# wait = SQ_WAIT_ANY (parked on s_waitcnt / barrier), stall = SQ_WAIT_INST_ANY (issue stall), active = SQ_ACTIVE_INST_ANY")
for (k, g), d in sorted(tab.items()):
    wc = d.get("SQ_WAVE_CYCLES")
    pct = lambda c: ("%5.1f%%" % (100.0 * d[c] / wc)) if (wc and c in d) else "   n/a"
    print(k[:110])
    print(f"    grid {g}  wait {pct('SQ_WAIT_ANY')}  stall {pct('SQ_WAIT_INST_ANY')}  active {pct('SQ_ACTIVE_INST_ANY')}  "
          f"lds-issue-stall {pct('SQ_WAIT_INST_LDS')}  lds-active {pct('SQ_ACTIVE_INST_LDS')}  valu-active {pct('SQ_ACTIVE_INST_VALU')}")
    print("    " + "  ".join(f"{c}={v:.4g}" for c, v in sorted(d.items())))
    req = d.get("TCC_REQ_sum", d.get("TCC_REQ"))
    if req:
        hit, miss = d.get("TCC_HIT_sum", d.get("TCC_HIT")), d.get("TCC_MISS_sum", d.get("TCC_MISS"))
        print(f"    L2 requests {req * 64 / 1e6:.1f} MB (64-byte requests)" + (f", hit {100.0 * hit / req:.1f}%" if hit else "") +
              (f", misses {miss * 64 / 1e6:.1f} MB" if miss else ""))
    if "SQ_LDS_BANK_CONFLICT" in d and d.get("SQ_ACTIVE_INST_LDS"):
        print(f"    LDS bank-conflict cycles / LDS active cycles = {d['SQ_LDS_BANK_CONFLICT'] / d['SQ_ACTIVE_INST_LDS']:.3f}")
