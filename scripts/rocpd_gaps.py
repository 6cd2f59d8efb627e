"""Busy / idle anatomy of a rocprofv3 kernel trace (rocpd SQLite): python scripts/rocpd_gaps.py in.db [tail_fraction]
Over the last `tail_fraction` (default 0.5) of the LAUNCHES (not of the time: a capture pass is a long pause without kernels) -- the
timed replays, past warm-up and capture -- prints the wall span, the
UNION of kernel intervals (two streams overlap), the idle time between kernels, the number of launches, and the kernels by total time."""
import sqlite3
import sys


def main(db_path, frac=0.5):
    cur = sqlite3.connect(db_path).cursor()
    rows = sorted((s, e, name) for name, s, e in cur.execute("select name, start, end from kernels"))
    rows = rows[int(len(rows) * (1.0 - frac)):]
    span = rows[-1][1] - rows[0][0]
    busy, cur_s, cur_e = 0, rows[0][0], rows[0][1]
    gaps = []
    for s, e, _ in rows[1:]:
        if s > cur_e:
            busy += cur_e - cur_s
            gaps.append(s - cur_e)
            cur_s, cur_e = s, e
        else:
            cur_e = max(cur_e, e)
    busy += cur_e - cur_s
    ksum = sum(e - s for s, e, _ in rows)
    print(f"launches {len(rows)}  span {span / 1e6:.3f} ms  busy(union) {busy / 1e6:.3f} ms  idle {sum(gaps) / 1e6:.3f} ms in {len(gaps)} gaps "
          f"(median {sorted(gaps)[len(gaps) // 2] / 1e3:.2f} us)  kernel sum {ksum / 1e6:.3f} ms")
    per = {}
    for s, e, name in rows:
        c, t = per.get(name, (0, 0))
        per[name] = (c + 1, t + e - s)
    for name, (c, t) in sorted(per.items(), key=lambda kv: -kv[1][1])[:70]:
        print(f"{t / 1e6:9.3f} ms {100.0 * t / ksum:5.1f}%  x{c:5d}  avg {t / c / 1e3:8.2f} us  {name[:150]}")


if __name__ == "__main__":
    main(sys.argv[1], float(sys.argv[2]) if len(sys.argv) > 2 else 0.5)
