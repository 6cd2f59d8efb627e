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
    big = []   # (gap, kernel that ended last before it, kernel that starts after it)
    cur_e, last = rows[0][1], rows[0][2]
    for s, e, name in rows[1:]:
        if s > cur_e:
            big.append((s - cur_e, last, name))
        if e >= cur_e:
            cur_e, last = e, name
    print(f"launches {len(rows)}  span {span / 1e6:.3f} ms  busy(union) {busy / 1e6:.3f} ms  idle {sum(gaps) / 1e6:.3f} ms in {len(gaps)} gaps "
          f"(median {sorted(gaps)[len(gaps) // 2] / 1e3:.2f} us)  kernel sum {ksum / 1e6:.3f} ms")
    edges = (2e3, 5e3, 10e3, 20e3, 50e3, 100e3, 1e12)
    hist = [[0, 0] for _ in edges]
    for g in gaps:
        k = next(i for i, x in enumerate(edges) if g < x)
        hist[k][0] += 1
        hist[k][1] += g
    print("gap histogram (us: count, total ms): " + "  ".join(
        f"<{int(x / 1e3) if x < 1e12 else 'inf'}: {c}, {t / 1e6:.2f}" for x, (c, t) in zip(edges, hist)))
    short = lambda n: n.replace("void ", "").replace("m355::", "").split("(")[0][:60]
    pairs = {}
    for g, a, b in big:
        if g >= 20e3:
            c, t = pairs.get((short(a), short(b)), (0, 0))
            pairs[(short(a), short(b))] = (c + 1, t + g)
    print("gaps >= 20 us by (kernel before -> kernel after):")
    for (a, b), (c, t) in sorted(pairs.items(), key=lambda kv: -kv[1][1])[:25]:
        print(f"   {t / 1e6:7.3f} ms  x{c:4d}  avg {t / c / 1e3:7.1f} us   {a}  ->  {b}")
    per = {}
    for s, e, name in rows:
        c, t = per.get(name, (0, 0))
        per[name] = (c + 1, t + e - s)
    for name, (c, t) in sorted(per.items(), key=lambda kv: -kv[1][1])[:70]:
        print(f"{t / 1e6:9.3f} ms {100.0 * t / ksum:5.1f}%  x{c:5d}  avg {t / c / 1e3:8.2f} us  {name[:150]}")


if __name__ == "__main__":
    main(sys.argv[1], float(sys.argv[2]) if len(sys.argv) > 2 else 0.5)
