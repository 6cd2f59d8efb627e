"""Kernel-time summary (the --stats table) from a rocprofv3 rocpd database: python scripts/rocpd_stats.py in.db out.csv
rocprofv3 7.2 writes only the SQLite (rocpd) output unless --output-format csv is given; this reproduces the
kernel_stats.csv columns from the `kernels` view so that the summary can be committed under profiles/."""
import csv
import sqlite3
import statistics
import sys


def main(db_path, out_path):
    cur = sqlite3.connect(db_path).cursor()
    per = {}
    for name, s, e in cur.execute("select name, start, end from kernels"):
        per.setdefault(name, []).append(e - s)
    tot = sum(sum(v) for v in per.values())
    with open(out_path, "w", newline="") as f:
        w = csv.writer(f, quoting=csv.QUOTE_NONNUMERIC)
        w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs", "StdDev"])
        for name, v in sorted(per.items(), key=lambda kv: -sum(kv[1])):
            w.writerow([name, len(v), sum(v), sum(v) / len(v), 100.0 * sum(v) / tot, min(v), max(v),
                        statistics.pstdev(v) if len(v) > 1 else 0.0])
    print(f"{len(per)} kernels, {tot / 1e6:.3f} ms total -> {out_path}")


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
