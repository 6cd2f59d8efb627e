"""Per-kernel average of one PMC counter from a rocprofv3 rocpd database (run on the GPU box: the databases of a
full bench run exceed what gpurun merges back):  python scripts/rocpd_pmc.py pmc_results.db out.csv"""
import csv
import sqlite3
import sys


def main(db_path, out_path):
    cur = sqlite3.connect(db_path).cursor()
    cols = [d[1] for d in cur.execute("pragma table_info(counters_collection)")]
    print("counters_collection columns:", cols)
    name_col = "kernel_name" if "kernel_name" in cols else "name"
    q = (f"select {name_col}, counter_name, count(*), avg(value), sum(value) from counters_collection "
         f"group by {name_col}, counter_name order by 5 desc")
    rows = list(cur.execute(q))
    with open(out_path, "w", newline="") as f:
        w = csv.writer(f, quoting=csv.QUOTE_NONNUMERIC)
        w.writerow(["Kernel", "Counter", "Dispatches", "AvgValue", "SumValue"])
        for r in rows:
            w.writerow(list(r))
    for r in rows[:12]:
        print(r[0][:70], r[1], r[2], "%.1f" % r[3])


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
