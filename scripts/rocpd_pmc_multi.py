"""per (kernel, grid) averages of all collected counters from a rocprofv3 rocpd db"""
import sqlite3, sys
cur = sqlite3.connect(sys.argv[1]).cursor()
rows = list(cur.execute("select kernel_name, grid_size, counter_name, avg(value), count(*) from counters_collection where kernel_name like '%m355::k_%' group by kernel_name, grid_size, counter_name"))
tab = {}
for k, g, c, v, n in rows:
    tab.setdefault((k[:70], g), {})[c] = v
for (k, g), d in sorted(tab.items()):
    print(k, g)
    print("   ", {c: round(v) for c, v in sorted(d.items())})
