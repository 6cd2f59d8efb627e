"""HBM traffic per launch of each kernel family from the two PMC passes (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, KiB
per dispatch):  hbm_bytes = (2 * FETCH_SIZE + WRITE_SIZE) * 1024.  The factor 2 is the gfx950 correction of
MI355X_MICROARCH.md (FETCH_SIZE counts 64 B per 128-B request of a wide coalesced read); WRITE_SIZE is uncalibrated.
Families aggregate the template instantiations of one kernel (the bench's HIP-event timers do the same).
    python scripts/pmc_traffic.py fetch.csv write.csv kernel_stats.csv out.json"""
import csv
import json
import re
import sys


def family(name):
    m = re.search(r"m355::(k_\w+)", name)
    return m.group(1) if m else None


def load(path):
    per = {}
    with open(path) as f:
        for row in csv.DictReader(f):
            fam = family(row["Kernel"])
            if fam:
                n, s = per.get(fam, (0, 0.0))
                per[fam] = (n + int(float(row["Dispatches"])), s + float(row["SumValue"]))
    return per


def main(fetch_csv, write_csv, stats_csv, out_json):
    fe, wr = load(fetch_csv), load(write_csv)
    dur = {}
    with open(stats_csv) as f:
        for row in csv.DictReader(f):
            fam = family(row["Name"])
            if fam:
                n, t = dur.get(fam, (0, 0.0))
                dur[fam] = (n + int(float(row["Calls"])), t + float(row["TotalDurationNs"]))
    out = {}
    for fam in sorted(set(fe) | set(wr)):
        nf, sf = fe.get(fam, (0, 0.0))
        nw, sw = wr.get(fam, (0, 0.0))
        e = {"fetch_kib_per_launch": sf / nf if nf else None, "write_kib_per_launch": sw / nw if nw else None}
        if nf and nw:
            e["hbm_bytes_per_launch"] = (2.0 * sf / nf + sw / nw) * 1024.0
        if fam in dur:
            e["rocprof_avg_us"] = dur[fam][1] / dur[fam][0] / 1e3
            e["rocprof_calls"] = dur[fam][0]
        out[fam] = e
    # the sources these counters were collected on (2dimageto3dmodel_amd/build.py source_hash): bench.py reports the traffic only for this tree
    import importlib.util
    import os
    spec = importlib.util.spec_from_file_location("m355_build", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))),
                                                                             "2dimageto3dmodel_amd", "build.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    out["_csrc_sha256"] = mod.source_hash()
    json.dump(out, open(out_json, "w"), indent=1, sort_keys=True)
    print(json.dumps(out.get("k_conv_glds", {})))


if __name__ == "__main__":
    main(*sys.argv[1:5])
