#!/usr/bin/env python3
"""Condenses rocprofv3 CSV output into the small summaries committed under profiles/.

  summarize_rocprof.py stats <dir> <out.csv>      per-kernel count / total / mean duration
  summarize_rocprof.py pmc <dir> <out.json>       per-kernel mean of every collected counter
"""
import csv
import glob
import json
import os
import sys
from collections import defaultdict


def short(name):
    # strip template arguments' noise but keep what tells the instantiations apart
    return name.replace("tsp::", "").split("(")[0][:110]


def stats(d, out):
    rows = []
    for f in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
        with open(f) as fh:
            rows += list(csv.DictReader(fh))
    agg = defaultdict(lambda: [0, 0.0, 1e30, 0.0])
    for r in rows:
        dur = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3  # us
        a = agg[short(r["Kernel_Name"])]
        a[0] += 1
        a[1] += dur
        a[2] = min(a[2], dur)
        a[3] = max(a[3], dur)
    tot = sum(a[1] for a in agg.values()) or 1.0
    with open(out, "w") as fh:
        w = csv.writer(fh)
        w.writerow(["kernel", "calls", "total_us", "mean_us", "min_us", "max_us", "pct"])
        for k, a in sorted(agg.items(), key=lambda kv: -kv[1][1]):
            w.writerow([k, a[0], "%.1f" % a[1], "%.3f" % (a[1] / a[0]), "%.3f" % a[2], "%.3f" % a[3],
                        "%.2f" % (100 * a[1] / tot)])
    print(open(out).read())


def pmc(d, out):
    agg = defaultdict(lambda: defaultdict(lambda: [0, 0.0]))
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        with open(f) as fh:
            for r in csv.DictReader(fh):
                a = agg[short(r["Kernel_Name"])][r["Counter_Name"]]
                a[0] += 1
                a[1] += float(r["Counter_Value"])
    res = {k: {c: {"dispatches": a[0], "mean": a[1] / a[0]} for c, a in v.items()} for k, v in agg.items()}
    with open(out, "w") as fh:
        json.dump(res, fh, indent=1, sort_keys=True)
    print(json.dumps(res, indent=1, sort_keys=True)[:4000])


if __name__ == "__main__":
    {"stats": stats, "pmc": pmc}[sys.argv[1]](sys.argv[2], sys.argv[3])
