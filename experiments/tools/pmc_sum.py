"""Sums rocprofv3 counter CSVs per kernel family: python tools/pmc_sum.py <dir>
(<dir>/<pass>/**/p_counter_collection.csv, <dir>/trace/**/t_kernel_stats.csv)."""
import csv
import glob
import os
import re
import sys
from collections import defaultdict

root = sys.argv[1]


def family(name):
    m = re.match(r"(?:void )?(?:hexl_amd::)?(\w+)<\s*\(?(?:bool\)?)?\s*(\w+)", name)
    base = name.split("<")[0].split("::")[-1]
    if m:
        d = m.group(2)
        d = {"true": "fwd", "false": "inv", "1": "fwd", "0": "inv"}.get(d, d)
        return f"{base}<{d}>"
    return base


for f in glob.glob(os.path.join(root, "trace", "**", "*kernel_stats.csv"), recursive=True):
    print("== kernel stats", f)
    for row in csv.DictReader(open(f)):
        print("  %-60s calls %5s avg %10.1f us total %10.1f us" % (
            family(row["Name"])[:60], row["Calls"], float(row["AverageNs"]) / 1e3,
            float(row["TotalDurationNs"]) / 1e3))

for f in sorted(glob.glob(os.path.join(root, "*", "**", "*counter_collection.csv"), recursive=True)):
    acc = defaultdict(lambda: defaultdict(float))
    calls = defaultdict(set)
    for row in csv.DictReader(open(f)):
        k = family(row["Kernel_Name"])
        acc[k][row["Counter_Name"]] += float(row["Counter_Value"])
        calls[k].add(row["Dispatch_Id"])
    print("== counters", os.path.relpath(f, root))
    for k in acc:
        n = len(calls[k])
        print("  %-50s dispatches %4d  " % (k[:50], n) +
              "  ".join("%s/launch=%.4g" % (c, v / n) for c, v in acc[k].items()))
