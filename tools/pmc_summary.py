#!/usr/bin/env python3
"""Summarise rocprofv3 CSV output directories: per kernel name, average counter values per
dispatch (--pmc passes) and average duration (--kernel-trace passes).
    python tools/pmc_summary.py DIR [DIR ...] [--match substr]
FETCH_SIZE / WRITE_SIZE are reported in KiB by rocprofv3; on gfx950 FETCH_SIZE counts 64 B per 128-B
request for wide coalesced streams (MI355X_MICROARCH.md, HBM section) -- the summary prints the raw value
and the x2-corrected bytes."""
import csv
import glob
import os
import re
import sys
from collections import defaultdict


def short(name):
    name = name.replace("(anonymous namespace)::", "").replace("void hsefr::", "")
    name = re.sub(r"\(.*$", "", name)
    return name[:70]


def main():
    dirs = [a for a in sys.argv[1:] if not a.startswith("--")]
    match = None
    if "--match" in sys.argv:
        match = sys.argv[sys.argv.index("--match") + 1]
        dirs = [d for d in dirs if d != match]
    counters = defaultdict(lambda: defaultdict(list))   # kernel -> counter -> [values per dispatch]
    durs = defaultdict(list)
    pass_durs = defaultdict(lambda: defaultdict(list))  # counter -> kernel -> [us per dispatch IN THE PASS that collected the counter]
    for d in dirs:
        for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
            per_dispatch = defaultdict(float)
            names = {}
            stamps = {}
            for row in csv.DictReader(open(f)):
                key = (f, row["Dispatch_Id"], row["Counter_Name"])
                per_dispatch[key] += float(row["Counter_Value"])
                names[(f, row["Dispatch_Id"])] = row["Kernel_Name"]
                if row.get("Start_Timestamp") and row.get("End_Timestamp"):
                    stamps[(f, row["Dispatch_Id"])] = (int(row["End_Timestamp"]) - int(row["Start_Timestamp"])) / 1e3
            for (ff, did, cn), v in per_dispatch.items():
                counters[short(names[(ff, did)])][cn].append(v)
                if (ff, did) in stamps:
                    pass_durs[cn][short(names[(ff, did)])].append(stamps[(ff, did)])
        for f in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
            for row in csv.DictReader(open(f)):
                durs[short(row["Kernel_Name"])].append((int(row["End_Timestamp"]) - int(row["Start_Timestamp"])) / 1e3)
    kernels = sorted(set(counters) | set(durs))
    for k in kernels:
        if match and match not in k:
            continue
        line = "%-70s" % k
        if durs.get(k):
            v = sorted(durs[k])
            line += " n=%-5d avg_us=%9.2f med_us=%9.2f" % (len(v), sum(v) / len(v), v[len(v) // 2])
        print(line)
        for cn, vals in sorted(counters.get(k, {}).items()):
            avg = sum(vals) / len(vals)
            extra = ""
            if cn == "FETCH_SIZE":
                extra = "  (KiB; x2-corrected = %.1f MB)" % (avg * 1024 * 2 / 1e6)
            if cn == "WRITE_SIZE":
                extra = "  (KiB; = %.1f MB)" % (avg * 1024 / 1e6)
            pd = pass_durs.get(cn, {}).get(k)
            if pd:
                extra += "  pass_us=%.2f" % (sum(pd) / len(pd))
            print("    %-34s n=%-5d avg=%16.1f%s" % (cn, len(vals), avg, extra))


if __name__ == "__main__":
    main()
