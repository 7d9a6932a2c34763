#!/usr/bin/env python3
"""Best-of-rounds table of tools/ps_ko_run.sh's output: ms / step and the epilogue-GEMM layers' microseconds per library build."""
import collections, re, sys
best = collections.defaultdict(lambda: 1e9)
lay = collections.defaultdict(dict)
cur = None
for line in open(sys.argv[1]):
    if line.startswith("=="):
        _, _, rnd, lib, cfg = line.split()
        cur = (cfg, lib)
        continue
    m = re.search(r"([\d.]+) ms/step", line)
    if m:
        best[cur] = min(best[cur], float(m.group(1)))
        continue
    m = re.match(r"\s+(\d+) kind\s+(\d+).*?([\d.]+) us", line)
    if m:
        lay[cur][int(m.group(1))] = min(lay[cur].get(int(m.group(1)), 1e9), float(m.group(3)))
for k in sorted(best):
    print("%-13s %-18s %.3f ms/step   layers us: %s" % (k[0], k[1], best[k], " ".join("%.1f" % lay[k][i] for i in sorted(lay[k]))))
