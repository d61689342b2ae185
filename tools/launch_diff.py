#!/usr/bin/env python3
"""Per-kernel-name difference of two per-launch tables (tools/launch_table.py): python tools/launch_diff.py a.tsv b.tsv"""
import csv
import sys
from collections import defaultdict


def load(f):
    d = defaultdict(lambda: [0, 0.0])
    for r in list(csv.reader(open(f), delimiter="\t"))[2:]:
        d[r[1]][0] += 1
        d[r[1]][1] += float(r[2])
    return d


a, b = load(sys.argv[1]), load(sys.argv[2])
rows = sorted(((b[k][1] - a[k][1], k, max(a[k][0], b[k][0]), a[k][1], b[k][1]) for k in set(a) | set(b)), reverse=True)
top = int(sys.argv[3]) if len(sys.argv) > 3 else 12
for d, k, n, x, y in rows[:top] + [(0, "...", 0, 0, 0)] + rows[-top:]:
    print("%+.3f %-36s n=%-3d %.3f -> %.3f" % (d, k, n, x, y))
print("total %.3f -> %.3f ms" % (sum(v[1] for v in a.values()), sum(v[1] for v in b.values())))
