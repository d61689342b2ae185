#!/usr/bin/env python3
"""Per-kernel averages of the PMC counters in a rocprofv3 rocpd database
(`rocprofv3 --kernel-trace --pmc ...`): one block per (kernel, grid size)."""
import sqlite3
import sys


def main(path, pat="%", out=None):
    db = sqlite3.connect(path)
    rows = db.execute(
        "select p.name, k.grid_x / k.workgroup_x, p.counter_name, avg(p.counter_value), "
        "count(*), avg(p.duration) from pmc_events p join kernels k on k.dispatch_id = p.dispatch_id "
        "where p.name like ? group by p.name, k.grid_x, p.counter_name order by p.name, k.grid_x",
        (pat,)).fetchall()
    f = open(out, "w") if out else sys.stdout
    cur = None
    for name, blocks, cname, val, n, dur in rows:
        key = (name, blocks)
        if key != cur:
            cur = key
            f.write("%s  blocks_x=%d  dispatches=%d  avg_us=%.1f\n" % (name[:100], blocks, n, dur / 1e3))
        f.write("    %-26s %.5g\n" % (cname, val))


if __name__ == "__main__":
    main(*sys.argv[1:])
