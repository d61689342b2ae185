#!/usr/bin/env python3
"""Per-kernel averages of the PMC counters in a rocprofv3 rocpd database
(`rocprofv3 --kernel-trace --pmc ...`): one block per (kernel, grid size, LDS size)."""
import sqlite3
import sys


def main(path, pat="%", out=None):
    db = sqlite3.connect(path)
    rows = db.execute(
        "select p.name, k.grid_x / k.workgroup_x, p.counter_name, avg(p.counter_value), "
        "count(*), avg(p.duration), k.lds_size from pmc_events p join kernels k on k.dispatch_id = p.dispatch_id "
        "where p.name like ? group by p.name, k.grid_x, k.lds_size, p.counter_name "
        "order by p.name, k.grid_x, k.lds_size",
        (pat,)).fetchall()
    f = open(out, "w") if out else sys.stdout
    cur = None
    for name, blocks, cname, val, n, dur, lds in rows:
        key = (name, blocks, lds)
        if key != cur:
            cur = key
            f.write("%s  blocks_x=%d  lds_bytes=%d  dispatches=%d  avg_us=%.1f\n" % (name[:100], blocks, lds or 0, n, dur / 1e3))
        f.write("    %-26s %.5g\n" % (cname, val))


if __name__ == "__main__":
    main(*sys.argv[1:])
