#!/usr/bin/env python3
"""Summarise a rocprofv3 rocpd SQLite database (`--kernel-trace --stats`) into the
per-kernel table that is committed under profiles/: calls, total / average /
min / max duration, share of GPU time, LDS bytes, VGPR/AGPR counts, grid size."""
import sqlite3
import sys


def main(path, out):
    db = sqlite3.connect(path)
    rows = db.execute(
        "select name, count(*), sum(duration), avg(duration), min(duration), max(duration), "
        "max(lds_size), max(vgpr_count), max(accum_vgpr_count), max(sgpr_count), "
        "max(grid_x*grid_y*grid_z/(workgroup_x*workgroup_y*workgroup_z)) "
        "from kernels group by name order by sum(duration) desc").fetchall()
    total = sum(r[2] for r in rows)
    with open(out, "w") as f:
        f.write("# rocprofv3 --kernel-trace --stats summary (source: %s)\n" % path)
        f.write("# total GPU kernel time %.3f ms over %d dispatches\n" % (total / 1e6, sum(r[1] for r in rows)))
        f.write("kernel,calls,total_us,avg_us,min_us,max_us,pct,lds_bytes,vgpr,agpr,sgpr,max_workgroups\n")
        for r in rows:
            f.write('"%s",%d,%.1f,%.2f,%.2f,%.2f,%.2f,%d,%d,%d,%d,%d\n' %
                    (r[0], r[1], r[2] / 1e3, r[3] / 1e3, r[4] / 1e3, r[5] / 1e3, 100.0 * r[2] / total,
                     r[6] or 0, r[7] or 0, r[8] or 0, r[9] or 0, r[10] or 0))
        # the same, split by launch size for the kernels that run at several sizes (so that the
        # average of ONE layer can be compared with bench.py's HIP-event figure for it)
        f.write("# --- by (kernel, workgroups per launch), top 12 ---\n")
        f.write("kernel,workgroups,calls,avg_us,min_us,max_us\n")
        by = db.execute(
            "select name, grid_x*grid_y*grid_z/(workgroup_x*workgroup_y*workgroup_z) as wgs, count(*), "
            "avg(duration), min(duration), max(duration), sum(duration) from kernels "
            "group by name, wgs order by sum(duration) desc limit 12").fetchall()
        for r in by:
            f.write('"%s",%d,%d,%.2f,%.2f,%.2f\n' % (r[0], r[1], r[2], r[3] / 1e3, r[4] / 1e3, r[5] / 1e3))
    print("wrote", out)


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
