#!/usr/bin/env python3
"""HBM-side traffic per launch of the kernels bench.py's roofline names, from two rocprofv3 PMC
passes over the bench command (tools/pmc_pass.sh ... FETCH_SIZE and ... WRITE_SIZE; the two
counters do not fit one pass on gfx950) -> profiles/<round>_pmc_traffic.json.

    python3 tools/pmc_traffic.py <fetch.db> <write.db> <time_batch> <out.json> "<command>"

The counters are in KiB; FETCH_SIZE is doubled as MI355X_MICROARCH.md (section HBM) prescribes
for wide coalesced reads on gfx950.  The file records a SHA-256 of the csrc/ sources it was
measured on; bench.py reports `traffic` only when that matches the tree it runs from.
"""
import hashlib
import json
import os
import sqlite3
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "jarvis-hybridnet_amd", "csrc")

# bench kernel name -> (SQL LIKE pattern of the device function, which (grid, lds) group)
SPECS = {
    "conv3d_k3s1wino_46x46@32": ("%conv3d_wino_pw_kernel%", "most_dispatches"),
    "conv3d_k3s1wino_92x92@16": ("%conv3d_wino_pw_kernel%", "fewest_dispatches"),
    "reproject_gather": ("%repro_cube_kernel%", "largest_grid"),
    "bifpn_node_56x56@64": ("%bifpn_rows_kernel<2,%", "largest_grid"),
    "preprocess_resize": ("%preprocess_resize%", "largest_grid"),
}


def csrc_sha256():
    """SHA-256 over the names and contents of every source file that goes into libjarvis_hip.so."""
    h = hashlib.sha256()
    for name in sorted(os.listdir(CSRC)):
        if name.endswith((".hip", ".h")) or name == "Makefile":
            h.update(name.encode() + b"\0")
            h.update(open(os.path.join(CSRC, name), "rb").read())
    return h.hexdigest()


def groups(db_path, pattern, counter):
    db = sqlite3.connect(db_path)
    return db.execute(
        "select p.name, k.grid_x / k.workgroup_x, k.lds_size, avg(p.counter_value), count(*), avg(p.duration) "
        "from pmc_events p join kernels k on k.dispatch_id = p.dispatch_id "
        "where p.name like ? and p.counter_name = ? group by p.name, k.grid_x, k.lds_size",
        (pattern, counter)).fetchall()


def pick(rows, rule):
    if not rows:
        return None
    if rule == "most_dispatches":
        return max(rows, key=lambda r: r[4])
    if rule == "fewest_dispatches":
        return min(rows, key=lambda r: r[4])
    return max(rows, key=lambda r: r[1])


def main(fetch_db, write_db, time_batch, out, command):
    res = {"_comment": "HBM-side bytes per launch from rocprofv3 PMC passes (separate --pmc FETCH_SIZE and --pmc "
                       "WRITE_SIZE runs of the command below; counters in KiB; FETCH_SIZE doubled as "
                       "MI355X_MICROARCH.md prescribes for wide coalesced reads on gfx950). Written by "
                       "tools/pmc_traffic.py.",
           "command": command, "time_batch": int(time_batch), "csrc_sha256": csrc_sha256()}
    for bench_name, (pattern, rule) in SPECS.items():
        f = pick(groups(fetch_db, pattern, "FETCH_SIZE"), rule)
        w = pick(groups(write_db, pattern, "WRITE_SIZE"), rule)
        if f is None or w is None:
            print("pmc_traffic: no dispatches match %s" % pattern, file=sys.stderr)
            continue
        if (f[1], f[2]) != (w[1], w[2]):
            print("pmc_traffic: %s: fetch / write passes picked different launch groups %s vs %s" %
                  (bench_name, f[1:3], w[1:3]), file=sys.stderr)
            continue
        res[bench_name] = {
            "kernel": "%s (blocks %d, lds %d)" % (f[0][:80], f[1], f[2] or 0),
            "dispatches_per_pass": [f[4], w[4]], "avg_us": [f[5] / 1e3, w[5] / 1e3],
            "fetch_size_kib": f[3], "write_size_kib": w[3],
            "hbm_bytes_per_launch": (2.0 * f[3] + w[3]) * 1024.0,
        }
    with open(out, "w") as fh:
        json.dump(res, fh, indent=1)
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    if len(sys.argv) == 2 and sys.argv[1] == "--sha":
        print(csrc_sha256())
    else:
        main(*sys.argv[1:6])
