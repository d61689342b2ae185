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

# bench kernel name -> list of (SQL LIKE pattern of the device function, which launch group of that function).
# A bench name that covers several device functions / shapes (all InstanceNorm passes, both fused stems, the two
# node forms of a level) gets the dispatch-weighted AVERAGE per launch over its parts.
SPECS = {
    "conv3d_k3s1wino_46x46@32": [("%conv3d_wino_pw_kernel%", "most_dispatches")],
    "conv3d_k3s1wino_92x92@16": [("%conv3d_wino_pw_kernel%", "fewest_dispatches")],
    "reproject_gather": [("%repro_cube_kernel%", "largest_grid")],
    "bifpn_node_56x56@64": [("%bifpn_rows_kernel<56, 2, 1, 0, 2, true>%", "largest_grid")],
    "bifpn_node_56x64@64": [("%bifpn_rows_kernel<56, 3, 1, 2, 0, false>%", "largest_grid")],
    "bifpn_node_56x56@32": [("%bifpn_rows_kernel<56, 2, 1, 0, 2, false>%", "largest_grid"),
                            ("%bifpn_rows_kernel<56, 3, 0, 0, 2, false>%", "largest_grid")],
    "norm_apply": [("%norm_apply_kernel%", "all")],
    "conv2d_k4s2T_64x23@128": [("%deconv4_fused_kernel%", "largest_grid")],
    "stem_conv_k3s2_3x16@128": [("%stem_conv_kernel<1,%", "largest_grid"), ("%stem_conv_kernel<2,%", "largest_grid")],
    "conv3d_k3s2_23x46@32": [("%conv_mfma_kernel<3, 3, 2,%", "largest_grid")],
    "conv2d_k3s1_16x16@128": [("%conv_mfma_kernel<2, 3, 1, 1, 16, 16, 1, 2>%", "largest_grid")],
    "conv2d_k5s2_16x96@32": [("%conv_mfma_kernel<2, 5, 2,%", "largest_grid")],
    "deconv_k4s2T_c1": [("%deconv_c1_kernel%", "largest_grid")],
}


def workload_of(command):
    """(--config, --model-size) of a bench command line (bench.py's defaults when absent)."""
    words = command.split()

    def opt(name, default):
        return words[words.index(name) + 1] if name in words and words.index(name) + 1 < len(words) else default
    return opt("--config", "cfg3"), opt("--model-size", "small")


def csrc_sha256():
    """SHA-256 over the names and contents of every source file that goes into libjarvis_hip.so."""
    h = hashlib.sha256()
    for name in sorted(os.listdir(CSRC)):
        if name.endswith((".hip", ".h")) or name == "Makefile":
            h.update(name.encode() + b"\0")
            h.update(open(os.path.join(CSRC, name), "rb").read())
    return h.hexdigest()


def groups(db_path, pattern, counter):
    """(function, workgroups, lds, avg counter, dispatches, avg duration) per launch group of the functions that
    match `pattern`."""
    db = sqlite3.connect(db_path)
    return db.execute(
        "select p.name, k.grid_x / k.workgroup_x, k.lds_size, avg(p.counter_value), count(*), avg(p.duration) "
        "from pmc_events p join kernels k on k.dispatch_id = p.dispatch_id "
        "where p.name like ? and p.counter_name = ? group by p.name, k.grid_x, k.lds_size",
        (pattern, counter)).fetchall()


def pick(rows, rule):
    """The launch group(s) a rule selects: a list of rows."""
    if not rows:
        return []
    if rule == "all":
        return sorted(rows, key=lambda r: (r[0], r[1], r[2]))
    if rule == "most_dispatches":
        return [max(rows, key=lambda r: r[4])]
    if rule == "fewest_dispatches":
        return [min(rows, key=lambda r: r[4])]
    return [max(rows, key=lambda r: r[1])]


def main(fetch_db, write_db, time_batch, out, command):
    res = {"_comment": "HBM-side bytes per launch from rocprofv3 PMC passes (separate --pmc FETCH_SIZE and --pmc "
                       "WRITE_SIZE runs of the command below; counters in KiB; FETCH_SIZE doubled as "
                       "MI355X_MICROARCH.md prescribes for wide coalesced reads on gfx950). A bench name that covers "
                       "several device functions / shapes carries the dispatch-weighted average per launch. Written by "
                       "tools/pmc_traffic.py.",
           "command": command, "time_batch": int(time_batch), "csrc_sha256": csrc_sha256(),
           # the workload the byte counts belong to: bench.py reports them only for this config and model size
           # (several bench kernel names -- norm_apply, reproject_gather, deconv_k4s2T_c1 -- carry no shape)
           "config": workload_of(command)[0], "model_size": workload_of(command)[1]}
    for bench_name, parts in SPECS.items():
        f, w = [], []
        for pattern, rule in parts:
            f += pick(groups(fetch_db, pattern, "FETCH_SIZE"), rule)
            w += pick(groups(write_db, pattern, "WRITE_SIZE"), rule)
        if not f or not w:
            print("pmc_traffic: no dispatches match %s" % bench_name, file=sys.stderr)
            continue
        if [(r[0], r[1], r[2]) for r in f] != [(r[0], r[1], r[2]) for r in w]:
            print("pmc_traffic: %s: fetch / write passes picked different launch groups" % bench_name, file=sys.stderr)
            continue
        nf, nw = sum(r[4] for r in f), sum(r[4] for r in w)
        fetch = sum(r[3] * r[4] for r in f) / nf
        write = sum(r[3] * r[4] for r in w) / nw
        res[bench_name] = {
            "kernel": "; ".join("%s (blocks %d, lds %d)" % (r[0][:80], r[1], r[2] or 0) for r in f[:4]) +
                      (" ... %d launch groups" % len(f) if len(f) > 4 else ""),
            "dispatches_per_pass": [nf, nw],
            "avg_us": [sum(r[5] * r[4] for r in f) / nf / 1e3, sum(r[5] * r[4] for r in w) / nw / 1e3],
            "fetch_size_kib": fetch, "write_size_kib": write,
            "hbm_bytes_per_launch": (2.0 * fetch + write) * 1024.0,
        }
    with open(out, "w") as fh:
        json.dump(res, fh, indent=1)
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    if len(sys.argv) == 2 and sys.argv[1] == "--sha":
        print(csrc_sha256())
    else:
        main(*sys.argv[1:6])
