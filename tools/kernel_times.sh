#!/bin/bash
# Per-kernel average durations of one single-stream bench run (rocprofv3 --kernel-trace), filtered by a name
# pattern:   bash tools/kernel_times.sh bifpn [extra bench args]
pat=${1:-bifpn}; shift
root=$(pwd)
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/kt_run
rocprofv3 --kernel-trace --stats -d /tmp/kt_run -o run -- python3 $root/bench.py --steps 10 --warmup 3 --streams 1 \
  --no-reduced-precision --no-cpu-baseline --no-uint8 --no-secondary "$@" > /tmp/kt_run.log 2>&1 || { tail -5 /tmp/kt_run.log; exit 1; }
db=$(find /tmp/kt_run -name "*.db" | head -1)
python3 - $db "$pat" <<'PY'
import sqlite3, sys
db = sqlite3.connect(sys.argv[1])
tot = db.execute("select sum(duration) from kernels").fetchone()[0]
print("total kernel ms", tot / 1e6)
q = ("select name, grid_x*grid_y*grid_z/(workgroup_x*workgroup_y*workgroup_z) as wgs, count(*), avg(duration) "
     "from kernels where name like ? group by name, wgs having count(*) >= 20 order by sum(duration) desc")
for r in db.execute(q, ("%" + sys.argv[2] + "%",)):
    print("%-60s wgs %6d  n %5d  avg %8.2f us" % (r[0][9:69], r[1], r[2], r[3] / 1e3))
PY
