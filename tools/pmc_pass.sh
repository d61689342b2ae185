#!/bin/bash
# One rocprofv3 PMC pass over a command (counters in their own run: no tracing domains
# besides --kernel-trace), summarised per kernel into profiles/<tag>_<counters>.txt.
#   tools/pmc_pass.sh <tag> "<counter list>" <kernel name pattern> -- python3 prog.py args...
set -e
tag=$1; counters=$2; pat=$3; shift 4
root=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
out=$root/gpurun_out/pmc_$tag
rm -rf /tmp/pmc_$tag; mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp && cd "$root"
# (bounded: a counter group the tool cannot collect has been seen to hang its finaliser for 20 minutes)
if ! timeout -k 10 ${PMC_TIMEOUT:-420} rocprofv3 --kernel-trace --pmc $counters -d /tmp/pmc_$tag -- "$@" > "$out/run.log" 2>&1; then
  echo "pmc_pass: rocprofv3 failed, see $out/run.log" >&2; tail -5 "$out/run.log" >&2; exit 1
fi
db=$(find /tmp/pmc_$tag -name "*.db" | head -1)
[ -n "$db" ] || { echo "pmc_pass: no rocpd database under /tmp/pmc_$tag" >&2; exit 1; }
name=$(echo $counters | tr ' ' '_')
python3 "$root/tools/rocpd_pmc.py" "$db" "$pat" "$out/${tag}_${name}.txt"
cat "$out/${tag}_${name}.txt" | head -30
