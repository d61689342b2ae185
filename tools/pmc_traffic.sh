#!/bin/bash
# HBM-side traffic of the roofline kernels on THIS tree: two rocprofv3 PMC passes (FETCH_SIZE,
# WRITE_SIZE; counters in their own runs, program directly after `--`) over the bench command at
# the benched time batch, summarised into profiles/<round>_pmc_traffic.json with the SHA-256 of
# csrc/ (bench.py reports `traffic` only for a matching tree).  Run on the GPU box:
#   tools/pmc_traffic.sh r04
set -e
round=${1:-r06}
root=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$root"
T=32
cmd="bench.py --streams 1 --time-batch $T --steps 2 --warmup 1 --no-cpu-baseline --no-uint8 --no-secondary --profile-passes 1"
for c in FETCH_SIZE WRITE_SIZE; do
  tools/pmc_pass.sh ${round}_bench_T$T "$c" "%" -- python3 $cmd > /dev/null
  cp gpurun_out/pmc_${round}_bench_T$T/${round}_bench_T${T}_$c.txt gpurun_out/${round}_bench_T${T}_pmc_$c.txt
  mv /tmp/pmc_${round}_bench_T$T /tmp/pmc_${round}_$c
done
f=$(find /tmp/pmc_${round}_FETCH_SIZE -name "*.db" | head -1)
w=$(find /tmp/pmc_${round}_WRITE_SIZE -name "*.db" | head -1)
python3 tools/pmc_traffic.py "$f" "$w" $T gpurun_out/${round}_pmc_traffic.json "python3 $cmd"
