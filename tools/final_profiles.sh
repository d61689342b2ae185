#!/bin/bash
# Regenerates the judged artefacts on THIS tree (run on the GPU box): PMC traffic of the roofline kernels, the bench
# lines of configs[2] / [1] / [4], and the rocprofv3 kernel-trace summary of a single-stream run.
#   bash tools/final_profiles.sh r04z
tag=${1:-r04z}
root=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$root"
mkdir -p gpurun_out
bash tools/pmc_traffic.sh r04 > gpurun_out/${tag}_pmc.log 2>&1 && cp gpurun_out/r04_pmc_traffic.json profiles/r04_pmc_traffic.json
python3 bench.py > gpurun_out/${tag}_bench_cfg3_3x32.json 2> gpurun_out/${tag}_bench_cfg3.err
python3 bench.py --config cfg5 --time-batch 8 --no-secondary > gpurun_out/${tag}_bench_cfg5_3x8.json 2> /dev/null
python3 bench.py --config cfg2 --no-secondary > gpurun_out/${tag}_bench_cfg2_3x32.json 2> /dev/null
python3 bench.py --config ex72 --no-secondary --no-cpu-baseline > gpurun_out/${tag}_bench_ex72_3x24.json 2> /dev/null
python3 bench.py --model-size medium --no-secondary --no-cpu-baseline --no-uint8 --no-reduced-precision > gpurun_out/${tag}_bench_cfg3_medium_3x32.json 2> /dev/null
python3 bench.py --model-size large --no-secondary --no-cpu-baseline --no-uint8 --no-reduced-precision > gpurun_out/${tag}_bench_cfg3_large_3x32.json 2> /dev/null
cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/kt_final
rocprofv3 --kernel-trace --stats -d /tmp/kt_final -o run -- python3 $root/bench.py --streams 1 --steps 10 --warmup 3 \
  --no-cpu-baseline --no-uint8 --no-secondary > /tmp/kt_final.log 2>&1
db=$(find /tmp/kt_final -name "*.db" | head -1)
python3 $root/tools/rocpd_summary.py $db $root/gpurun_out/${tag}_bench_cfg3_1x32_kernel_stats.csv
cd $root
for f in gpurun_out/${tag}_bench_*.json; do python3 - $f <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).readline())
r = d.get("roofline", {})
print(sys.argv[1], d["value"], "traffic", r.get("traffic"), "frac", r.get("frac"),
      "bf16x3", (d.get("reduced_precision") or {}).get("value"))
PY
done
