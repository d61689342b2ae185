#!/bin/bash
# Regenerates the judged artefacts on THIS tree (run on the GPU box): PMC traffic of the roofline kernels, the bench
# lines of configs[2] (small / medium / large), configs[1]'s geometry, configs[4] and the shipped 72^3 geometry, the full
# per-kernel table, per-launch tables, and the rocprofv3 kernel-trace summary of a single-stream run.  Everything lands
# in gpurun_out/<tag>_*; copy what is to be judged into profiles/.
#   bash tools/final_profiles.sh r06
tag=${1:-r06}
root=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$root"
mkdir -p gpurun_out
bash tools/pmc_traffic.sh $tag > gpurun_out/${tag}_pmc.log 2>&1 && cp gpurun_out/${tag}_pmc_traffic.json profiles/${tag}_pmc_traffic.json
# (the full per-kernel table of the headline run: JH_BENCH_TOP)
JH_BENCH_TOP=200 python3 bench.py > gpurun_out/${tag}_bench_cfg3_3x32.json 2> gpurun_out/${tag}_bench_cfg3.err
python3 bench.py --config cfg5 --time-batch 8 --no-secondary > gpurun_out/${tag}_bench_cfg5_3x8.json 2> /dev/null
python3 bench.py --config cfg2 --no-secondary > gpurun_out/${tag}_bench_cfg2_3x32.json 2> /dev/null
python3 bench.py --config ex72 --no-secondary --no-cpu-baseline > gpurun_out/${tag}_bench_ex72_3x24.json 2> /dev/null
python3 bench.py --model-size medium --no-secondary --no-cpu-baseline --no-uint8 --no-reduced-precision > gpurun_out/${tag}_bench_cfg3_medium_3x32.json 2> /dev/null
python3 bench.py --model-size large --no-secondary --no-cpu-baseline --no-uint8 --no-reduced-precision > gpurun_out/${tag}_bench_cfg3_large_3x32.json 2> /dev/null
# the reference's DEFAULT configuration (medium / medium, 320 / 320, 72^3): its own bench line, with the CPU baseline
python3 bench.py --config def320 --no-secondary > gpurun_out/${tag}_bench_def320_3x16.json 2> /dev/null
python3 tools/launch_table.py --config def320 --out gpurun_out/${tag}_launches_def320_medium.tsv > /dev/null 2>&1
for m in small medium large; do python3 tools/launch_table.py --model-size $m --out gpurun_out/${tag}_launches_cfg3_$m.tsv > /dev/null 2>&1; done
python3 tools/launch_table.py --config ex72 --out gpurun_out/${tag}_launches_ex72_small.tsv > /dev/null 2>&1
python3 tools/launch_table.py --config cfg5 --out gpurun_out/${tag}_launches_cfg5_small.tsv > /dev/null 2>&1
python3 tools/launch_table.py -T 1 --out gpurun_out/${tag}_launches_cfg3_small_T1.tsv > /dev/null 2>&1
python3 tools/launch_table.py -T 1 --model-size medium --out gpurun_out/${tag}_launches_cfg3_medium_T1.tsv > /dev/null 2>&1
cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/kt_final
rocprofv3 --kernel-trace --stats -d /tmp/kt_final -o run -- python3 $root/bench.py --streams 1 --steps 10 --warmup 3 \
  --no-cpu-baseline --no-uint8 --no-secondary > /tmp/kt_final.log 2>&1
db=$(find /tmp/kt_final -name "*.db" | head -1)
python3 $root/tools/rocpd_summary.py $db $root/gpurun_out/${tag}_bench_cfg3_1x32_kernel_stats.csv
cd $root
for f in gpurun_out/${tag}_bench_*.json; do python3 - $f <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
r = d.get("roofline", {})
print(sys.argv[1], round(d["value"], 1), "traffic", r.get("traffic"), r.get("traffic_note"), "frac", r.get("frac"),
      "bf16x3", (d.get("reduced_precision") or {}).get("value"), "lat", d.get("single_frame_latency_ms"))
PY
done
