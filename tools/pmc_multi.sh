#!/bin/bash
# Several rocprofv3 PMC passes (one counter group each, ';'-separated) over one command, all
# summarised for kernels matching a name pattern into gpurun_out/<tag>_pmc.txt.
#   tools/pmc_multi.sh <tag> "<kernel pattern>" "<group1>;<group2>;..." -- python3 prog.py args...
tag=$1; pat=$2; groups=$3; shift 4
root=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
out=$root/gpurun_out/${tag}_pmc.txt
: > "$out"
IFS=';' read -ra G <<< "$groups"
i=0
for g in "${G[@]}"; do
  i=$((i+1))
  if "$root/tools/pmc_pass.sh" ${tag}_g$i "$g" "$pat" -- "$@" > /dev/null 2>&1; then
    cat "$root"/gpurun_out/pmc_${tag}_g$i/*.txt >> "$out"
  else
    echo "group $i FAILED: $g" >> "$out"; tail -3 "$root/gpurun_out/pmc_${tag}_g$i/run.log" >> "$out"
  fi
done
cat "$out"
