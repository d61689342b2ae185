# longer soak of the configurations whose BiFPN node kernels changed late in the round (profiles/r06_soak_long.txt)
out=gpurun_out/r06_soak_long.txt; : > $out
run() { echo "== $1 T=$2 reps=$3" >> $out; CASE=$1 T=$2 REPS=$3 timeout 1500 python tools/soak_check.py 2>&1 | grep "^soak" >> $out; }
run cfg3_large 32 300
run cfg3_large 4 1500
run cfg3_large 1 1500
run default_medium_320 16 600
run default_medium_320 4 1500
run cfg3_medium 32 300
run cfg3_medium 1 1500
cat $out
