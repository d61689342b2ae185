# soak of the round: every fixture geometry, both time-batch classes of the wide models (profiles/<tag>_soak.txt)
tag=${1:-r06}; out=gpurun_out/${tag}_soak.txt; : > $out
run() { echo "== $1 T=$2 reps=$3" >> $out; CASE=$1 T=$2 REPS=$3 timeout 900 python tools/soak_check.py 2>&1 | grep "^soak" >> $out; }
run cfg3 32 300
run cfg3_medium 32 80
run cfg3_large 32 60
run cfg3_large 4 300
run cfg3_medium 4 300
run ex72 24 150
run default_medium_320 16 150
run default_medium_320 4 300
run cfg3_cam_black 32 100
run cfg2 8 300
cat $out
