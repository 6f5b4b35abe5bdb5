#!/bin/bash
# After `gpurun -- bash tools/profile_round.sh` (+ the bench lines written as gpurun_out/fin_<workload>.json): copy the summaries into profiles/
# under this round's names and tag the counter file with the commit whose kernels it measured.   tools/collect_profiles.sh r02
set -e
R=${1:-r02}; P=gpurun_out/prof
cp $P/stats/run_kernel_stats.csv profiles/${R}_bench_kernel_stats.csv
cp $P/bench_under_rocprof.json profiles/${R}_bench_under_rocprof.json
cp $P/step_sequence.txt profiles/${R}_step_sequence.txt
cp $P/pmc_traffic.txt profiles/${R}_pmc_traffic.txt
cp $P/pmc_sq.txt profiles/${R}_pmc_sq.txt
python3 - "$R" <<'PY'
import json, subprocess, sys
r = sys.argv[1]
h = subprocess.run(["git", "log", "-1", "--format=%h"], capture_output=True, text=True).stdout.strip()
d = json.load(open("gpurun_out/prof/pmc_traffic.json"))
d["_commit"] = h + " (kernels of this commit; bench.py --no-graph under rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, tools/profile_round.sh)"
json.dump(d, open(f"profiles/{r}_pmc_traffic.json", "w"), indent=1)
PY
for w in mnist vol64-f32 decode decode-fp8 vol128-fp8; do
  [ -f $P/stats_$w/run_kernel_stats.csv ] && cp $P/stats_$w/run_kernel_stats.csv profiles/${R}_bench_${w}_kernel_stats.csv && cp $P/bench_${w}_under_rocprof.json profiles/${R}_bench_${w}_under_rocprof.json
done
for f in gpurun_out/fin_*.json; do w=$(basename $f .json); cp $f profiles/${R}_bench_${w#fin_}.json; done
ls profiles | grep "^$R" | wc -l
