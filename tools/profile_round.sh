#!/bin/bash
# Round profile set, one GPU session:  tools/profile_round.sh  -> gpurun_out/prof/{stats,pmc_fetch,pmc_write,pmc_sq}/ + bench JSON under the profiler
# (rocprofv3 --pmc passes carry no tracing flags; the program follows `--` directly.)
R=$GRAFT_REPO_ROOT
P=$R/gpurun_out/prof
rm -rf $P; mkdir -p $P
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $P/stats -o run --output-format csv -- python3 $R/bench.py --cpu-seconds 0 > $P/bench_under_rocprof.json 2> $P/stats.err
echo "stats pass done"
PMCARGS="--no-graph --steps 3 --warmup 2 --roofline-steps 0 --cpu-seconds 0 --min-timed-s 0"
rocprofv3 --pmc FETCH_SIZE -d $P/pmc_fetch -o run --output-format csv -- python3 $R/bench.py $PMCARGS > /dev/null 2> $P/pmc_fetch.err
echo "fetch pass done"
rocprofv3 --pmc WRITE_SIZE -d $P/pmc_write -o run --output-format csv -- python3 $R/bench.py $PMCARGS > /dev/null 2> $P/pmc_write.err
echo "write pass done"
rocprofv3 --pmc SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE -d $P/pmc_sq -o run --output-format csv -- python3 $R/bench.py $PMCARGS > /dev/null 2> $P/pmc_sq.err
echo "sq pass done"
cd $R
python3 tools/step_profile.py gpurun_out/prof/stats 60 --seq > gpurun_out/prof/step_sequence.txt 2>&1
python3 tools/pmc_traffic.py gpurun_out/prof gpurun_out/prof/pmc_traffic.json > gpurun_out/prof/pmc_traffic.txt 2>&1
python3 tools/pmc_sq.py gpurun_out/prof/pmc_sq > gpurun_out/prof/pmc_sq.txt 2>&1
# keep only the summaries (the raw traces are tens of MB)
find $P -name "*kernel_trace.csv" -delete; find $P -name "*agent_info.csv" -delete
ls -la $P $P/stats | head -30
# kernel-stat summaries of the other BASELINE.json workloads (configs[1], configs[2], configs[4])
cd /tmp
for w in mnist vol64-f32 decode decode-fp8 vol128-fp8; do
  case $w in
    decode-fp8) ARGS="--workload decode --dtype fp8";;
    vol128-fp8) ARGS="--dtype fp8";;
    *) ARGS="--workload $w";;
  esac
  rocprofv3 --kernel-trace --stats -d $P/stats_$w -o run --output-format csv -- python3 $R/bench.py $ARGS --cpu-seconds 0 --no-secondary > $P/bench_${w}_under_rocprof.json 2> $P/stats_$w.err
  find $P/stats_$w -name "*kernel_trace.csv" -delete; find $P/stats_$w -name "*agent_info.csv" -delete
  echo "$w stats done"
done
cd $R
