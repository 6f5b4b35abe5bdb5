#!/bin/bash
# FETCH_SIZE / WRITE_SIZE passes of a secondary workload (separate --pmc runs, no tracing flags, program directly after --):
#   tools/pmc_workload.sh <workload> <marker kernel substring> <marker count per step> [extra bench args]  -> gpurun_out/pmc_<workload>.txt
R=$GRAFT_REPO_ROOT; W=$1; M=$2; C=$3; shift 3
P=$R/gpurun_out/pmc_$W; rm -rf $P; mkdir -p $P
cd /tmp && export TMPDIR=/tmp
ARGS="--workload $W --no-graph --steps 3 --warmup 2 --roofline-steps 0 --cpu-seconds 0 --min-timed-s 0 $*"
rocprofv3 --pmc FETCH_SIZE -d $P/pmc_fetch -o run --output-format csv -- python3 $R/bench.py $ARGS > /dev/null 2> $P/fetch.err
rocprofv3 --pmc WRITE_SIZE -d $P/pmc_write -o run --output-format csv -- python3 $R/bench.py $ARGS > /dev/null 2> $P/write.err
cd $R && python3 tools/pmc_generic.py $P "$M" $C > gpurun_out/pmc_$W.txt 2>&1
rm -rf $P/pmc_fetch $P/pmc_write
cat gpurun_out/pmc_$W.txt | cut -c1-170
