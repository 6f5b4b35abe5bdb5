#!/bin/bash
# Launch-floor experiment: the same bench under different HIP runtime knobs (each in its own process).
cd $GRAFT_REPO_ROOT
run() { echo "== $*"; env "$@" python bench.py --steps 100 --warmup 5 --cpu-seconds 0 --roofline-steps 0 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['value'])"; }
run A=1
run HIP_FORCE_DEV_KERNARG=1
run HIP_FORCE_DEV_KERNARG=0
run DEBUG_CLR_GRAPH_PACKET_CAPTURE=1
run DEBUG_CLR_GRAPH_PACKET_CAPTURE=0
run DEBUG_HIP_KERNARG_COPY_OPT=0
run DEBUG_HIP_FORCE_GRAPH_QUEUES=1
run DEBUG_HIP_GRAPH_BATCH_SIZE=1024
run GPU_MAX_HW_QUEUES=1
run HSA_ENABLE_INTERRUPT=0
run HIP_FORCE_DEV_KERNARG=1 DEBUG_CLR_GRAPH_PACKET_CAPTURE=1
