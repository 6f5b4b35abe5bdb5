#!/bin/bash
# A/B of library builds on the decode sweep: tools/ab_decode.sh <rounds> lib1.so lib2.so ...   (interleaved rounds; bf16 and fp8, resize to 128^3)
cd $GRAFT_REPO_ROOT
rounds=$1; shift
for r in $(seq 1 $rounds); do
  for lib in "$@"; do
    for dt in bf16 fp8; do
      CVAE_HIP_LIB=$GRAFT_REPO_ROOT/causal_vae_amd/$lib python bench.py --workload decode --dtype $dt --steps 20 --warmup 3 --cpu-seconds 0 --roofline-steps 0 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$lib', '$dt', round(d['ms_per_step'],4), round(d['value']))"
    done
  done
done
