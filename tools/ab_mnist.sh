#!/bin/bash
# A/B of library builds on the MNIST adversarial step: tools/ab_mnist.sh <rounds> lib1.so lib2.so ...
cd $GRAFT_REPO_ROOT
rounds=$1; shift
for r in $(seq 1 $rounds); do
  for lib in "$@"; do
    CVAE_HIP_LIB=$GRAFT_REPO_ROOT/causal_vae_amd/$lib python bench.py --workload mnist --steps 50 --warmup 5 --cpu-seconds 0 --roofline-steps 0 --min-timed-s 0.4 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$lib', round(d['ms_per_step'],4), round(d['value']))"
  done
done
