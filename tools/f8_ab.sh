#!/bin/bash
# Device-side durations (rocprofv3 --kernel-trace) of the fp8 forward launches for several library builds:  tools/f8_ab.sh lib1.so lib2.so ...
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for lib in "$@"; do
  for mode in "--fp8 --fp8-plain" "--fp8 --fp8-side out8" "--fp8 --fp8-side amax" "--fp8" ""; do
    rm -rf /tmp/kt; export CVAE_HIP_LIB=$R/causal_vae_amd/$lib
    rocprofv3 --kernel-trace -d /tmp/kt -o run --output-format csv -- python3 $R/tools/kbench.py enc2.fwd enc3.fwd enc4.fwd dec1.fwd dec2.fwd dec3.fwd $mode --iters 30 > /dev/null 2>&1
    echo "=== $lib [$mode]"
    python3 $R/tools/ktrace_avg.py /tmp/kt conv_data conv_splitk | cut -c1-150
  done
done
