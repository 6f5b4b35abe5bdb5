cd $GRAFT_REPO_ROOT
for r in 1 2 3; do
  for lib in libcvae_hip.so libcvae_hip_nowalk.so; do
    for dt in bf16 fp8; do
      CVAE_HIP_LIB=$GRAFT_REPO_ROOT/causal_vae_amd/$lib python bench.py --workload decode --dtype $dt --steps 20 --warmup 3 --cpu-seconds 0 --roofline-steps 0 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$lib', '$dt', round(d['ms_per_step'],4), d['value'])"
    done
  done
done
