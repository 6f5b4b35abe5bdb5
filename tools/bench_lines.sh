#!/bin/bash
# The round's bench lines, one GPU session (after tools/profile_round.sh, so that bench.py finds the counter file):  tools/bench_lines.sh
# -> gpurun_out/fin_<name>.json, which tools/collect_profiles.sh copies to profiles/<round>_bench_<name>.json
cd $GRAFT_REPO_ROOT
O=gpurun_out
python bench.py > $O/fin_vol128.json 2> $O/fin_vol128.err && echo vol128 done
python bench.py --dtype fp8 --cpu-seconds 0 > $O/fin_vol128-fp8.json 2> $O/fin_vol128-fp8.err && echo vol128 fp8 done
python bench.py --workload mnist > $O/fin_mnist.json 2> $O/fin_mnist.err && echo mnist done
python bench.py --workload vol64-f32 > $O/fin_vol64-f32.json 2> $O/fin_vol64-f32.err && echo vol64-f32 done
python bench.py --workload vol128-vessel --cpu-seconds 0 > $O/fin_vol128-vessel.json 2> $O/fin_vol128-vessel.err && echo vessel done
python bench.py --workload decode --cpu-seconds 0 > $O/fin_decode_bf16_resize128.json 2> $O/fin_decode.err && echo decode done
python bench.py --workload decode --dtype fp8 --cpu-seconds 0 > $O/fin_decode_fp8_resize128.json 2>> $O/fin_decode.err && echo decode fp8 done
python bench.py --workload decode --decode-native --cpu-seconds 0 > $O/fin_decode_bf16_native.json 2>> $O/fin_decode.err && echo decode native done
python bench.py --workload decode --dtype fp8 --decode-native --cpu-seconds 0 > $O/fin_decode_fp8_native.json 2>> $O/fin_decode.err && echo decode fp8 native done
for f in $O/fin_*.json; do python - "$f" <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
print(sys.argv[1].split('/')[-1], d.get("value"), d.get("unit"), d.get("ms_per_step"), (d.get("roofline") or {}).get("frac"))
PY
done
