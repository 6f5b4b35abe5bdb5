import json, sys
d = json.load(open(sys.argv[1]))
print(f"{d['value']:.1f} samples/s  {d['ms_per_step']:.3f} ms/step  conv {d.get('conv_ms_per_step', 0):.3f} ms  graph={d.get('hip_graph')}")
for k, v in list(d.get("kernels", {}).items())[: int(sys.argv[2]) if len(sys.argv) > 2 else 12]:
    print(f"{v['avg_ms'] * 1e3:7.1f} us {v['tflops']:7.1f} TF  {k}")
