"""What does a dependent launch cost INSIDE a torch-captured HIP graph, for this library's kernels?  (tools/probes/launch_gap_probe.hip is the plain-HIP
twin: 1.7-1.9 us per node there; rocprofv3 shows 4.7 us for every trivial kernel of the replayed train step.)"""
import os, sys, time
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from causal_vae_amd import _lib as L
from causal_vae_amd._lib import lib, ptr, stream

dev = torch.device("cuda")
cnt = torch.zeros(4, dtype=torch.int32, device=dev)
one = torch.zeros(1, device=dev)
big = torch.zeros(16 << 20, device=dev)           # 64 MB
coef = torch.ones((), device=dev)


def timed_graph(fn, K, reps=30):
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        for _ in range(3):
            fn(0)
    torch.cuda.current_stream().wait_stream(s)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, capture_error_mode="thread_local"):
        for i in range(K):
            fn(i)
    best = 1e9
    for _ in range(reps):
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); g.replay(); e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1))
    return best * 1e3 / K


K = 64
print("lib tiny (cvae_counter_add) x64 in a torch graph      : %.2f us/launch" % timed_graph(lambda i: lib.cvae_counter_add(ptr(cnt), 1, stream()), K))
print("torch tiny (one.add_(1)) x64 in a torch graph         : %.2f us/launch" % timed_graph(lambda i: one.add_(1.0), K))
n = big.numel()
print("lib scale 64 MB x16                                   : %.2f us/launch" % timed_graph(lambda i: lib.cvae_scale(ptr(big), n, ptr(coef), stream()), 16))
def alt(i):
    if i & 1:
        lib.cvae_counter_add(ptr(cnt), 1, stream())
    else:
        lib.cvae_scale(ptr(big), n, ptr(coef), stream())
print("lib scale 64 MB + tiny, pairs x16                     : %.2f us/pair" % (2 * timed_graph(alt, 32)))
n4 = 1 << 20
def alt4(i):
    if i & 1:
        lib.cvae_counter_add(ptr(cnt), 1, stream())
    else:
        lib.cvae_scale(ptr(big), n4, ptr(coef), stream())
print("lib scale 4 MB x32                                    : %.2f us/launch" % timed_graph(lambda i: lib.cvae_scale(ptr(big), n4, ptr(coef), stream()), 32))
print("lib scale 4 MB + tiny, pairs x16                      : %.2f us/pair" % (2 * timed_graph(alt4, 32)))
