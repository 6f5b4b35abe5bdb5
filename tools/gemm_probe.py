"""Time the three bf16-operand linear products of one layer shape through the C-ABI (HIP events, graph-free): tools/gemm_probe.py M K N"""
import os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from causal_vae_amd import _lib as L
lib = L.lib
M, K, N = (int(v) for v in sys.argv[1:4]) if len(sys.argv) > 3 else (1024, 3158, 512)
dev = "cuda:0"
x, W, b = torch.randn(M, K, device=dev), torch.randn(N, K, device=dev) * 0.02, torch.randn(N, device=dev)
g = torch.randn(M, N, device=dev)
y, dx, dW, db = torch.empty(M, N, device=dev), torch.empty(M, K, device=dev), torch.empty(N, K, device=dev), torch.empty(N, device=dev)
ws = [torch.empty(max(1, int(lib.cvae_linear_workspace_bytes(M, K, N, op)) // 4), device=dev) for op in range(3)]
st = torch.cuda.current_stream().cuda_stream
p = L.ptr
calls = {
    "fwd": lambda: lib.cvae_linear_fwd_bf16(p(x), p(W), p(b), p(y), M, K, N, K, N, 0, p(ws[0]), ws[0].numel() * 4, st),
    "bwd_data": lambda: lib.cvae_linear_bwd_data_bf16(p(g), p(W), p(dx), M, K, N, N, K, p(ws[1]), ws[1].numel() * 4, st),
    "bwd_weight": lambda: lib.cvae_linear_bwd_weight_bf16(p(g), p(x), p(dW), p(db), M, K, N, N, K, p(ws[2]), ws[2].numel() * 4, st),
}
for name, f in calls.items():
    for _ in range(5):
        assert f() == 0
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50):
        f()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / 50
    print(f"{name:10s} M{M} K{K} N{N}: {us:7.1f} us  {2.0 * M * K * N / us * 1e-6:7.1f} TFLOP/s")
