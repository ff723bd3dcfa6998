#!/usr/bin/env python3
"""Calibration only (NOT a product path): what the vendor fp32 GEMM (torch.mm -> rocBLAS/hipBLASLt) achieves on the
model's Linear shapes on this GPU, as a known-good reference for judging libcswin_hip's gemm_kernel."""
import os, sys, torch
def timed(fn, reps=20, rounds=5):
    fn(); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(reps): fn()
    g.replay(); torch.cuda.synchronize()
    best = 1e9
    for _ in range(rounds):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) * 1e-3 / reps)
    return best
torch.backends.cuda.matmul.allow_tf32 = False
B = 24; tot = {"fwd": 0, "dx": 0, "dw": 0}; flops = 0
counts = {"s1": 2, "s2": 4, "s3": 18, "s4": 2}
for si, (L, C) in enumerate([(3136, 64), (784, 128), (196, 256), (49, 512)]):
    M = B * L
    for nm, N, K in (("qkv", 3 * C, C), ("proj", C, C), ("fc1", 4 * C, C), ("fc2", C, 4 * C)):
        x = torch.randn(M, K, device="cuda"); w = torch.randn(N, K, device="cuda"); dy = torch.randn(M, N, device="cuda")
        y = torch.empty(M, N, device="cuda"); dx = torch.empty(M, K, device="cuda"); dw = torch.empty(N, K, device="cuda")
        tf = timed(lambda: torch.mm(x, w.t(), out=y)); tdx = timed(lambda: torch.mm(dy, w, out=dx)); tdw = timed(lambda: torch.mm(dy.t(), x, out=dw))
        fl = 2.0 * M * N * K; c = counts[f"s{si+1}"]
        tot["fwd"] += c * tf; tot["dx"] += c * tdx; tot["dw"] += c * tdw; flops += 3 * c * fl
        print(f"s{si+1}.{nm:5s} M={M:6d} N={N:5d} K={K:5d}  fwd {tf*1e6:7.1f}us {fl/tf/1e12:6.1f}TF | dx {tdx*1e6:7.1f}us {fl/tdx/1e12:6.1f}TF | dw {tdw*1e6:7.1f}us {fl/tdw/1e12:6.1f}TF")
s = sum(tot.values())
print("per-step totals (ms):", {k: round(v * 1e3, 3) for k, v in tot.items()}, "sum", round(s * 1e3, 3), f"-> {flops/s/1e12:.1f} TF/s")
