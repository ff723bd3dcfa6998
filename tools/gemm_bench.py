#!/usr/bin/env python3
"""Device-time micro-benchmark of the GEMM family on the model's shapes (GPU box): each entry point is captured 20x
into a hipGraph and replayed, so host launch overhead is excluded.  CSWIN_GEMM_TILE=1|2|3 forces a tile shape."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cswin_unet_amd._lib import call, lib, ptr, stream, precision

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

B = int(os.environ.get("BATCH", "24"))
import cswin_unet_amd
cswin_unet_amd.set_matmul_precision(os.environ.get("MATMUL", "fp32"))      # MATMUL=bf16: bf16 operands (streaming-bound proxy)
shapes = []
for si, (L, C) in enumerate([(3136, 64), (784, 128), (196, 256), (49, 512)]):
    M = B * L
    shapes += [(f"s{si+1}.qkv", M, 3 * C, C), (f"s{si+1}.proj", M, C, C), (f"s{si+1}.fc1", M, 4 * C, C), (f"s{si+1}.fc2", M, C, 4 * C)]
tot = {"fwd": 0, "dx": 0, "dw": 0}
flops = 0
counts = {"s1": 2, "s2": 4, "s3": 18, "s4": 2}
for name, M, N, K in shapes:
    x = torch.randn(M, K, device="cuda"); w = torch.randn(N, K, device="cuda"); b = torch.randn(N, device="cuda")
    dy = torch.randn(M, N, device="cuda"); y = torch.empty(M, N, device="cuda"); dx = torch.empty(M, K, device="cuda")
    dw = torch.empty(N, K, device="cuda"); db = torch.empty(N, device="cuda")
    nbytes = lib().cswin_linear_bwd_weight_workspace(M, N, K)
    ws = torch.empty(nbytes // 4 + 4, device="cuda")
    tf = timed(lambda: call("cswin_linear_fwd", ptr(x), None, 0, ptr(w), ptr(b), ptr(y), None, None, None, 1, M, N, K, precision(), 0, stream()))
    tdx = timed(lambda: call("cswin_linear_bwd_data", ptr(dy), ptr(w), ptr(dx), None, 0, None, None, 1, None, M, N, K, precision(), 0, stream()))
    tdw = timed(lambda: call("cswin_linear_bwd_weight", ptr(dy), ptr(x), None, 0, None, 1, ptr(dw), ptr(db), ptr(ws), nbytes, M, N, K, None, precision(), stream()))
    fl = 2.0 * M * N * K
    c = counts[name[:2]]
    tot["fwd"] += c * tf; tot["dx"] += c * tdx; tot["dw"] += c * tdw; flops += 3 * c * fl
    by = 4.0 * (M * K + N * K + M * N)        # minimal bytes of one pass (each operand once)
    print(f"{name:8s} M={M:6d} N={N:5d} K={K:5d}  fwd {tf*1e6:7.1f}us {fl/tf/1e12:6.1f}TF {by/tf/1e12:4.2f}TB/s | dx {tdx*1e6:7.1f}us {fl/tdx/1e12:6.1f}TF {by/tdx/1e12:4.2f}TB/s | dw {tdw*1e6:7.1f}us {fl/tdw/1e12:6.1f}TF {by/tdw/1e12:4.2f}TB/s")
s = sum(tot.values())
print("per-step totals (ms):", {k: round(v * 1e3, 3) for k, v in tot.items()}, "sum", round(s * 1e3, 3), f"-> {flops/s/1e12:.1f} TF/s over {flops/1e9:.0f} GFLOP")
