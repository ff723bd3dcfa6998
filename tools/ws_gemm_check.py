#!/usr/bin/env python3
"""Weight-stationary GEMM family (csrc/wsgemm.hip) vs the tiled family (csrc/gemm.hip) on the model's Linear shapes:
max |diff| / rms of the outputs (same inputs, both through the C ABI) and device time of each (hipGraph replay)."""
import ctypes, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cswin_unet_amd._lib import call, lib, ptr, stream, precision
from gemm_bench_util import timed

B = int(os.environ.get("BATCH", "24"))
H = lib()
H.cswin_debug_set_ws_gemm.argtypes = [ctypes.c_int]
shapes = []
for si, (L, C) in enumerate([(3136, 64), (784, 128), (196, 256), (49, 512)]):
    M = B * L
    shapes += [(f"s{si+1}.qkv", M, 3 * C, C), (f"s{si+1}.proj", M, C, C), (f"s{si+1}.fc1", M, 4 * C, C), (f"s{si+1}.fc2", M, C, 4 * C)]
counts = {"s1": 2, "s2": 4, "s3": 18, "s4": 2}
tot = {"fwd_old": 0, "fwd_ws": 0, "dx_old": 0, "dx_ws": 0}
worst = 0.0
for name, M, N, K in shapes:
    x = torch.randn(M, K, device="cuda"); w = torch.randn(N, K, device="cuda") / K ** 0.5; b = torch.randn(N, device="cuda")
    dy = torch.randn(M, N, device="cuda"); res = torch.randn(M, N, device="cuda"); pre = torch.randn(M, K, device="cuda")
    rs = torch.rand(B, device="cuda") + 0.5
    L = M // B
    outs = {}
    times = {}
    for on in (0, 1):
        H.cswin_debug_set_ws_gemm(on)
        y = torch.empty(M, N, device="cuda"); ya = torch.empty(M, N, device="cuda"); yr = torch.empty(M, N, device="cuda")
        dx = torch.empty(M, K, device="cuda"); dxg = torch.empty(M, K, device="cuda")
        f_plain = lambda: call("cswin_linear_fwd", ptr(x), None, 0, ptr(w), ptr(b), ptr(y), None, None, None, 1, M, N, K, precision(), 0, stream())
        f_act = lambda: call("cswin_linear_fwd", ptr(x), None, 0, ptr(w), ptr(b), ptr(y), ptr(ya), None, None, 1, M, N, K, precision(), 0, stream())
        f_res = lambda: call("cswin_linear_fwd", ptr(x), None, 0, ptr(w), ptr(b), ptr(yr), None, ptr(res), ptr(rs), L, M, N, K, precision(), 0, stream())
        d_plain = lambda: call("cswin_linear_bwd_data", ptr(dy), ptr(w), ptr(dx), None, 0, None, None, 1, None, M, N, K, precision(), 0, stream())
        d_gelu = lambda: call("cswin_linear_bwd_data", ptr(dy), ptr(w), ptr(dxg), None, 0, ptr(pre), ptr(rs), L, None, M, N, K, precision(), 0, stream())
        f_act(); f_res(); d_gelu()
        times[on] = (timed(f_plain), timed(d_plain))
        torch.cuda.synchronize()
        outs[on] = [t.clone() for t in (y, ya, yr, dx, dxg)]
    errs = []
    for a, c in zip(outs[0], outs[1]):
        errs.append(float((a - c).abs().max() / (a.pow(2).mean().sqrt() + 1e-30)))
    worst = max(worst, max(errs))
    fl = 2.0 * M * N * K
    c = counts[name[:2]]
    tot["fwd_old"] += c * times[0][0]; tot["fwd_ws"] += c * times[1][0]; tot["dx_old"] += c * times[0][1]; tot["dx_ws"] += c * times[1][1]
    print(f"{name:8s} M={M:6d} N={N:5d} K={K:5d}  fwd tiled {times[0][0]*1e6:6.1f}us ws {times[1][0]*1e6:6.1f}us ({fl/times[1][0]/1e12:5.1f}TF) | "
          f"dx tiled {times[0][1]*1e6:6.1f}us ws {times[1][1]*1e6:6.1f}us ({fl/times[1][1]/1e12:5.1f}TF) | max err/rms {max(errs):.2e}", flush=True)
print("per-step totals (ms):", {k: round(v * 1e3, 3) for k, v in tot.items()}, "worst err", f"{worst:.2e}")
assert worst < 2e-5, worst
