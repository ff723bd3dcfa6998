#!/usr/bin/env python3
"""In-kernel timeline of the weight-stationary GEMM (debug stamps of wave 0 of every workgroup, s_memtime = 100 MHz ticks?
no: shader cycles).  Usage: ws_stamps.py M N K [fwd|dx]"""
import ctypes, os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cswin_unet_amd._lib import call, lib, ptr, stream, precision
M, N, K = (int(v) for v in sys.argv[1:4])
mode = sys.argv[4] if len(sys.argv) > 4 else "fwd"
H = lib()
H.cswin_debug_set_ws_stamps.argtypes = [ctypes.c_void_p]
x = torch.randn(M, K, device="cuda"); w = torch.randn(N, K, device="cuda"); b = torch.randn(N, device="cuda")
dy = torch.randn(M, N, device="cuda"); y = torch.empty(M, N, device="cuda"); dx = torch.empty(M, K, device="cuda")
st = torch.zeros(512, 16, dtype=torch.int64, device="cuda")
fn = (lambda: call("cswin_linear_fwd", ptr(x), None, 0, ptr(w), ptr(b), ptr(y), None, None, None, 1, M, N, K, precision(), 0, stream())) if mode == "fwd" else \
     (lambda: call("cswin_linear_bwd_data", ptr(dy), ptr(w), ptr(dx), None, 0, None, None, 1, None, M, N, K, precision(), 0, stream()))
for _ in range(5): fn()
torch.cuda.synchronize()
H.cswin_debug_set_ws_stamps(ctypes.c_void_p(st.data_ptr()))
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record(); fn(); e1.record(); torch.cuda.synchronize()
H.cswin_debug_set_ws_stamps(None)
s = st.cpu().numpy()
s = s[s[:, 0] > 0]
t0 = s[:, 0].min()
print(f"{mode} M={M} N={N} K={K}: {len(s)} workgroups, launch {e0.elapsed_time(e1)*1e3:.1f} us; stamps in cycles relative to the first workgroup's start")
names = ["start", "W regs + DMA issued", "first data landed"] + [f"tile{(i-3)//2} {'mfma done' if (i-3)%2==0 else 'epilogue done'}" for i in range(3, 15)] + ["end"]
for k in range(16):
    col = s[:, k]
    ok = col > 0
    if not ok.any(): continue
    v = col[ok] - t0
    print(f"  {names[k]:28s} min {v.min():7d}  median {int(np.median(v)):7d}  max {v.max():7d}   (n={ok.sum()})")
print("  total span (max end - min start):", (s[:, 15].max() - t0), "cycles")
