#!/usr/bin/env python3
"""In-kernel timeline of one GEMM launch: gemm_stamps.py M N K mode(fwd|dx|dw)  [MATMUL=bf16]"""
import os, sys, ctypes
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cswin_unet_amd._lib import call, lib, ptr, stream, precision
M, N, K = (int(a) for a in sys.argv[1:4]); mode = sys.argv[4]
x = torch.randn(M, K, device="cuda"); w = torch.randn(N, K, device="cuda"); b = torch.randn(N, device="cuda")
dy = torch.randn(M, N, device="cuda"); y = torch.empty(M, N, device="cuda"); dx = torch.empty(M, K, device="cuda")
st = torch.zeros(65536, 8, dtype=torch.int64, device="cuda")
import cswin_unet_amd
cswin_unet_amd.set_matmul_precision(os.environ.get("MATMUL", "fp32"))
dw = torch.empty(N, K, device="cuda"); db = torch.empty(N, device="cuda")
nbytes = lib().cswin_linear_bwd_weight_workspace(M, N, K); ws = torch.empty(nbytes // 4 + 4, device="cuda")
h = lib(); h.cswin_debug_set_stamps.argtypes = [ctypes.c_void_p]
S16 = os.environ.get("STORE16", "0") != "0"        # both operands stored as bf16 (io_bf16 = 5): the LDS-DMA kernel, csrc/gemm16.hip
x16, w16, dy16 = x.bfloat16(), w.bfloat16(), dy.bfloat16()
def run():
    if S16 and mode == "fwd": call("cswin_linear_fwd", ptr(x16), None, 0, ptr(w16), ptr(b), ptr(y), None, None, None, 1, M, N, K, 1, 5, stream())
    elif S16 and mode == "dx": call("cswin_linear_bwd_data", ptr(dy16), ptr(w16), ptr(dx), None, 0, None, None, 1, None, M, N, K, 1, 5, stream())
    elif mode == "fwd": call("cswin_linear_fwd", ptr(x), None, 0, ptr(w), ptr(b), ptr(y), None, None, None, 1, M, N, K, precision(), 0, stream())
    elif mode == "dw": call("cswin_linear_bwd_weight", ptr(dy), ptr(x), None, 0, None, 1, ptr(dw), ptr(db), ptr(ws), nbytes, M, N, K, None, precision(), stream())
    else: call("cswin_linear_bwd_data", ptr(dy), ptr(w), ptr(dx), None, 0, None, None, 1, None, M, N, K, precision(), 0, stream())
for _ in range(3): run()
torch.cuda.synchronize()
h.cswin_debug_set_stamps(ctypes.c_void_p(st.data_ptr()))
run(); torch.cuda.synchronize()
h.cswin_debug_set_stamps(None)
s = st.cpu().numpy(); s = s[s[:, 0] != 0]
t0 = s[:, 0].min()
print(f"{mode} M={M} N={N} K={K} blocks={len(s)}")
print("kernel span (cycles):", s[:, 3].max() - t0)
for name, a, b_ in (("prologue", 0, 1), ("mainloop", 1, 2), ("epilogue", 2, 3), ("total", 0, 3)):
    d = s[:, b_] - s[:, a]
    print(f"  {name:9s} mean {d.mean():9.0f}  p10 {np.percentile(d,10):9.0f}  p90 {np.percentile(d,90):9.0f} cycles")
st_rel = s[:, 0] - t0
print("  start offsets: p50 %d p90 %d max %d ; end offsets: p10 %d p50 %d max %d" % (np.percentile(st_rel,50), np.percentile(st_rel,90), st_rel.max(), np.percentile(s[:,3]-t0,10), np.percentile(s[:,3]-t0,50), (s[:,3]-t0).max()))

# wall-clock view (s_memrealtime, 100 MHz, common to all XCDs): when do workgroups start / end?
r0, r1 = s[:, 5], s[:, 6]
base = r0.min()
print("  realtime (us): first start 0, last start %.2f, first end %.2f, last end %.2f" % ((r0.max()-base)/100, (r1.min()-base)/100, (r1.max()-base)/100))
for x in sorted(set(s[:, 4] & 15)):
    m = (s[:, 4] & 15) == x
    print(f"   xcd {x}: {m.sum()} wgs, start {(r0[m].min()-base)/100:.2f}..{(r0[m].max()-base)/100:.2f}, end {(r1[m].min()-base)/100:.2f}..{(r1[m].max()-base)/100:.2f} us; mean life {(r1[m]-r0[m]).mean()/100:.2f} us")

# how many workgroups are alive at once? (sweep over the 100 MHz realtime stamps)
ev = np.concatenate([np.stack([r0, np.ones_like(r0)], 1), np.stack([r1, -np.ones_like(r1)], 1)])
ev = ev[np.argsort(ev[:, 0], kind="stable")]
alive = np.cumsum(ev[:, 1])
dur = np.diff(ev[:, 0])
print("  workgroups alive: max %d, time-average %.0f (of %d launched); per CU %.2f" % (alive.max(), (alive[:-1] * dur).sum() / max(dur.sum(), 1), len(s), (alive[:-1] * dur).sum() / max(dur.sum(), 1) / 256))
