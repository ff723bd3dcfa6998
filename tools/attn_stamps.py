#!/usr/bin/env python3
"""In-kernel timeline of the attention kernels: attn_stamps.py stage(1..4) [batch] [fwd]"""
import os, sys, ctypes
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cswin_unet_amd._lib import call, lib, ptr, stream
si = int(sys.argv[1]) - 1; batch = int(sys.argv[2]) if len(sys.argv) > 2 else 24; FWD = len(sys.argv) > 3 and sys.argv[3] == "fwd"
E, heads, split = 64, [2, 4, 8, 16], [1, 2, 7, 7]
C, reso = E << si, 56 >> si; L = reso * reso
single = si == 3
idx = [-1] if single else [0, 1]; hb = [heads[si]] if single else [heads[si] // 2] * 2
nb, cb = len(idx), C // len(idx)
dev = "cuda"
qkv = torch.randn(batch, L, 3 * C, device=dev); w = [torch.randn(cb, 9, device=dev) / 3 for _ in idx]; b = [torch.randn(cb, device=dev) * .02 for _ in idx]
dy = torch.randn(batch, L, C, device=dev); y = torch.empty(batch, L, C, device=dev); y0 = torch.empty_like(y); lse = torch.empty(batch, sum(hb), L, device=dev)
dqkv = torch.empty_like(qkv); dw = [torch.empty_like(t) for t in w]; db = [torch.empty_like(t) for t in b]
ia, ha = (ctypes.c_int * nb)(*idx), (ctypes.c_int * nb)(*hb)
pa = lambda ts: (ctypes.c_void_p * nb)(*[t.data_ptr() for t in ts])
nbytes = lib().cswin_attn_bwd_workspace(batch, reso, C, nb, ha, ia, split[si]); ws = torch.empty(nbytes // 4 + 4, device=dev)
call("cswin_attn_fwd", ptr(qkv), pa(w), pa(b), ptr(y), ptr(y0), ptr(lse), batch, reso, C, nb, ha, ia, split[si], 0.0, 0.0, 0, None, 0, stream())
def bwd(): call("cswin_attn_bwd", ptr(qkv), pa(w), pa(b), ptr(lse), ptr(y0), ptr(dy), ptr(dqkv), pa(dw), pa(db), ptr(ws), nbytes, batch, reso, C, nb, ha, ia, split[si], 0.0, None, 0.0, 0, None, 0, stream())
for _ in range(3): bwd()
torch.cuda.synchronize()
st = torch.zeros(1 << 16, 8, dtype=torch.int64, device=dev)
h = lib(); h.cswin_debug_set_attn_stamps.argtypes = [ctypes.c_void_p]
def fwd(): call("cswin_attn_fwd", ptr(qkv), pa(w), pa(b), ptr(y), ptr(y0), ptr(lse), batch, reso, C, nb, ha, ia, split[si], 0.0, 0.0, 0, None, 0, stream())
h.cswin_debug_set_attn_stamps(ctypes.c_void_p(st.data_ptr())); (fwd if FWD else bwd)(); torch.cuda.synchronize(); h.cswin_debug_set_attn_stamps(None)
s = st.cpu().numpy(); s = s[s[:, 0] != 0]
print(f"stage {si+1} {'forward' if FWD else 'backward'}: {len(s)} workgroups")
names = ["A1 q / lse -> LDS, barrier", "S tiles; A2 dO / V / delta -> LDS, barrier", "B LePE wgrad", "barrier, C fused loop, dK/dV", "barrier, K image, barrier", "D dQ"]
if FWD: names = ["K -> LDS, barrier", "S + softmax", "V -> LDS, barrier, PV", "LePE + store"]
for k, nm in enumerate(names):
    d = s[:, k + 1] - s[:, k]
    print(f"  {nm:44s} mean {d.mean():8.0f} p90 {np.percentile(d, 90):8.0f} shader cycles (s_memtime)")
d = s[:, len(names)] - s[:, 0]; print(f"  {'workgroup life':44s} mean {d.mean():8.0f} p90 {np.percentile(d, 90):8.0f} max {d.max():8.0f}")
