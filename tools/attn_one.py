#!/usr/bin/env python3
"""Run the stripe-attention fwd+bwd kernels of one stage a few times (for rocprofv3 --pmc passes): attn_one.py stage [reps]"""
import os, sys, ctypes
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cswin_unet_amd._lib import call, lib, ptr, stream
si = int(sys.argv[1]) - 1; reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3; batch = 24
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
for _ in range(reps):
    call("cswin_attn_fwd", ptr(qkv), pa(w), pa(b), ptr(y), ptr(y0), ptr(lse), batch, reso, C, nb, ha, ia, split[si], 0.0, 0.0, 0, None, 0, stream())
    call("cswin_attn_bwd", ptr(qkv), pa(w), pa(b), ptr(lse), ptr(y0), ptr(dy), ptr(dqkv), pa(dw), pa(db), ptr(ws), nbytes, batch, reso, C, nb, ha, ia, split[si], 0.0, None, 0.0, 0, None, 0, stream())
torch.cuda.synchronize()
print("algorithmic MB: fwd", 16 * L * C * batch / 1e6, "bwd", 28 * L * C * batch / 1e6)
