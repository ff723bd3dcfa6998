#!/usr/bin/env python3
"""Run one GEMM shape a few times (for rocprofv3 --pmc passes): gemm_one.py M N K [reps]"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cswin_unet_amd._lib import call, lib, ptr, stream, precision
M, N, K = (int(a) for a in sys.argv[1:4])
reps = int(sys.argv[4]) if len(sys.argv) > 4 else 5
x = torch.randn(M, K, device="cuda"); w = torch.randn(N, K, device="cuda"); b = torch.randn(N, device="cuda")
dy = torch.randn(M, N, device="cuda"); y = torch.empty(M, N, device="cuda"); dx = torch.empty(M, K, device="cuda")
dw = torch.empty(N, K, device="cuda"); db = torch.empty(N, device="cuda")
nbytes = lib().cswin_linear_bwd_weight_workspace(M, N, K)
ws = torch.empty(nbytes // 4 + 4, device="cuda")
for _ in range(reps):
    call("cswin_linear_fwd", ptr(x), None, 0, ptr(w), ptr(b), ptr(y), None, None, None, 1, M, N, K, precision(), 0, stream())
    call("cswin_linear_bwd_data", ptr(dy), ptr(w), ptr(dx), None, 0, None, None, 1, None, M, N, K, precision(), 0, stream())
    call("cswin_linear_bwd_weight", ptr(dy), ptr(x), None, 0, None, 1, ptr(dw), ptr(db), ptr(ws), nbytes, M, N, K, None, precision(), stream())
torch.cuda.synchronize()
