#!/usr/bin/env python3
"""bf16-mode weight-gradient batch of one CSWinBlock per stage (the four Linears in one launch): device time.
Run twice: CSWIN_WGRAD16=0 (tiled family, bf16 operands) and default (wgrad16.hip)."""
import ctypes, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import cswin_unet_amd
from cswin_unet_amd._lib import ReduceJob, WgradDesc, call, lib, stream
from gemm_bench_util import timed
cswin_unet_amd.set_matmul_precision(os.environ.get("MATMUL", "bf16"))
B = 24
tot = 0.0
for si, (L, C, cnt) in enumerate([(3136, 64, 2), (784, 128, 4), (196, 256, 18), (49, 512, 2)]):
    M = B * L
    shapes = [(C, 4 * C), (4 * C, C), (C, C), (3 * C, C)]
    wg, jobs, keep = (WgradDesc * 4)(), (ReduceJob * 4)(), []
    for i, (N, K) in enumerate(shapes):
        dy = torch.randn(M, N, device="cuda"); x = torch.randn(M, K, device="cuda")
        dw = torch.empty(N, K, device="cuda"); db = torch.empty(N, device="cuda")
        nbytes = lib().cswin_linear_bwd_weight_workspace(M, N, K)
        ws = torch.empty(nbytes // 4 + 4, device="cuda")
        keep += [dy, x, dw, db, ws]
        wg[i].dy, wg[i].x, wg[i].row_scale, wg[i].dw, wg[i].dbias = dy.data_ptr(), x.data_ptr(), None, dw.data_ptr(), db.data_ptr()
        wg[i].workspace, wg[i].ws_bytes, wg[i].rows_per_sample, wg[i].M, wg[i].N, wg[i].K = ws.data_ptr(), nbytes, 1, M, N, K
        wg[i].precision = 1 if os.environ.get("MATMUL", "bf16") == "bf16" else 0
    def run():
        call("cswin_linear_bwd_weight_batch", ctypes.cast(wg, ctypes.c_void_p), 4, ctypes.cast(jobs, ctypes.c_void_p), None, 0, stream())
        call("cswin_rows_sum_multi", ctypes.cast(jobs, ctypes.c_void_p), 4, stream())
    t = timed(run)
    fl = sum(2.0 * M * N * K for N, K in shapes)
    tot += cnt * t
    print(f"stage {si+1}: batch of 4 + slab reduction {t*1e6:7.1f} us  ({fl/t/1e12:6.1f} TF/s)")
print(f"per step: {tot*1e3:.3f} ms")
