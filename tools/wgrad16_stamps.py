#!/usr/bin/env python3
"""In-kernel timeline of the bf16 weight-gradient batch of one stage-3 CSWinBlock (wgrad16.hip), all operands stored as bf16."""
import ctypes, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import cswin_unet_amd
from cswin_unet_amd._lib import ReduceJob, WgradDesc, call, lib, stream
cswin_unet_amd.set_matmul_precision("bf16")
B, L, C = 24, 196, 256
if len(sys.argv) > 2:
    L, C = int(sys.argv[1]), int(sys.argv[2])
M = B * L
shapes = [(C, 4 * C), (4 * C, C), (C, C), (3 * C, C)]
wg, jobs, keep = (WgradDesc * 4)(), (ReduceJob * 4)(), []
for i, (N, K) in enumerate(shapes):
    dy = torch.randn(M, N, device="cuda").bfloat16(); x = torch.randn(M, K, device="cuda").bfloat16()
    dw = torch.empty(N, K, device="cuda"); db = torch.empty(N, device="cuda")
    nbytes = lib().cswin_linear_bwd_weight_workspace(M, N, K)
    ws = torch.empty(nbytes // 4 + 4, device="cuda")
    keep += [dy, x, dw, db, ws]
    wg[i].dy, wg[i].x, wg[i].row_scale, wg[i].dw, wg[i].dbias = dy.data_ptr(), x.data_ptr(), None, dw.data_ptr(), db.data_ptr()
    wg[i].workspace, wg[i].ws_bytes, wg[i].rows_per_sample, wg[i].M, wg[i].N, wg[i].K = ws.data_ptr(), nbytes, 1, M, N, K
    wg[i].precision, wg[i].io_bf16 = 1, 3
st = torch.zeros(65536, 8, dtype=torch.int64, device="cuda")
h = lib(); h.cswin_debug_set_stamps.argtypes = [ctypes.c_void_p]
def run():
    call("cswin_linear_bwd_weight_batch", ctypes.cast(wg, ctypes.c_void_p), 4, ctypes.cast(jobs, ctypes.c_void_p), None, 0, stream())
for _ in range(3): run()
torch.cuda.synchronize()
h.cswin_debug_set_stamps(ctypes.c_void_p(st.data_ptr()))
run(); torch.cuda.synchronize()
h.cswin_debug_set_stamps(None)
s = st.cpu().numpy(); s = s[s[:, 0] != 0]
print(f"wgrad16 batch M={M} C={C}: {len(s)} workgroups")
for name, a, b_ in (("first tile in LDS", 0, 1), ("main loop", 1, 2), ("slab + bias store", 2, 3), ("total", 0, 3)):
    d = s[:, b_] - s[:, a]
    print(f"  {name:18s} mean {d.mean():9.0f}  p10 {np.percentile(d,10):9.0f}  p90 {np.percentile(d,90):9.0f} cycles")
r0, r1 = s[:, 5], s[:, 6]; base = r0.min()
print("  realtime (us): last start %.2f, first end %.2f, last end %.2f; mean life %.2f" % ((r0.max()-base)/100, (r1.min()-base)/100, (r1.max()-base)/100, (r1-r0).mean()/100))
ev = np.concatenate([np.stack([r0, np.ones_like(r0)], 1), np.stack([r1, -np.ones_like(r1)], 1)])
ev = ev[np.argsort(ev[:, 0], kind="stable")]; alive = np.cumsum(ev[:, 1]); dur = np.diff(ev[:, 0])
print("  workgroups alive: max %d, time-average %.0f" % (alive.max(), (alive[:-1] * dur).sum() / max(dur.sum(), 1)))
