#!/usr/bin/env python3
"""Does running a layer's data-gradient and weight-gradient GEMMs concurrently (two streams) beat running them back to back?
(feasibility probe for a grouped launch)  gemm_overlap.py"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cswin_unet_amd._lib import call, lib, ptr, precision
from ctypes import c_void_p

def timed(fn, reps=10, rounds=5):
    """device time per call: `reps` calls captured into one hipGraph (stream forks/joins included) and replayed"""
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

B = 24
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
for name, M, N, K in [("s3.qkv", 4704, 768, 256), ("s3.proj", 4704, 256, 256), ("s3.fc1", 4704, 1024, 256), ("s3.fc2", 4704, 256, 1024),
                      ("s2.fc1", 18816, 512, 128), ("s1.fc1", 75264, 256, 64), ("s4.fc1", 1176, 2048, 512)]:
    x = torch.randn(M, K, device="cuda"); w = torch.randn(N, K, device="cuda")
    dy = torch.randn(M, N, device="cuda"); dx = torch.empty(M, K, device="cuda")
    dw = torch.empty(N, K, device="cuda"); db = torch.empty(N, device="cuda")
    nbytes = lib().cswin_linear_bwd_weight_workspace(M, N, K); ws = torch.empty(nbytes // 4 + 4, device="cuda")
    def dgrad(st): call("cswin_linear_bwd_data", ptr(dy), ptr(w), ptr(dx), None, 0, None, None, 1, None, M, N, K, precision(), 0, c_void_p(st.cuda_stream))
    def wgrad(st): call("cswin_linear_bwd_weight", ptr(dy), ptr(x), None, 0, None, 1, ptr(dw), ptr(db), ptr(ws), nbytes, M, N, K, None, precision(), c_void_p(st.cuda_stream))
    def seq():
        cur = torch.cuda.current_stream()
        dgrad(cur); wgrad(cur)
    def par():
        cur = torch.cuda.current_stream()
        s2.wait_stream(cur)
        dgrad(cur); wgrad(s2)
        cur.wait_stream(s2)
    td, tw = timed(lambda: dgrad(torch.cuda.current_stream())), timed(lambda: wgrad(torch.cuda.current_stream()))
    ts, tp = timed(seq), timed(par)
    print(f"{name:8s} dgrad {td*1e6:6.1f} wgrad {tw*1e6:6.1f} sum {1e6*(td+tw):6.1f} | back-to-back {ts*1e6:6.1f} | two streams {tp*1e6:6.1f} us")
