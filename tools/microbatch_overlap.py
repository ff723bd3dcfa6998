#!/usr/bin/env python3
"""Do two half-batch GEMM chains on two streams beat one full-batch chain?  Stage-3 forward Linears (qkv, proj, fc1+GELU, fc2)
of a CSWinBlock, B = 24 on one stream vs 2 x B = 12 on two streams, both captured in one hipGraph (10 blocks deep)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cswin_unet_amd._lib import call, ptr, stream, precision

def chain_buffers(M, C):
    mk = lambda *s: torch.randn(*s, device="cuda")
    return dict(x=mk(M, C), qkv=mk(M, 3 * C), att=mk(M, C), x1=mk(M, C), pre=mk(M, 4 * C), act=mk(M, 4 * C), y=mk(M, C),
                dy=mk(M, C), dpre=mk(M, 4 * C), dh=mk(M, C))

def run_chain(b, w, M, C, bwd):
    s = stream()
    call("cswin_linear_fwd", ptr(b["x"]), None, 0, ptr(w["qkv"]), ptr(w["bqkv"]), ptr(b["qkv"]), None, None, None, 1, M, 3 * C, C, precision(), 0, s)
    call("cswin_linear_fwd", ptr(b["att"]), None, 0, ptr(w["proj"]), ptr(w["bp"]), ptr(b["x1"]), None, ptr(b["x"]), None, 1, M, C, C, precision(), 0, s)
    call("cswin_linear_fwd", ptr(b["x1"]), None, 0, ptr(w["fc1"]), ptr(w["b1"]), ptr(b["pre"]), ptr(b["act"]), None, None, 1, M, 4 * C, C, precision(), 0, s)
    call("cswin_linear_fwd", ptr(b["act"]), None, 0, ptr(w["fc2"]), ptr(w["b2"]), ptr(b["y"]), None, ptr(b["x1"]), None, 1, M, C, 4 * C, precision(), 0, s)
    if bwd:
        call("cswin_linear_bwd_data", ptr(b["dy"]), ptr(w["fc2"]), ptr(b["dpre"]), None, 0, ptr(b["pre"]), None, 1, None, M, C, 4 * C, precision(), 0, s)
        call("cswin_linear_bwd_data", ptr(b["dpre"]), ptr(w["fc1"]), ptr(b["dh"]), None, 0, None, None, 1, None, M, 4 * C, C, precision(), 0, s)
        call("cswin_linear_bwd_data", ptr(b["dh"]), ptr(w["proj"]), ptr(b["att"]), None, 0, None, None, 1, None, M, C, C, precision(), 0, s)
        call("cswin_linear_bwd_data", ptr(b["qkv"]), ptr(w["qkv"]), ptr(b["dh"]), None, 0, None, None, 1, None, M, 3 * C, C, precision(), 0, s)

def timed_graph(fn, reps=5):
    fn(); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        fn()
    g.replay(); torch.cuda.synchronize()
    best = 1e9
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) * 1e3)
    return best

OFFSET = os.environ.get("OFFSET", "0") != "0"
for L, C in ((196, 256), (784, 128), (3136, 64)):
    w = dict(qkv=torch.randn(3 * C, C, device="cuda"), bqkv=torch.randn(3 * C, device="cuda"), proj=torch.randn(C, C, device="cuda"),
             bp=torch.randn(C, device="cuda"), fc1=torch.randn(4 * C, C, device="cuda"), b1=torch.randn(4 * C, device="cuda"),
             fc2=torch.randn(C, 4 * C, device="cuda"), b2=torch.randn(C, device="cuda"))
    full = chain_buffers(24 * L, C)
    halves = [chain_buffers(12 * L, C) for _ in range(2)]
    thirds = [chain_buffers(8 * L, C) for _ in range(3)]
    side = [torch.cuda.Stream() for _ in range(2)]
    for bwd in (False, True):
        def one():
            for _ in range(10): run_chain(full, w, 24 * L, C, bwd)
        def multi(parts, Mp):
            def f():
                main = torch.cuda.current_stream()
                streams = [main] + side[:len(parts) - 1]
                for st in streams[1:]: st.wait_stream(main)
                for si_, (st, b) in enumerate(zip(streams, parts)):
                    with torch.cuda.stream(st):
                        if OFFSET and si_ > 0:          # de-phase the chains: an extra half-size GEMM in front of the later streams
                            for _ in range(si_):
                                call("cswin_linear_fwd", ptr(b["x"]), None, 0, ptr(w["proj"]), ptr(w["bp"]), ptr(b["dh"]), None, None, None, 1, Mp, C, C, precision(), 0, stream())
                        for _ in range(10): run_chain(b, w, Mp, C, bwd)
                for st in streams[1:]: main.wait_stream(st)
            return f
        t1 = timed_graph(one)
        t2 = timed_graph(multi(halves, 12 * L))
        t3 = timed_graph(multi(thirds, 8 * L))
        print(f"L={L} C={C} {'fwd+dgrad' if bwd else 'fwd'}: one stream B=24 {t1/10:7.1f} us/block | 2 streams x B=12 {t2/10:7.1f} | 3 streams x B=8 {t3/10:7.1f}", flush=True)
