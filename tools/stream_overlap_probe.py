#!/usr/bin/env python3
"""Do dependent-launch boundaries of two independent kernel chains overlap?  Chains of small C-ABI kernels (LayerNorm forward on a
(rows, 256) tensor) on one stream vs split over two / four streams, eagerly and inside one hipGraph."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import cswin_unet_amd                                            # noqa: F401  (loads the library)
from cswin_unet_amd._lib import call, ptr

dev = torch.device("cuda:0")
N = 64


def bufs(rows):
    x = torch.randn(rows, 256, device=dev)
    return dict(x=x, g=torch.ones(256, device=dev), b=torch.zeros(256, device=dev), y=torch.empty_like(x),
                m=torch.empty(rows, device=dev), r=torch.empty(rows, device=dev), rows=rows)


def chain(B, n, st):
    s = torch.cuda.current_stream().cuda_stream if st is None else st.cuda_stream
    import ctypes
    for _ in range(n):
        call("cswin_layernorm_fwd", ptr(B["x"]), ptr(B["g"]), ptr(B["b"]), ptr(B["y"]), ptr(B["m"]), ptr(B["r"]), B["rows"], 256, 1e-5, 0,
             ctypes.c_void_p(s))


def run(rows, nstreams, graph):
    sets = [bufs(rows) for _ in range(nstreams)]
    streams = [torch.cuda.Stream() for _ in range(nstreams)]
    main = torch.cuda.Stream()

    def body():
        cur = torch.cuda.current_stream()
        for s in streams:
            s.wait_stream(cur)
        for B, s in zip(sets, streams):
            with torch.cuda.stream(s):
                chain(B, N // nstreams, s)
        for s in streams:
            cur.wait_stream(s)

    with torch.cuda.stream(main):
        body()
        torch.cuda.synchronize()
        if graph:
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, stream=main):
                body()
            fn = g.replay
        else:
            fn = body
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(20):
            fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / 20


for rows in (256, 4704, 18816):
    for graph in (False, True):
        ts = [run(rows, k, graph) for k in (1, 2, 4)]
        print(f"rows {rows:6d} {'graph' if graph else 'eager'}: {N} launches total on 1 / 2 / 4 streams: "
              + " / ".join(f"{t * 1e6:7.1f}" for t in ts) + " us", flush=True)
