#!/usr/bin/env python3
"""Headline benchmark: CSWin-UNet (cswin_tiny_224_lite) training images/sec on MI355X.

    python bench.py --gpus N --steps K --warmup W          (N > 1 without a launcher: spawns its own N ranks)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

One "step" = forward -> 0.4*CE + 0.6*Dice -> backward (-> RCCL all-reduce) -> SGD(momentum) update on a synthetic
224x224 9-class minibatch of 24 images per GPU (BASELINE.json configs[1]; weak scaling), fp32, inputs resident in HBM.
Rank 0 prints ONE JSON line.  At N=1 it also carries
  * "roofline": the stripe-attention kernels (the kernels north_star names), timed live with HIP events on the launch
    stream, algorithmic FLOPs from SURVEY.md 8(d): 4*L*N*C per block per image forward, x2 more for backward;
  * "roofline_dominant_by_time": the same measurement for the GEMM family (every Linear of the 26 blocks, fwd + dgrad +
    wgrad), which is ~73 % of the step;
  * "cpu_baseline": the CPU oracle's identical training step timed on this host's cores (kind "port").
"""
import argparse
import json
import os
import sys
import time

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_F32_MATRIX_TFLOPS = 157.3      # MI355X_MICROARCH.md: v_mfma_f32_* dense peak
PEAK_BF16_MATRIX_TFLOPS = 2500.0    # MI355X_MICROARCH.md: bf16 MFMA dense peak (~2.5 PF; 16 x the fp32 matrix rate)
PEAK_HBM_GBPS = 8000.0


def log(*a):
    print(*a, file=sys.stderr, flush=True)


def _graph_time(fn, reps=20, rounds=5):
    """Average device time of one call: `reps` launches captured in a hipGraph on the current stream, replayed between
    two HIP events on that same stream (host launch overhead excluded; best of `rounds`)."""
    fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(reps):
            fn()
    g.replay()
    torch.cuda.synchronize()
    best = float("inf")
    for _ in range(rounds):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        g.replay()
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) * 1e-3 / reps)
    return best


def attention_roofline(batch, cfg, img_size=224, bf16=False):
    """Times the fused stripe-attention kernels for the four stage shapes of the model at this batch through the C ABI.
    Algorithmic FLOPs: QK^T + PV = 4*L*N*C per block per image forward; backward = 2x that (dQ, dK, dV, dP).
    Algorithmic bytes: q,k,v in + y out = 16*L*C per block per image forward; backward q,k,v,dy in + dqkv out = 28*L*C."""
    import ctypes
    from cswin_unet_amd._lib import call, lib, ptr, stream, precision
    dev = "cuda"
    E, depth, heads, split = cfg.EMBED_DIM, cfg.DEPTH, cfg.NUM_HEADS, cfg.SPLIT_SIZE
    reso0 = img_size // 4
    rows, tot_flops, tot_time, tot_bytes = [], 0.0, 0.0, 0.0
    g = torch.Generator(device="cpu").manual_seed(7)
    for si in range(4):
        C, reso = E << si, reso0 >> si
        L = reso * reso
        single = si == 3 or reso == split[si]
        idx = [-1] if single else [0, 1]
        hb = [heads[si]] if single else [heads[si] // 2] * 2
        nb, cb = len(idx), C // len(idx)
        n_tok = reso * reso if single else reso * split[si]
        # bf16 mode: q / k / v / y / dqkv stored as bf16 and bf16 matrix instructions (attention storage mode 7), as the model runs it
        mode, adt = (7, torch.bfloat16) if bf16 else (0, torch.float32)
        qkv = torch.randn(batch, L, 3 * C, generator=g).to(dev).to(adt)
        w = [(torch.randn(cb, 9, generator=g) / 3).to(dev) for _ in idx]
        b = [(torch.randn(cb, generator=g) * 0.02).to(dev) for _ in idx]
        dy = torch.randn(batch, L, C, generator=g).to(dev)
        y = torch.empty(batch, L, C, device=dev, dtype=adt)
        y0 = torch.empty_like(y)                                     # P V without the LePE term: written by the forward for the backward
        lse = torch.empty(batch, sum(hb), L, device=dev)
        dqkv = torch.empty_like(qkv)
        dw, db = [torch.empty_like(t) for t in w], [torch.empty_like(t) for t in b]
        ia, ha = (ctypes.c_int * nb)(*idx), (ctypes.c_int * nb)(*hb)
        pa = lambda ts: (ctypes.c_void_p * nb)(*[t.data_ptr() for t in ts])
        nbytes = lib().cswin_attn_bwd_workspace(batch, reso, C, nb, ha, ia, split[si])
        ws = torch.empty(nbytes // 4 + 4, device=dev)
        t_f = _graph_time(lambda: call("cswin_attn_fwd", ptr(qkv), pa(w), pa(b), ptr(y), ptr(y0), ptr(lse), batch, reso, C, nb, ha, ia,
                                       split[si], 0.0, 0.0, 0, None, mode, stream()))
        t_b = _graph_time(lambda: call("cswin_attn_bwd", ptr(qkv), pa(w), pa(b), ptr(lse), ptr(y0), ptr(dy), ptr(dqkv), pa(dw), pa(db), ptr(ws),
                                       nbytes, batch, reso, C, nb, ha, ia, split[si], 0.0, None, 0.0, 0, None, mode, stream()))
        flops_f = 4.0 * L * n_tok * C * batch
        n_blocks = 2 * depth[si]
        bytes_f, bytes_b = 16.0 * L * C * batch, 28.0 * L * C * batch
        if bf16:                                   # q, k, v, y, dqkv at 2 B; dy stays fp32
            bytes_f, bytes_b = 8.0 * L * C * batch, 18.0 * L * C * batch
        rows.append({"stage": si + 1, "window_tokens": n_tok, "launches_per_step": n_blocks,
                     "fwd_us": round(t_f * 1e6, 2), "bwd_us": round(t_b * 1e6, 2),
                     "fwd_tflops": round(flops_f / t_f / 1e12, 2), "bwd_tflops": round(2 * flops_f / t_b / 1e12, 2),
                     "fwd_hbm_gbps": round(bytes_f / t_f / 1e9, 1), "bwd_hbm_gbps": round(bytes_b / t_b / 1e9, 1)})
        tot_flops += n_blocks * 3.0 * flops_f
        tot_time += n_blocks * (t_f + t_b)
        tot_bytes += n_blocks * (bytes_f + bytes_b)
    achieved = tot_flops / tot_time / 1e12
    # HBM bytes of the same 52 launches from PMC counters (separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes,
    # FETCH_SIZE doubled per MI355X_MICROARCH.md; tools/attn_pmc.sh -> profiles/round2_attn_pmc.json).  Only valid for
    # the profiled configuration (224x224, batch 24).
    traffic, traffic_src = None, None
    pmc = os.path.join(ROOT, "profiles", "round3_attn_pmc.json")
    if batch == 24 and img_size == 224 and os.path.exists(pmc) and not bf16:
        per = json.load(open(pmc))["per_launch"]
        traffic = int(sum(2 * depth[si] * (per[f"stage{si + 1}"]["fwd_bytes"] + per[f"stage{si + 1}"]["bwd_bytes"]) for si in range(4)))
        traffic_src = "STORED profile profiles/round3_attn_pmc.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of these kernels, tools/attn_pmc.sh), not measured in this run"
    return {"kernel": "attn_fwd3_kernel + attn_bwd3_kernel (+ the slab reduction launch of the LePE gradients), all 26 blocks of one step", "bound": "mfma",
            "achieved": round(achieved, 2), "peak": PEAK_BF16_MATRIX_TFLOPS if bf16 else PEAK_F32_MATRIX_TFLOPS, "unit": "TFLOP/s",
            "frac": round(achieved / (PEAK_BF16_MATRIX_TFLOPS if bf16 else PEAK_F32_MATRIX_TFLOPS), 4), "traffic": traffic,
            "matrix_instructions": "bf16 (v_mfma_f32_16x16x32_bf16 / 16x16x16_bf16), fp32 accumulate" if bf16 else "fp32 (v_mfma_f32_16x16x4_f32)",
            "traffic_note": "bytes per step (52 launches); algorithmic bytes per step = %d (q, k, v, dy in / y, dqkv out; the kernels also write and re-read y0 = P V, 8 L C more per block and image)" % int(tot_bytes),
            "traffic_source": traffic_src,
            "algorithmic_gflop_per_step": round(tot_flops / 1e9, 2), "time_per_step_ms": round(tot_time * 1e3, 3),
            "hbm_frac_algorithmic": round(tot_bytes / tot_time / 1e9 / PEAK_HBM_GBPS, 4), "per_stage": rows}



def gemm_family_roofline(batch, cfg, img_size=224):
    """The kernel family that dominates the step by time (~73 %): every Linear of the 26 CSWinBlocks, forward / data-gradient /
    weight-gradient, timed like the attention kernels (hipGraph replay between HIP events) through the C ABI and weighted by
    how often each shape runs in a step.  Algorithmic FLOPs = 2*M*N*K per GEMM."""
    from cswin_unet_amd._lib import call, lib, ptr, stream, precision
    E, depth = cfg.EMBED_DIM, cfg.DEPTH
    reso0 = img_size // 4
    g = torch.Generator(device="cpu").manual_seed(11)
    tot_t, tot_f = 0.0, 0.0
    for si in range(4):
        C, L = E << si, (reso0 >> si) ** 2
        M = batch * L
        for N, K in ((3 * C, C), (C, C), (4 * C, C), (C, 4 * C)):
            x, w, b = (torch.randn(*sh, generator=g).to("cuda") for sh in ((M, K), (N, K), (N,)))
            dy = torch.randn(M, N, generator=g).to("cuda")
            y, dx, dw, db = (torch.empty(*sh, device="cuda") for sh in ((M, N), (M, K), (N, K), (N,)))
            nbytes = lib().cswin_linear_bwd_weight_workspace(M, N, K)
            ws = torch.empty(nbytes // 4 + 4, device="cuda")
            t = _graph_time(lambda: call("cswin_linear_fwd", ptr(x), None, 0, ptr(w), ptr(b), ptr(y), None, None, None, 1, M, N, K, precision(), 0, stream()))
            t += _graph_time(lambda: call("cswin_linear_bwd_data", ptr(dy), ptr(w), ptr(dx), None, 0, None, None, 1, None, M, N, K, precision(), 0, stream()))
            t += _graph_time(lambda: call("cswin_linear_bwd_weight", ptr(dy), ptr(x), None, 0, None, 1, ptr(dw), ptr(db), ptr(ws), nbytes,
                                          M, N, K, None, precision(), stream()))
            tot_t += 2 * depth[si] * t
            tot_f += 2 * depth[si] * 3 * 2.0 * M * N * K
    achieved = tot_f / tot_t / 1e12
    return {"kernel": "gemm_kernel (Linear fwd + dgrad + wgrad incl. slab reduction, 16 shapes weighted by use: 312 launches of a step)",
            "bound": "mfma", "achieved": round(achieved, 2), "peak": PEAK_F32_MATRIX_TFLOPS, "unit": "TFLOP/s",
            "frac": round(achieved / PEAK_F32_MATRIX_TFLOPS, 4), "traffic": None,
            "algorithmic_gflop_per_step": round(tot_f / 1e9, 1), "time_per_step_ms": round(tot_t * 1e3, 3),
            "note": "peak is the 2.4 GHz data-sheet figure; an MFMA-only probe (tools/micro/mfma_peak.hip, profiles/round1_notes.md) sustained 137-146 TFLOP/s on this chip in round 1 -- not re-measured in this run"}

def cpu_baseline(batch, steps=2):
    """The CPU oracle's training step (same model, loss, optimiser, synthetic batch) on the host cores."""
    from oracle import cswin_oracle as O
    cores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    cores = max(1, min(cores, int(os.environ.get("CSWIN_CPU_THREADS", "16"))))   # a 1-GPU box's CPU share is 16 cores
    torch.set_num_threads(cores)
    log(f"[bench] cpu baseline: oracle training step, batch {batch}, {cores} threads ...")
    g = torch.Generator().manual_seed(1234)
    P = O.golden_params()
    img = torch.randn(batch, 1, 224, 224, generator=g)
    lab = torch.randint(0, 9, (batch, 224, 224), generator=g)
    M = {}
    O.train_step(P, M, img, lab, 0.05)          # warm-up
    log(f"[bench] cpu baseline: warm-up step done")
    t0 = time.perf_counter()
    for i in range(steps):
        O.train_step(P, M, img, lab, 0.05)
        log(f"[bench] cpu baseline: step {i + 1}/{steps} at {time.perf_counter() - t0:.1f}s")
    dt = (time.perf_counter() - t0) / steps
    return {"value": round(batch / dt, 3), "unit": "images/sec", "cores": cores, "kind": "port",
            "sample": f"{steps} timed training steps (+1 warm-up) of batch {batch}, cswin_tiny_224_lite fp32, "
                      f"torch-CPU oracle (oracle/cswin_oracle.py)", "ms_per_step": round(dt * 1e3, 1)}


def _spawn_ranks(n, argv):
    """`python bench.py --gpus N` without a launcher: start N fresh child processes (one rank per GPU, RCCL over xGMI) with
    the torchrun environment and relay rank 0's JSON line.  Decided BEFORE anything in this process touches the GPU: the
    parent never initialises HIP and never re-execs -- it only waits for its children and returns the worst exit code."""
    import socket
    import subprocess
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")       # dmabuf IPC only on this host driver (RCCL needs it)
        # rank 0 inherits stdout (the one JSON line); the other ranks' stdout goes to stderr
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + argv, env=env,
                                      stdout=None if r == 0 else sys.stderr))
    rc, deadline = 0, None
    try:
        pending = list(procs)
        while pending:
            for p in list(pending):
                code = p.poll()
                if code is None:
                    continue
                pending.remove(p)
                if code != 0 and rc == 0:
                    rc = code
                    deadline = time.time() + 20.0        # the other ranks get 20 s to fail (and say why) by themselves ...
            if deadline is not None and time.time() > deadline:
                for q in pending:                        # ... then they are blocked in a collective on the dead rank
                    q.terminate()
                deadline = time.time() + 1e9
            time.sleep(0.2)
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
    return rc


def _timed_steps(trainer, img, lab, steps, warmup, world, dev):
    """2 set-up steps (eager warm-up + hipGraph capture), `warmup` untimed steps, then EXACTLY `steps` steps between barrier +
    synchronize on both sides (wall clock, MAX over ranks).  Every timed step is also bracketed by HIP events on the trainer's
    stream: their per-step times give the median the survey asks for (8d) next to the mean the driver's contract defines."""
    for _ in range(2):
        trainer.train_step(img, lab)
    for _ in range(warmup):
        trainer.train_step(img, lab)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(steps + 1)]
    t0 = time.perf_counter()
    for i in range(steps):
        ev[i].record()
        stats = trainer.train_step(img, lab)
    ev[steps].record()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    step_ms = sorted(ev[i].elapsed_time(ev[i + 1]) for i in range(steps))
    return elapsed, step_ms, stats


def _median(v):
    n = len(v)
    return 0.5 * (v[(n - 1) // 2] + v[n // 2])


def bf16_mode_line(args, config, num_classes, dev):
    """BASELINE configs[2]'s per-GPU workload (same model, batch and step; bf16 MFMA for every Linear / conv and the attention
    products, bf16 block activations and weight shadow, fp32 accumulation / residual stream / statistics / master weights) for 20
    timed steps, after the fp32 headline has been measured: an extra key of the fp32 line, not a headline of its own."""
    import cswin_unet_amd
    from cswin_unet_amd.networks.vision_transformer import CSwinUnet
    from cswin_unet_amd.trainer import DataParallelTrainer, synthetic_batch
    prev = cswin_unet_amd.set_matmul_precision("bf16")
    try:
        torch.manual_seed(1234)
        model = CSwinUnet(config, img_size=config.DATA.IMG_SIZE, num_classes=num_classes).to(dev)
        model.train()
        trainer = DataParallelTrainer(model, num_classes, base_lr=0.05, max_iterations=1000, group=None, use_graph=not args.no_graph,
                                      allreduce_dtype=torch.bfloat16)
        img, lab = synthetic_batch(args.batch, config.DATA.IMG_SIZE, num_classes, 1234, dev)
        steps = 20
        elapsed, step_ms, stats = _timed_steps(trainer, img, lab, steps, 5, 1, dev)
        loss = [float(v) for v in stats.tolist()]
        out = {"workload": "BASELINE configs[2] per GPU: cswin_tiny_224_lite, synthetic 224x224 9-class, bs=%d, bf16" % args.batch,
               "dtype": "bf16 GEMM / attention operands + bf16 block activations / weight shadow, f32 accumulate, residual stream, statistics, master weights",
               "steps": steps, "warmup": 5, "ms_per_step": round(elapsed / steps * 1e3, 3), "ms_per_step_median": round(_median(step_ms), 3),
               "value": round(args.batch * steps / elapsed, 2), "unit": "images/sec",
               "final_loss": {"loss": round(loss[0], 5), "ce": round(loss[1], 5), "dice": round(loss[2], 5)}}
        if not args.skip_roofline:
            out["roofline"] = attention_roofline(args.batch, config.MODEL.CSWIN, config.DATA.IMG_SIZE, bf16=True)
        del trainer, model
        return out
    finally:
        cswin_unet_amd.set_matmul_precision(prev)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--batch", type=int, default=24, help="images per GPU")
    ap.add_argument("--cfg", default=os.path.join(ROOT, "configs", "cswin_tiny_224_lite.yaml"))
    ap.add_argument("--no-graph", action="store_true", help="launch kernels eagerly instead of replaying hipGraphs")
    ap.add_argument("--skip-roofline", action="store_true")
    ap.add_argument("--skip-cpu", action="store_true")
    ap.add_argument("--skip-bf16", action="store_true", help="do not append the bf16-mode measurement to the fp32 headline line")
    ap.add_argument("--cpu-batch", type=int, default=24)
    ap.add_argument("--matmul", choices=["fp32", "bf16"], default="fp32",
                    help="fp32: exact fp32 MFMA (BASELINE configs[1], the default and the headline); bf16 (configs[2..4]): bf16 MFMA "
                         "for every Linear / conv, block activations and a weight shadow stored as bf16, fp32 accumulation, "
                         "residual stream, attention arithmetic, statistics and master weights")
    ap.add_argument("--img-size", type=int, default=None, help="override DATA.IMG_SIZE (384 uses split [1,2,12,12])")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(_spawn_ranks(args.gpus, sys.argv[1:]))

    from cswin_unet_amd.config import get_config
    from cswin_unet_amd.networks.vision_transformer import CSwinUnet
    from cswin_unet_amd.trainer import DataParallelTrainer, init_distributed, synthetic_batch

    world_env = int(os.environ.get("WORLD_SIZE", "1"))
    if world_env != args.gpus:
        sys.exit(f"bench.py: --gpus {args.gpus} but the launcher set WORLD_SIZE={world_env}")
    n_dev = torch.cuda.device_count()                # counting devices does not initialise HIP
    if n_dev < args.gpus and not (n_dev >= 1 and os.environ.get("CSWIN_DIST_BACKEND") == "gloo"):   # gloo: ranks may share a GPU (rehearsal)
        sys.exit(f"bench.py rank {os.environ.get('RANK', '0')}: needs {args.gpus} HIP device(s), this host shows {n_dev} "
                 f"(there is no CPU path)")
    rank, local, world, group = init_distributed()
    dev = torch.device("cuda", torch.cuda.current_device())
    over = {}
    if args.img_size:
        over["DATA.IMG_SIZE"] = args.img_size
        if args.img_size == 384:
            over["MODEL.CSWIN.SPLIT_SIZE"] = [1, 2, 12, 12]           # the yaml split does not divide 24
    config = get_config(args.cfg, **over)
    import cswin_unet_amd
    cswin_unet_amd.set_matmul_precision(args.matmul)
    num_classes = 9
    torch.manual_seed(1234)
    model = CSwinUnet(config, img_size=config.DATA.IMG_SIZE, num_classes=num_classes).to(dev)
    model.train()
    total_steps = args.warmup + args.steps + 8
    trainer = DataParallelTrainer(model, num_classes, base_lr=0.05, max_iterations=max(total_steps, 1000), group=group,
                                  use_graph=not args.no_graph,
                                  allreduce_dtype=torch.bfloat16 if args.matmul == "bf16" else None)
    img, lab = synthetic_batch(args.batch, config.DATA.IMG_SIZE, num_classes, 1234 + rank, dev)

    log(f"[bench] rank {rank}/{world}: model built, capturing ...")
    elapsed, step_ms, stats = _timed_steps(trainer, img, lab, args.steps, args.warmup, world, dev)
    loss = [float(v) for v in stats.tolist()]
    log(f"[bench] timed region done: {elapsed / args.steps * 1e3:.3f} ms/step")
    dp_diag = None
    if world > 1:
        # outside the timed region: five more steps with a HIP event after every backward phase, around the wait for the
        # collectives and after the update, so that one multi-GPU run says where its time went (rank 0's view)
        trainer.enable_timing()
        for _ in range(5):
            trainer.train_step(img, lab)
        dp_diag = trainer.collect_timing()
        trainer.enable_timing(False)

    if rank == 0:
        ms = elapsed / args.steps * 1e3
        name = os.path.splitext(os.path.basename(args.cfg))[0]
        headline = config.DATA.IMG_SIZE == 224 and name == "cswin_tiny_224_lite"
        metric = "training images/sec (224x224, cswin_tiny)" if headline else \
            f"training images/sec ({config.DATA.IMG_SIZE}x{config.DATA.IMG_SIZE}, {name}) [not the headline configuration]"
        out = {"metric": metric, "value": round(world * args.batch * args.steps / elapsed, 2),
               "unit": "images/sec", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
               "ms_per_step": round(ms, 3), "ms_per_step_median": round(_median(step_ms), 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
               "dtype": "f32" if args.matmul == "fp32" else "bf16 GEMM operands + bf16 block activations / weight shadow, f32 accumulate, residual stream, attention math, master weights", "data": "synthetic",
               "config": {"workload": f"{os.path.splitext(os.path.basename(args.cfg))[0]}, synthetic {config.DATA.IMG_SIZE}x{config.DATA.IMG_SIZE} 9-class, "
                                      f"bs={args.batch}/GPU, {args.matmul} matmul, "
                                      "fwd + 0.4CE+0.6Dice + bwd + SGD(momentum) step, drop_path 0.2",
                          "global_batch": world * args.batch, "img_size": config.DATA.IMG_SIZE,
                          "parallelism": f"dp{world}", "hip_graph": not args.no_graph},
               "final_loss": {"loss": round(loss[0], 5), "ce": round(loss[1], 5), "dice": round(loss[2], 5)},
               "model_tflops": round(33.2e9 * world * args.batch * args.steps / elapsed / 1e12, 2) if config.DATA.IMG_SIZE == 224 and config.MODEL.CSWIN.EMBED_DIM == 64 else None}
        if dp_diag is not None:
            dp_diag["note"] = ("mean ms over 5 instrumented steps after the timed region (rank 0): per backward phase (loss + decoder | deep encoder | "
                               "shallow encoder; each phase's gradient buckets go on the wire when it ends), the time the compute stream then "
                               "waited for the collectives (+ bf16 unpack), and the SGD update")
            out["dp_diagnostics"] = dp_diag
        if world == 1 and not args.skip_roofline:
            log("[bench] attention roofline sub-benchmark ...")
            out["roofline"] = attention_roofline(args.batch, config.MODEL.CSWIN, config.DATA.IMG_SIZE, bf16=args.matmul == "bf16")
            if args.matmul == "fp32":
                log("[bench] GEMM family roofline sub-benchmark ...")
                out["roofline_dominant_by_time"] = gemm_family_roofline(args.batch, config.MODEL.CSWIN, config.DATA.IMG_SIZE)
        if world == 1 and headline and args.matmul == "fp32" and not args.skip_bf16:
            log("[bench] bf16 mode (BASELINE configs[2] per-GPU workload), 20 steps ...")
            del trainer
            out["bf16_mode"] = bf16_mode_line(args, config, num_classes, dev)
        if world == 1 and not args.skip_cpu:
            out["cpu_baseline"] = cpu_baseline(args.cpu_batch)
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
