"""Data-parallel training step for CSWin-UNet on MI355X (counterpart of the reference's trainer.py:20-95).

Same objective and schedule as ``trainer_synapse``: loss = 0.4*CE + 0.6*Dice (trainer.py:55-57), SGD momentum 0.9 /
weight decay 1e-4 (:42), poly learning rate base_lr*(1 - it/max_it)^0.9 applied after the step (:61-63).  What changes is
the execution model, MI355X-first:

  * one process per GPU (torch.distributed, backend "nccl" = RCCL over xGMI) instead of the reference's single-process
    nn.DataParallel (:37-38): replicas are persistent, only gradients travel.
  * the reference computes the loss on the gathered GLOBAL batch; CE is linear in the per-rank means but soft Dice is
    not, so the 1 + 3*ncls partial sums are all-reduced before the Dice ratio is formed (28 floats) -- the objective
    is the reference's, not a per-rank Dice.
  * gradients are packed by one multi-tensor launch into a flat buffer which is all-reduced in a few large buckets
    (sized for 7 point-to-point xGMI links, not for many small NVSwitch messages) and consumed by one fused SGD launch.
  * the step is captured into four hipGraphs (forward + loss sums | loss + decoder backward + packing | encoder backward
    merge2..norm + packing | encoder backward patch-embed..stage2 + packing) with the collectives between them, so ~700
    kernel launches cost four graph launches on the host, the decoder's gradient all-reduce overlaps the encoder backward,
    the deep encoder's (stage 3/4: 92 % of the encoder bytes) overlaps the shallow encoder backward, and only the last
    ~2.7 MB bucket is exposed.

The protocol (which collectives, which scalings) lives in ``DataParallelTrainer`` and is device agnostic; the device work
lives in an *engine*.  ``HipEngine`` is the product (C ABI kernels, hipGraphs).  tests/test_dp_gloo.py drives the same
protocol with a CPU engine over gloo to check 2-rank == 1-rank-global-batch semantics without a GPU.
"""
import os

import torch
import torch.distributed as dist

from ._lib import call, lib, ptr, stream
from .optim import FlatSGD


def init_distributed():
    """(rank, local_rank, world, group).  torchrun / torch.distributed.run environment; world 1 needs nothing."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if torch.cuda.is_available():
        torch.cuda.set_device(local % max(torch.cuda.device_count(), 1))
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        # "nccl" IS RCCL on ROCm.  CSWIN_DIST_BACKEND=gloo lets several ranks share one GPU (rehearsals on a 1-GPU box:
        # RCCL refuses two ranks on one device); the protocol is the same, only the transport differs.
        backend = os.environ.get("CSWIN_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, local, world, (dist.group.WORLD if world > 1 else None)


def synthetic_batch(batch, img_size, num_classes, seed, device):
    """Synthetic stand-in for a Synapse minibatch (datasets/dataset_synapse.py:62-69 after RandomGenerator):
    image (B, 1, H, W) fp32 ~ N(0,1), label (B, H, W) int64 uniform in [0, num_classes)."""
    g = torch.Generator(device="cpu").manual_seed(seed)
    img = torch.randn(batch, 1, img_size, img_size, generator=g)
    lab = torch.randint(0, num_classes, (batch, img_size, img_size), generator=g)
    return img.to(device), lab.to(device)


def poly_lr(base_lr, iter_num, max_iterations):
    return base_lr * (1.0 - iter_num / max_iterations) ** 0.9


def scale_lr_for_batch(base_lr, batch_size):
    """train.py:104-105: base_lr *= batch_size / 24 only when batch_size != 24 and batch_size % 6 == 0 (per-GPU batch)."""
    return base_lr * batch_size / 24 if (batch_size != 24 and batch_size % 6 == 0) else base_lr


DECODER_PREFIXES = ("stage_up", "upsample", "concat_linear", "norm_up", "output")


class HipEngine:
    """Device side of one training step on this rank's MI355X: C-ABI kernels, optionally replayed from hipGraphs.

    Backward runs in two phases -- decoder half, then encoder half (the U-Net's boundary tensors are the bottleneck and
    the three skips) -- each ending with the multi-tensor packing of its gradients, so that the data-parallel protocol
    can put the decoder bucket on the wire (RCCL, own stream) while the encoder half is still computing."""

    def __init__(self, model, num_classes, lr, momentum, weight_decay, w_ce, w_dice, use_graph=True):
        self.model, self.ncls, self.w_ce, self.w_dice = model, num_classes, w_ce, w_dice
        self.core = model.cswin_unet if hasattr(model, "cswin_unet") else model
        names = [n for n, p in model.named_parameters() if p.requires_grad]
        self.opt = FlatSGD(model.parameters(), lr=lr, momentum=momentum, weight_decay=weight_decay)
        is_dec = [any(seg.startswith(DECODER_PREFIXES) for seg in n.split(".")[:2]) for n in names]
        self.n_enc = is_dec.index(True) if True in is_dec else len(names)
        self.split_backward = 0 < self.n_enc < len(names) and all(is_dec[self.n_enc:]) and hasattr(self.core, "forward_features")
        self.core.detach_decoder_inputs = self.split_backward
        # second cut inside the encoder, in front of merge2: parameters [n_mid, n_enc) = merge2, stage3, merge3, stage4, norm
        self.n_mid = next(i for i, n in enumerate(names) if "merge2" in n.split(".")[:2]) if self.split_backward else 0
        dev = self.opt.flat_param.device
        self.sums = torch.zeros(1 + 3 * num_classes, dtype=torch.float32, device=dev)
        self.stats = torch.zeros(3, dtype=torch.float32, device=dev)          # [loss, ce, dice] of the last step
        self._coef = torch.zeros(2 * num_classes, dtype=torch.float32, device=dev)
        # nn.Dropout with p > 0 (no reference config has one): the dropout kernels take their seeds from the HOST generator when the op
        # is called -- once, at capture, for a replayed graph -- and add the device-resident epoch counter (ops.dropout_epoch) to
        # them; the step advances that counter by one kernel inside graph A, so every replay draws fresh masks
        self._live_dropout = any(isinstance(m, torch.nn.Dropout) and m.p > 0 for m in model.modules())
        self.use_graph, self._graphs, self._logits = use_graph, None, None
        self._wire = {}

    # flat views the protocol all-reduces / broadcasts
    @property
    def flat_param(self):
        return self.opt.flat_param

    @property
    def flat_grad(self):
        return self.opt.flat_grad

    def set_lr(self, lr):
        self.opt.set_lr(lr)

    # ---- eager pieces -------------------------------------------------------------------------------------------
    def _forward_sums(self, img, lab):
        if self._live_dropout:
            from .ops import advance_dropout_epoch
            advance_dropout_epoch(img.device)           # inside graph A when captured: a new mask set per replayed step
        logits = self.model(img)
        B, ncls = logits.shape[:2]
        hw = logits.numel() // (B * ncls)
        nbytes = lib().cswin_loss_workspace(B, ncls, hw)
        ws = torch.empty(nbytes // 4 + 4, dtype=torch.float32, device=logits.device)
        call("cswin_loss_sums", ptr(logits.detach()), ptr(lab), ptr(self.sums), ptr(ws), nbytes, B, ncls, hw, 0, stream())
        return logits

    def _loss_grad(self, logits, lab, dice_grad_scale):
        B, ncls = logits.shape[:2]
        hw = logits.numel() // (B * ncls)
        dlogits = torch.empty_like(logits)
        call("cswin_loss_bwd", ptr(logits.detach()), ptr(lab), ptr(self._coef), None, ptr(dlogits),
             self.w_ce / float(B * hw), self.w_dice / ncls * dice_grad_scale, B, ncls, hw, 0, stream())
        return dlogits

    def _backward_decoder(self, logits, lab, dice_grad_scale):
        """loss backward + decoder half; returns the gradients of the boundary tensors."""
        dlogits = self._loss_grad(logits, lab, dice_grad_scale)
        params, n_enc = self.opt.params, self.n_enc
        self.opt.zero_grad()
        from .ops import engine_backward
        if not self.split_backward:
            with engine_backward(self.opt):
                grads = torch.autograd.grad([logits], params, [dlogits])
            for p, g in zip(params, grads):
                p.grad = g
            self.opt.gather_grads()
            return None
        core = self.core
        dec_in = list(core.dec_in)                    # detached leaves the decoder consumed (detach_decoder_inputs)
        with engine_backward(self.opt):
            grads = torch.autograd.grad([logits], params[n_enc:] + dec_in, [dlogits])
        for p, g in zip(params[n_enc:], grads):
            p.grad = g
        self.opt.gather_grads(n_enc, len(params))
        return [core.xb, core.x1, core.x2, core.x3], list(grads[len(params) - n_enc:])

    def _backward_encoder_deep(self, boundary):
        """merge2 .. norm (92 % of the encoder's parameters): from the bottleneck and the stage-3 skip back to the
        detached stage-2 output.  Returns what the shallow phase needs."""
        (xb, x1, x2, x3), (dxb, dx1, dx2, dx3) = boundary
        params, lo, hi = self.opt.params, self.n_mid, self.n_enc
        mid = self.core.enc_mid_in
        from .ops import engine_backward
        with engine_backward(self.opt):
            grads = torch.autograd.grad([xb, x3], params[lo:hi] + [mid], [dxb, dx3])
        for p, g in zip(params[lo:hi], grads):
            p.grad = g
        self.opt.gather_grads(lo, hi)
        return [x2, x1], [dx2 + grads[-1], dx1]

    def _backward_encoder(self, boundary, lo=0, hi=None):
        bound, dbound = boundary
        params = self.opt.params
        hi = self.n_enc if hi is None else hi
        from .ops import engine_backward
        with engine_backward(self.opt):
            grads = torch.autograd.grad(bound, params[lo:hi], dbound)
        for p, g in zip(params[lo:hi], grads):
            p.grad = g
        self.opt.gather_grads(lo, hi)

    def _phase_ranges(self):
        n = len(self.opt.params)
        if not self.split_backward:
            return [self.opt.flat_range(0, n)]
        if self.n_mid:
            return [self.opt.flat_range(self.n_enc, n), self.opt.flat_range(self.n_mid, self.n_enc), self.opt.flat_range(0, self.n_mid)]
        return [self.opt.flat_range(self.n_enc, n), self.opt.flat_range(0, self.n_enc)]

    def _capture(self, img, lab, dice_grad_scale):
        # PyTorch's capture recipe: a few eager iterations on a side stream (allocator state and autograd's
        # AccumulateGrad nodes then belong to a non-default stream), then capture all parts into one memory pool
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(3):
                logits = self._forward_sums(img, lab)
                self.finalize(lab.numel())
                boundary = self._backward_decoder(logits, lab, dice_grad_scale)
                if boundary is not None:
                    self._run_encoder_phases(boundary)
        torch.cuda.current_stream().wait_stream(side)
        self._img, self._lab = img.clone(), lab.clone()
        self.opt.zero_grad()
        graphs = [torch.cuda.CUDAGraph()]
        # thread_local: RCCL's watchdog thread polls events of earlier collectives while we capture; its calls must not
        # invalidate the capture (the default "global" mode polices every thread of the process)
        mode = dict(capture_error_mode="thread_local")
        with torch.cuda.graph(graphs[0], **mode):
            logits = self._forward_sums(self._img, self._lab)
        graphs.append(torch.cuda.CUDAGraph())
        with torch.cuda.graph(graphs[1], pool=graphs[0].pool(), **mode):
            boundary = self._backward_decoder(logits, self._lab, dice_grad_scale)
        if boundary is not None:
            for phase in self._encoder_phases(boundary):
                graphs.append(torch.cuda.CUDAGraph())
                with torch.cuda.graph(graphs[-1], pool=graphs[0].pool(), **mode):
                    phase()
        self._graphs = graphs

    def _encoder_phases(self, boundary):
        """The encoder half as callables, one per all-reduce bucket boundary (deep part first: it owns 92 % of the bytes)."""
        if not self.n_mid:
            return [lambda: self._backward_encoder(boundary)]
        state = {}

        def deep():
            state["shallow"] = self._backward_encoder_deep(boundary)

        def shallow():
            self._backward_encoder(state["shallow"], 0, self.n_mid)

        return [deep, shallow]

    def _run_encoder_phases(self, boundary):
        for phase in self._encoder_phases(boundary):
            phase()

    # ---- protocol hooks -----------------------------------------------------------------------------------------
    def forward_sums(self, img, lab, dice_grad_scale):
        """Forward + local loss partial sums -> self.sums (device)."""
        if self.use_graph and self._graphs is None:
            self._capture(img, lab, dice_grad_scale)
        if self._graphs is not None:
            if img.data_ptr() != self._img.data_ptr():
                self._img.copy_(img)
                self._lab.copy_(lab)
            # a replay runs no Python: a weight written from outside the step (load_state_dict, load_from, copy_ on a parameter)
            # since the last one is noticed here by its version counter and answered with one re-pack of the bf16 shadow
            from ._lib import shadows_current
            shadows_current(self.opt)
            self._graphs[0].replay()
        else:
            self._logits = self._forward_sums(img, lab)
            self._lab_eager = lab

    def finalize(self, n_pixels_global):
        """sums (already all-reduced) -> stats [loss, ce, dice] and the Dice gradient coefficients."""
        call("cswin_loss_finalize", ptr(self.sums), ptr(self.stats), ptr(self._coef), float(n_pixels_global), self.ncls,
             self.w_ce, self.w_dice, None, stream())

    def backward_phases(self, dice_grad_scale):
        """Generator: runs one backward phase per iteration and yields the [lo, hi) range of flat_grad it completed."""
        ranges = self._phase_ranges()
        if self._graphs is not None:
            for g, r in zip(self._graphs[1:], ranges):
                g.replay()
                yield r
        else:
            boundary = self._backward_decoder(self._logits, self._lab_eager, dice_grad_scale)
            yield ranges[0]
            if boundary is not None:
                for phase, r in zip(self._encoder_phases(boundary), ranges[1:]):
                    phase()
                    yield r
            self._logits = None

    def apply(self, grad_scale):
        self.opt.apply(grad_scale)

    # bf16 gradient wire (HIP kernels; the wire buffer of a bucket is allocated once and reused every step)
    def pack_wire(self, chunk, dtype, scale=1.0):
        if dtype != torch.bfloat16:
            raise ValueError(f"HipEngine wire dtype {dtype}: only torch.bfloat16 is implemented")
        key = (chunk.data_ptr(), chunk.numel())
        wire = self._wire.get(key)
        if wire is None:
            wire = self._wire[key] = torch.empty(chunk.numel(), dtype=torch.bfloat16, device=chunk.device)
        call("cswin_pack_bf16_scaled", ptr(chunk), ptr(wire), chunk.numel(), float(scale), stream())
        return wire

    def unpack_wire(self, wire, chunk):
        call("cswin_unpack_bf16", ptr(wire), ptr(chunk), chunk.numel(), stream())


class DataParallelTrainer:
    """The data-parallel protocol of one step (device agnostic; see module docstring)."""

    def __init__(self, model=None, num_classes=9, base_lr=0.05, max_iterations=1000, momentum=0.9, weight_decay=1e-4,
                 group=None, use_graph=True, buckets=2, w_ce=0.4, w_dice=0.6, engine=None, force_collectives=False,
                 allreduce_dtype=None):
        self.group = group
        self.world = dist.get_world_size(group) if group is not None else 1
        self.base_lr, self.max_iterations, self.iter_num = base_lr, max_iterations, 0
        self.engine = engine if engine is not None else HipEngine(model, num_classes, base_lr, momentum, weight_decay,
                                                                 w_ce, w_dice, use_graph)
        self.model = model
        self.nbuckets = max(1, buckets)
        # torch.bfloat16: gradients travel as bf16 (47 MB instead of 94 MB per step, BASELINE configs[2]); the sum is formed
        # in bf16 by the collective, the fp32 master weights and momentum are untouched.  None: fp32 on the wire.
        self.allreduce_dtype = allreduce_dtype
        # run the collectives even with one rank (they are identities then): lets a 1-GPU box exercise the RCCL path
        self.collectives = self.world > 1 or (force_collectives and group is not None)
        if self.collectives:                    # identical replicas: rank 0's initial weights everywhere
            dist.broadcast(self.engine.flat_param, src=0, group=group)
            if hasattr(self.engine, "opt"):
                self.engine.opt.refresh_shadow()        # the broadcast wrote the flat buffer directly (no parameter version moves)

    @property
    def stats(self):
        return self.engine.stats

    # ---- per-phase timing of the backward / all-reduce / update part of a step (bench.py's dp_diagnostics) ----
    _timing = None

    def _mark(self):
        e = torch.cuda.Event(enable_timing=True)
        e.record()
        return e

    def enable_timing(self, on=True):
        """Record a HIP event on the compute stream after the loss, after every backward phase, before and after the loop that
        makes the compute stream wait for the collectives (+ the bf16 unpack kernels) and after the update, for the steps that
        follow.  The wait-loop interval is the all-reduce time the backward did NOT hide."""
        self._timing = {"ev": [], "bucket_bytes": []} if on else None

    def collect_timing(self):
        """Mean milliseconds per recorded step: backward phases (in order), exposed all-reduce (+ unpack), update."""
        tm = self._timing
        if not tm or not tm["ev"]:
            return None
        torch.cuda.synchronize()
        rows = [[a.elapsed_time(b) for a, b in zip(ev[:-1], ev[1:])] for ev in tm["ev"]]
        mean = [sum(r[i] for r in rows) / len(rows) for i in range(len(rows[0]))]
        nph = len(mean) - 3          # events: loss | after each phase (its buckets enqueued) | before the waits | after them | after the update
        return {"steps": len(rows), "backward_phase_ms": [round(v, 3) for v in mean[:nph]],
                "allreduce_exposed_ms": round(mean[nph] + mean[nph + 1], 3), "update_ms": round(mean[nph + 2], 3),
                "bucket_bytes": tm["bucket_bytes"], "wire": "bf16 (pre-divided by world)" if self.allreduce_dtype is not None else "fp32",
                "world": self.world}

    def train_step(self, img, lab):
        """One optimisation step on this rank's shard.  Returns the device tensor [loss, ce, dice] (no host sync)."""
        if lab.dtype != torch.int64:
            lab = lab.long()
        lab = lab.contiguous()
        eng, world = self.engine, self.world
        # Dice is a function of GLOBAL sums; gradients are averaged over ranks afterwards, so the local Dice gradient
        # (already built from global coefficients) is pre-multiplied by world to survive the 1/world averaging.
        eng.forward_sums(img, lab, dice_grad_scale=float(world))
        if self.collectives:
            dist.all_reduce(eng.sums, group=self.group)                 # 1 + 3*ncls floats
        eng.finalize(lab.numel() * world)
        tm = self._timing
        if tm is not None:
            tm["ev"].append([self._mark()])
        works = []
        wired = self.collectives and self.allreduce_dtype is not None
        for lo, hi in eng.backward_phases(dice_grad_scale=float(world)):
            if tm is not None:
                tm["ev"][-1].append(self._mark())
            if self.collectives:
                # this phase's gradients are final: put them on the wire now (RCCL runs on its own stream, ordered after
                # the work enqueued so far) while the next backward phase computes.  Few large buckets: xGMI is 7
                # point-to-point links, per-message latency matters more than on a switched fabric.
                g = eng.flat_grad
                step = max((hi - lo + self.nbuckets - 1) // self.nbuckets, 1 << 22)      # never below 16 MB per message
                step = (step + 63) // 64 * 64                                             # buckets start 256-B aligned
                for o in range(lo, hi, step):
                    chunk = g[o:min(o + step, hi)]
                    if tm is not None and len(tm["ev"]) == 1:
                        tm["bucket_bytes"].append(chunk.numel() * (2 if wired else 4))
                    if not wired:
                        works.append((dist.all_reduce(chunk, group=self.group, async_op=True), None, None))
                    else:
                        # bf16 wire: the bucket is packed PRE-DIVIDED by the world size, so the collective's bf16 sum is the mean
                        # itself (no mantissa bits spent on a factor that is divided out again) and apply() below scales by 1
                        pack = getattr(eng, "pack_wire", None)          # HipEngine: HIP pack kernel into a persistent wire buffer
                        wire = pack(chunk, self.allreduce_dtype, 1.0 / world) if pack else (chunk / world).to(self.allreduce_dtype)
                        works.append((dist.all_reduce(wire, group=self.group, async_op=True), chunk, wire))
        if tm is not None:
            tm["ev"][-1].append(self._mark())
        for w, chunk, wire in works:
            w.wait()
            if wire is not None:
                unpack = getattr(eng, "unpack_wire", None)
                if unpack:
                    unpack(wire, chunk)
                else:
                    chunk.copy_(wire)
        if tm is not None:
            tm["ev"][-1].append(self._mark())
        eng.apply(grad_scale=1.0 if wired else 1.0 / world)
        if tm is not None:
            tm["ev"][-1].append(self._mark())
        self.iter_num += 1
        eng.set_lr(poly_lr(self.base_lr, self.iter_num - 1, self.max_iterations))   # trainer.py:61-63
        return eng.stats

    def state_dict(self):
        """Reference checkpoint format: the model's state_dict (trainer.py:84)."""
        return self.model.state_dict()


class _Prefetcher:
    """Pinned host batches -> device on a side stream, one batch ahead of the step that consumes them."""

    def __init__(self, loader, device):
        self.it, self.device = iter(loader), device
        self.stream = torch.cuda.Stream(device)
        self._next()

    def _next(self):
        try:
            batch = next(self.it)
        except StopIteration:
            self.batch = None
            return
        with torch.cuda.stream(self.stream):
            self.batch = (batch['image'].to(self.device, non_blocking=True), batch['label'].to(self.device, non_blocking=True))

    def __iter__(self):
        return self

    def __next__(self):
        if self.batch is None:
            raise StopIteration
        torch.cuda.current_stream(self.device).wait_stream(self.stream)
        img, lab = self.batch
        img.record_stream(torch.cuda.current_stream(self.device))
        lab.record_stream(torch.cuda.current_stream(self.device))
        self._next()
        return img, lab


def trainer_synapse(args, model, snapshot_path, group=None, log_every=1):
    """Counterpart of the reference's ``trainer_synapse(args, model, snapshot_path)`` (trainer.py:20-95) on the HIP engine.

    args: root_path, list_dir, img_size, num_classes, batch_size (per GPU), base_lr, max_epochs, optionally num_workers.
    Same dataset / augmentation / loss / optimiser / LR schedule / checkpoint schedule (``epoch_N.pth`` every third epoch
    of the second half and at the end, :79-90).  Differences, all execution-side: one process per GPU (pass the process
    group; each rank reads its own shard through a DistributedSampler) instead of nn.DataParallel; batches are prefetched
    to the device on a side stream; the last incomplete batch of an epoch is dropped because the step is a captured
    hipGraph with a fixed batch shape; tensorboard image logging is not reproduced."""
    import logging
    import random
    import sys

    from torch.utils.data import DataLoader
    from torch.utils.data.distributed import DistributedSampler

    from .checkpoint import save_checkpoint
    from .datasets import RandomGenerator, Synapse_dataset

    os.makedirs(snapshot_path, exist_ok=True)
    logging.basicConfig(filename=os.path.join(snapshot_path, "log.txt"), level=logging.INFO,
                        format='[%(asctime)s.%(msecs)03d] %(message)s', datefmt='%H:%M:%S')
    if not any(isinstance(h, logging.StreamHandler) and getattr(h, "stream", None) is sys.stdout for h in logging.getLogger().handlers):
        logging.getLogger().addHandler(logging.StreamHandler(sys.stdout))
    logging.info(str(args))
    rank = dist.get_rank(group) if group is not None else 0
    world = dist.get_world_size(group) if group is not None else 1
    device = next(model.parameters()).device
    db_train = Synapse_dataset(base_dir=args.root_path, list_dir=args.list_dir, split="train",
                               transform=RandomGenerator(output_size=[args.img_size, args.img_size]))
    print("The length of train set is: {}".format(len(db_train)))
    seed = getattr(args, "seed", 1234)

    n_workers = getattr(args, "num_workers", 8)

    def worker_init_fn(worker_id):
        # trainer.py:33-34 seeds `random` only, with seed + worker_id.  Under data parallelism every rank would then share
        # one augmentation coin-flip stream, so the seed is offset by rank * num_workers (rank 0 = the reference's seeds).
        random.seed(seed + rank * max(n_workers, 1) + worker_id)

    sampler = DistributedSampler(db_train, num_replicas=world, rank=rank, shuffle=True, seed=seed) if world > 1 else None
    loader = DataLoader(db_train, batch_size=args.batch_size, shuffle=sampler is None, sampler=sampler,
                        num_workers=n_workers, pin_memory=True, drop_last=True, worker_init_fn=worker_init_fn,
                        persistent_workers=False)    # like the reference: workers are re-created and re-seeded every epoch
    max_epoch = args.max_epochs
    max_iterations = max_epoch * len(loader)
    if max_iterations == 0:
        raise ValueError(f"trainer_synapse: {len(db_train)} samples give no full batch of {args.batch_size} per rank "
                         f"(the captured step has a fixed batch shape, the last incomplete batch is dropped)")
    logging.info("{} iterations per epoch. {} max iterations ".format(len(loader), max_iterations))
    model.train()
    trainer = DataParallelTrainer(model, args.num_classes, base_lr=args.base_lr, max_iterations=max_iterations, group=group)
    iter_num = 0
    for epoch_num in range(max_epoch):
        if sampler is not None:
            sampler.set_epoch(epoch_num)
        for image_batch, label_batch in _Prefetcher(loader, device):
            stats = trainer.train_step(image_batch, label_batch)
            iter_num += 1
            if log_every and iter_num % log_every == 0 and rank == 0:
                loss, loss_ce, _ = stats.tolist()                    # the only host sync of the loop
                logging.info('iteration %d : loss : %f, loss_ce: %f' % (iter_num, loss, loss_ce))
        save_interval = 3
        last = epoch_num >= max_epoch - 1
        if rank == 0 and (last or (epoch_num > int(max_epoch / 2) and (epoch_num + 1) % save_interval == 0)):
            save_mode_path = os.path.join(snapshot_path, 'epoch_' + str(epoch_num) + '.pth')
            save_checkpoint(model, save_mode_path)
            logging.info("save model to {}".format(save_mode_path))
    return "Training Finished!"
