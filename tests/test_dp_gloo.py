"""Data-parallel protocol on CPU: 2 ranks over gloo == one process on the global batch.

Drives cswin_unet_amd.trainer.DataParallelTrainer (the product's protocol: which sums are all-reduced, how gradients
are scaled and averaged, poly-LR) with a CPU engine built on the oracle, because the HIP engine needs a GPU.
Checks the reference's DataParallel semantics (loss on the gathered global batch, trainer.py:54-57): after 2 steps
the 2-rank run and the 1-rank run on the concatenated batch have the same loss, Dice and weights."""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp
import torch.nn.functional as F

from oracle import cswin_oracle as O
from oracle.determ import det_labels, det_normal

CFG = dict(img_size=64, embed_dim=32, depth=(1, 1, 1, 1), split_size=(1, 1, 2, 2), num_heads=(2, 2, 4, 8), num_classes=4)


class OracleEngine:
    """CPU stand-in for HipEngine with the same hooks (forward_sums / finalize / backward_pack / apply / set_lr)."""

    def __init__(self, lr, w_ce=0.4, w_dice=0.6, momentum=0.9, weight_decay=1e-4):
        self.P = O.golden_params(CFG)
        self.names = sorted(self.P)
        self.ncls, self.w_ce, self.w_dice = CFG["num_classes"], w_ce, w_dice
        self.lr, self.momentum, self.wd = lr, momentum, weight_decay
        n = sum(self.P[k].numel() for k in self.names)
        self.flat_param = torch.cat([self.P[k].detach().reshape(-1) for k in self.names])
        self.flat_grad = torch.zeros(n)
        self.flat_mom = torch.zeros(n)
        self.sums = torch.zeros(1 + 3 * self.ncls)
        self.stats = torch.zeros(3)

    def _sync_params(self):
        o = 0
        with torch.no_grad():
            for k in self.names:
                p = self.P[k]
                p.copy_(self.flat_param[o:o + p.numel()].view_as(p))
                o += p.numel()

    def set_lr(self, lr):
        self.lr = lr

    def forward_sums(self, img, lab, dice_grad_scale):
        self._sync_params()
        self.logits = O.cswin_forward(self.P, img, CFG)
        self.lab = lab
        self.local = O.dice_sums(self.logits, lab, self.ncls)                          # (3, ncls), differentiable
        ce_sum = F.cross_entropy(self.logits, lab, reduction="sum")
        self.sums.copy_(torch.cat([ce_sum.detach().reshape(1), self.local.detach().reshape(-1)]))

    def finalize(self, n_pixels_global):
        s = self.sums
        glob = s[1:].view(3, self.ncls)
        ce = s[0] / n_pixels_global
        dice = O.dice_from_sums(glob)
        self.stats.copy_(torch.stack([self.w_ce * ce + self.w_dice * dice, ce, dice]))
        self.glob = glob.clone()

    def backward_phases(self, dice_grad_scale):
        self._backward(dice_grad_scale)
        half = self.flat_grad.numel() // 2          # two phases, to exercise the per-phase bucketed all-reduce
        yield 0, half
        yield half, self.flat_grad.numel()

    def _backward(self, dice_grad_scale):
        # Dice evaluated at the GLOBAL sums, differentiated through this rank's contribution only
        s = self.local + (self.glob - self.local).detach()
        loss = self.w_ce * F.cross_entropy(self.logits, self.lab) + self.w_dice * dice_grad_scale * O.dice_from_sums(s)
        grads = torch.autograd.grad(loss, [self.P[k] for k in self.names])
        self.flat_grad.copy_(torch.cat([g.reshape(-1) for g in grads]))

    def apply(self, grad_scale):
        g = self.flat_grad * grad_scale + self.wd * self.flat_param
        self.flat_mom.mul_(self.momentum).add_(g)
        self.flat_param.sub_(self.lr * self.flat_mom)


def _run(rank, world, port, out, wire_dtype=None, batch=2):
    from cswin_unet_amd.trainer import DataParallelTrainer
    group = None
    if world > 1:
        dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
        group = dist.group.WORLD
    torch.set_num_threads(2)
    img = torch.from_numpy(det_normal("dp.img", (batch, 3, 64, 64)))
    lab = torch.from_numpy(det_labels("dp.lab", (batch, 64, 64), CFG["num_classes"]))
    if world > 1:
        per = batch // world
        img, lab = img[rank * per:(rank + 1) * per], lab[rank * per:(rank + 1) * per]
    tr = DataParallelTrainer(engine=OracleEngine(lr=0.05), base_lr=0.05, max_iterations=10, group=group, buckets=3,
                             allreduce_dtype=wire_dtype)
    hist = []
    for _ in range(2):
        hist.append(tr.train_step(img, lab).clone())
    res = {"stats": torch.stack(hist).numpy(), "w": tr.engine.flat_param.numpy().copy(), "lr": tr.engine.lr}
    if rank == 0:
        out.put(res)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def test_two_ranks_equal_global_batch():
    ctx = mp.get_context("spawn")
    q1, q2 = ctx.Queue(), ctx.Queue()
    _run(0, 1, 0, q1)
    single = q1.get(timeout=60)
    port = _free_port()
    procs = [ctx.Process(target=_run, args=(r, 2, port, q2), daemon=True) for r in range(2)]
    for p in procs:
        p.start()
    try:
        multi = q2.get(timeout=240)
    finally:
        for p in procs:
            p.join(60)
            if p.is_alive():
                p.kill()
    assert all(p.exitcode == 0 for p in procs), [p.exitcode for p in procs]
    assert np.allclose(single["stats"], multi["stats"], rtol=2e-5, atol=1e-6), (single["stats"], multi["stats"])
    assert np.allclose(single["w"], multi["w"], rtol=1e-4, atol=2e-6)
    assert abs(single["lr"] - 0.05 * (1 - 1 / 10) ** 0.9) < 1e-12 and single["lr"] == multi["lr"]
    assert single["stats"][1, 0] != single["stats"][0, 0]            # the step really changed the loss


def test_two_ranks_bf16_gradient_wire():
    """allreduce_dtype=bfloat16 (BASELINE configs[2]: bf16 gradients on the wire): same protocol, gradients rounded to
    8 mantissa bits before the sum; the weights after 2 steps stay within that rounding of the fp32-wire run."""
    ctx = mp.get_context("spawn")
    q1, q2 = ctx.Queue(), ctx.Queue()
    _run(0, 1, 0, q1)
    single = q1.get(timeout=60)
    port = _free_port()
    procs = [ctx.Process(target=_run, args=(r, 2, port, q2, torch.bfloat16), daemon=True) for r in range(2)]
    for p in procs:
        p.start()
    try:
        multi = q2.get(timeout=240)
    finally:
        for p in procs:
            p.join(60)
            if p.is_alive():
                p.kill()
    assert all(p.exitcode == 0 for p in procs), [p.exitcode for p in procs]
    assert np.allclose(single["stats"][0], multi["stats"][0], rtol=2e-5, atol=1e-6)      # first loss: no gradient used yet
    dw = np.abs(single["w"] - multi["w"]).max()
    step = np.abs(single["w"]).max() * 0 + 0.05                                            # lr: one step moves a weight by <= lr * |g|
    assert 0 < dw < 2 ** -7 * step * 40, dw                                               # bf16 rounding of the update, not garbage
    assert np.allclose(single["stats"][1], multi["stats"][1], rtol=5e-3)



def _spawn(world, wire_dtype, batch):
    ctx = mp.get_context("spawn")
    q1, q2 = ctx.Queue(), ctx.Queue()
    _run(0, 1, 0, q1, None, batch)
    single = q1.get(timeout=120)
    port = _free_port()
    procs = [ctx.Process(target=_run, args=(r, world, port, q2, wire_dtype, batch), daemon=True) for r in range(world)]
    for p in procs:
        p.start()
    try:
        multi = q2.get(timeout=480)
    finally:
        for p in procs:
            p.join(120)
            if p.is_alive():
                p.kill()
    assert all(p.exitcode == 0 for p in procs), [p.exitcode for p in procs]
    return single, multi


def test_eight_ranks_equal_global_batch():
    """The driver's SCALE run is world 8: the same protocol (28-float global-Dice all-reduce, bucket ranges of three phases, 1 / world
    averaging, poly-LR) on 8 gloo ranks with one image each == one process on the batch of 8."""
    single, multi = _spawn(8, None, 8)
    assert np.allclose(single["stats"], multi["stats"], rtol=5e-5, atol=1e-6), (single["stats"], multi["stats"])
    assert np.allclose(single["w"], multi["w"], rtol=2e-4, atol=4e-6)
    assert single["lr"] == multi["lr"]


def test_eight_ranks_bf16_wire_carries_the_mean():
    """bf16 wire at world 8: buckets are packed pre-divided by the world size, so the collective's bf16 sum is the gradient MEAN
    (not an 8-fold sum that is divided afterwards).  Every addend and every partial sum is then at the scale of the result: the
    update stays within bf16 rounding of the fp32-wire run, as at world 2."""
    single, multi = _spawn(8, torch.bfloat16, 8)
    assert np.allclose(single["stats"][0], multi["stats"][0], rtol=5e-5, atol=1e-6)      # first loss: no gradient used yet
    dw = np.abs(single["w"] - multi["w"]).max()
    assert 0 < dw < 2 ** -7 * 0.05 * 40, dw
    assert np.allclose(single["stats"][1], multi["stats"][1], rtol=5e-3)
