"""SGD(momentum, weight_decay) of the reference's training loop (trainer.py:42,60-63) on flat buffers.

All parameters are re-pointed into ONE flat fp32 buffer (32-B aligned slots), momentum and gradients
live in two more.  A step is two launches: a multi-tensor gather of the per-parameter .grad tensors
into the flat gradient buffer (the buffer RCCL all-reduces, in buckets, under data parallelism) and
one fused update kernel.  The learning rate lives in device memory so that a captured hipGraph can be
replayed under the poly schedule.
"""
import numpy as np
import torch

from ._lib import call, precision, ptr, register_shadow, stream

_CHUNK = 16384      # floats per gather workgroup


class FlatSGD:
    def __init__(self, params, lr, momentum=0.9, weight_decay=1e-4):
        self.params = [p for p in params if p.requires_grad]
        if not self.params:
            raise ValueError("FlatSGD got no trainable parameters")
        dev = self.params[0].device
        if dev.type != "cuda":
            raise RuntimeError("FlatSGD runs on a HIP device only")
        self.momentum, self.weight_decay = float(momentum), float(weight_decay)
        self.offsets, off = [], 0
        for p in self.params:
            self.offsets.append(off)
            off += (p.numel() + 7) // 8 * 8          # 32-B slots: the bf16 shadow of every parameter is 16-B aligned as well
        self.numel = off
        self.flat_param = torch.zeros(off, dtype=torch.float32, device=dev)
        self.flat_grad = torch.zeros(off, dtype=torch.float32, device=dev)
        self.flat_mom = torch.zeros(off, dtype=torch.float32, device=dev)
        with torch.no_grad():
            for p, o in zip(self.params, self.offsets):
                view = self.flat_param[o:o + p.numel()].view(p.shape)
                view.copy_(p.data)
                p.data = view                       # parameters now alias the flat buffer
        # bf16 working copy of the weights for the bf16 matmul mode (the Linears read it instead of rounding the fp32 master
        # weights in every GEMM); written by the update kernel, re-packed when something else writes the parameters
        self.flat_param16 = torch.empty(off, dtype=torch.bfloat16, device=dev)
        self.refresh_shadow()
        register_shadow(self, self.flat_param)
        self.lr_dev = torch.full((1,), float(lr), dtype=torch.float32, device=dev)
        self.lr = float(lr)
        # gather tables: one {src, dst, n} record per <= 16 Ki-float chunk, per gathered parameter range
        self._tables = {}
        self.param_groups = [{"lr": self.lr, "params": self.params}]   # torch.optim-like view for loops that poke lr

    # -- schedule -------------------------------------------------------------------------------------------
    def set_lr(self, lr):
        self.lr = float(lr)
        self.param_groups[0]["lr"] = self.lr
        self.lr_dev.fill_(self.lr)

    def zero_grad(self, set_to_none=True):
        for p in self.params:
            p.grad = None

    # -- step -----------------------------------------------------------------------------------------------
    def flat_range(self, first, last):
        """[lo, hi) element range of the flat buffers covered by parameters first..last-1."""
        lo = self.offsets[first]
        hi = self.offsets[last - 1] + (self.params[last - 1].numel() + 7) // 8 * 8
        return lo, hi

    def _gather_table(self, first, last):
        key = tuple(p.grad.data_ptr() if p.grad is not None else 0 for p in self.params[first:last])
        slot = self._tables.setdefault((first, last), {"key": None})
        if key != slot["key"]:
            rows = []
            base = self.flat_grad.data_ptr()
            for p, o, src in zip(self.params[first:last], self.offsets[first:last], key):
                if src == 0:
                    raise RuntimeError("FlatSGD.step(): a parameter has no gradient")
                if src == base + 4 * o:
                    continue                          # written in place (ops.engine_backward): nothing to pack
                if not p.grad.is_contiguous():
                    raise RuntimeError("FlatSGD.step(): non-contiguous gradient")
                n = p.numel()
                for c in range(0, n, _CHUNK):
                    rows.append((src + 4 * c, base + 4 * (o + c), min(_CHUNK, n - c)))
            slot["empty"] = not rows
            if not rows:
                slot["key"] = key
                return None
            if "host" not in slot:
                # pinned host side allocated once per range (outside any capture: the first call is an eager warm-up)
                slot["host"] = torch.zeros(len(rows), 3, dtype=torch.int64).pin_memory()
                slot["dev"] = torch.zeros(len(rows), 3, dtype=torch.int64, device=self.flat_grad.device)
            if not torch.cuda.is_current_stream_capturing():
                torch.cuda.current_stream().synchronize()      # a previous async upload may still read the host buffer
            slot["host"].numpy()[:] = np.asarray(rows, dtype=np.int64)
            # async upload from pinned memory: a memcpy node when captured (the host buffer lives with the optimiser)
            slot["dev"].copy_(slot["host"], non_blocking=True)
            slot["key"] = key
        return None if slot.get("empty") else slot["dev"]

    def gather_grads(self, first=0, last=None):
        """Pack p.grad of parameters first..last-1 into self.flat_grad (one launch)."""
        from .ops import join_wgrad_stream
        join_wgrad_stream()          # weight gradients may have been produced on the side stream
        last = len(self.params) if last is None else last
        t = self._gather_table(first, last)
        if t is not None:
            call("cswin_multi_copy", ptr(t), t.shape[0], stream())
        return self.flat_grad

    def apply(self, grad_scale=1.0):
        """p, m <- SGD(flat_grad * grad_scale) (one launch)."""
        call("cswin_sgd_flat", ptr(self.flat_param), ptr(self.flat_grad), ptr(self.flat_mom), self.numel, ptr(self.lr_dev),
             self.momentum, self.weight_decay, float(grad_scale), ptr(self.flat_param16) if precision() == 1 else None, stream())

    def refresh_shadow(self):
        call("cswin_pack_bf16", ptr(self.flat_param), ptr(self.flat_param16), self.numel, stream())

    def step(self, grad_scale=1.0):
        self.gather_grads()
        self.apply(grad_scale)

    def state_dict(self):
        return {"momentum": self.flat_mom.clone(), "lr": self.lr}

    def load_state_dict(self, sd):
        self.flat_mom.copy_(sd["momentum"])
        self.set_lr(sd["lr"])
