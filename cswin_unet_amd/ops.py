"""torch.autograd.Function wrappers over the C ABI (include/cswin_hip.h).

PyTorch is plumbing here: it owns device memory (caching allocator), the autograd tape and the
current stream.  All arithmetic of the hot path happens in libcswin_hip.so.  Every op raises
CswinHipError on a non-HIP tensor -- there is no CPU / eager fallback.
"""
import ctypes

import torch
from torch.autograd import Function
from torch.autograd.function import once_differentiable

from ._lib import ReduceJob, WgradDesc, act_bf16, call, dev_f32, lib, precision, ptr, shadow_ptr, stream

__all__ = ["layer_norm", "linear", "linear_pair", "mlp", "stripe_attention", "cswin_block", "conv_tokens", "patch_embed_conv", "carafe_reassemble",
           "tokens_to_nchw", "matmul_nn", "ce_dice_loss", "dropout", "img2windows", "windows2img"]


def join_wgrad_stream():
    """Kept for callers of round-2 code: weight gradients run on the calling stream (the side-stream variants measured slower
    under hipGraphs and were removed in round 3), so there is nothing to join."""


# ------------------------------------------------------------------------------------------------
# slab reductions and gradient placement of a backward pass
# ------------------------------------------------------------------------------------------------
# Every weight / LayerNorm / LePE gradient ends in a small deterministic reduction of partial slabs (cswin_reduce_job).  By
# default an op launches its own (a CSWinBlock: its six to eight in one launch) so that the gradient tensors it returns are
# complete when autograd sees them (accumulation into an existing .grad, hooks).  Inside `engine_backward(opt)` -- the HipEngine
# wraps each of its backward phases in it, and owns what happens to the gradients afterwards -- two things change:
#   * the jobs are QUEUED and reduced when the context exits, one launch per 48 jobs instead of 77 launches of ~7 us per step
#     (30 per-block batches + 47 stand-alone ones, most of that launch floor);
#   * parameter gradients are WRITTEN IN PLACE into the optimiser's flat gradient buffer (the reduction's output pointer is the
#     parameter's slot), so the pass that packed them afterwards (three launches, 2 x 94 MB per step) has nothing left to copy.
# Both rely on nobody reading a returned gradient before the context exits and on .grad being None when the pass starts.
_rq = {"on": False, "jobs": [], "keep": [], "slots": None}
MAX_REDUCE_JOBS = 48


class engine_backward:
    """with engine_backward(opt): ... one backward pass whose parameter gradients go straight to opt.flat_grad and whose slab
    reductions run once, at exit."""

    def __init__(self, opt=None):
        self.opt = opt

    def __enter__(self):
        self.prev = (_rq["on"], _rq["slots"])
        _rq["on"] = True
        if self.opt is not None:
            base = self.opt.flat_grad
            _rq["slots"] = (base, {p.data_ptr(): (o, p.numel()) for p, o in zip(self.opt.params, self.opt.offsets)})
        return self

    def __exit__(self, *exc):
        _rq["on"], _rq["slots"] = self.prev
        if exc[0] is None:
            flush_reductions()
        else:
            _rq["jobs"], _rq["keep"] = [], []
        return False


def flush_reductions():
    jobs, _rq["jobs"] = _rq["jobs"], []
    keep, _rq["keep"] = _rq["keep"], []
    st = stream()
    for i in range(0, len(jobs), MAX_REDUCE_JOBS):
        chunk = jobs[i:i + MAX_REDUCE_JOBS]
        arr = (ReduceJob * len(chunk))(*chunk)
        call("cswin_rows_sum_multi", ctypes.cast(arr, ctypes.c_void_p), len(chunk), st)
    del keep


TAIL_RIDER_JOBS = 16


def take_pending_reductions(maxn=TAIL_RIDER_JOBS):
    """Up to `maxn` queued reductions (oldest first) for a launch that can carry them at the end of its grid (cswin_linear_bwd_tail):
    (ctypes array or None, count).  The workspaces they read stay in the queue's keep-alive list until the next flush."""
    n = min(len(_rq["jobs"]), maxn) if _rq["on"] else 0
    if n == 0:
        return None, 0
    chunk = _rq["jobs"][:n]
    del _rq["jobs"][:n]
    return (ReduceJob * n)(*chunk), n


def _reduce_jobs(jobs, keep, leaf=True):
    """jobs: ReduceJob structs filled by entry points called with `deferred`; keep: the workspaces they read.  leaf: every output
    is the gradient of a leaf parameter (nothing else in this backward pass reads it), so the jobs may wait for the flush."""
    jobs = [j for j in jobs if j.part]                    # an entry point that had nothing to reduce leaves its slot zeroed
    if not jobs:
        return
    if not (leaf and _rq["on"]):                          # needed now (a composed weight's gradient feeds the next backward node)
        arr = (ReduceJob * len(jobs))(*jobs)
        call("cswin_rows_sum_multi", ctypes.cast(arr, ctypes.c_void_p), len(jobs), stream())
        return
    for j in jobs:
        c = ReduceJob()
        ctypes.memmove(ctypes.byref(c), ctypes.byref(j), ctypes.sizeof(ReduceJob))
        _rq["jobs"].append(c)
    _rq["keep"] += [t for t in keep if t is not None]


def _all_leaf(*ts):
    """True when every given tensor is a leaf of the autograd graph (a module parameter passed as it is, not a view or a
    function of one): only then is its gradient final when the op returns it -- and nobody's input."""
    return all(t is None or t.is_leaf for t in ts)


def _grad_like(w, leaf=True):
    """The tensor a parameter gradient is written to: the parameter's slot of the flat gradient buffer inside engine_backward (leaf
    parameters only), a fresh tensor otherwise."""
    return _grad_at(w.data_ptr() if leaf else 0, w.shape, w.device)


def _grad_at(param_ptr, shape, device):
    """Same for a parameter known by its address only (biases are not saved for backward; their data_ptr is; 0 = no slot)."""
    slots = _rq["slots"]
    if slots is not None and param_ptr:
        n = 1
        for d in shape:
            n *= int(d)
        hit = slots[1].get(param_ptr)
        if hit is not None and hit[1] == n:
            return slots[0][hit[0]:hit[0] + n].view(shape)
    return torch.empty(tuple(shape), dtype=torch.float32, device=device)


def _pptr(t):
    return t.data_ptr() if t is not None else 0


def _ws(nbytes, device):
    return torch.empty(max(int(nbytes), 16) // 4 + 4, dtype=torch.float32, device=device)


def _int_array(vals):
    return (ctypes.c_int * len(vals))(*vals)


def _ptr_array(tensors):
    return (ctypes.c_void_p * len(tensors))(*[t.data_ptr() for t in tensors])


# ------------------------------------------------------------------------------------------------
# LayerNorm
# ------------------------------------------------------------------------------------------------
class _LayerNorm(Function):
    @staticmethod
    def forward(ctx, x, gamma, beta, eps):
        x, gamma, beta = dev_f32(x, "layer_norm input"), dev_f32(gamma), dev_f32(beta)
        C = x.shape[-1]
        M = x.numel() // C
        y = torch.empty_like(x)
        mean = torch.empty(M, dtype=torch.float32, device=x.device)
        rstd = torch.empty_like(mean)
        call("cswin_layernorm_fwd", ptr(x), ptr(gamma), ptr(beta), ptr(y), ptr(mean), ptr(rstd), M, C, eps, 0, stream())
        ctx.save_for_backward(x, gamma, mean, rstd)
        ctx.leaf = _all_leaf(gamma, beta)
        ctx.beta_ptr = _pptr(beta) if ctx.leaf else 0
        return y

    @staticmethod
    @once_differentiable
    def backward(ctx, dy):
        x, gamma, mean, rstd = ctx.saved_tensors
        dy = dev_f32(dy)
        C = x.shape[-1]
        M = x.numel() // C
        dx = torch.empty_like(x)
        dg = _grad_like(gamma, ctx.leaf)
        db = _grad_at(ctx.beta_ptr, gamma.shape, gamma.device)
        nbytes = lib().cswin_layernorm_bwd_workspace(M, C)
        ws = _ws(nbytes, x.device)
        job = (ReduceJob * 1)()
        call("cswin_layernorm_bwd", ptr(dy), ptr(x), ptr(mean), ptr(rstd), ptr(gamma), None, ptr(dx), ptr(dg), ptr(db),
             ptr(ws), nbytes, M, C, ctypes.cast(job, ctypes.c_void_p), None, stream())
        _reduce_jobs(job, (ws,), ctx.leaf)
        return dx, dg, db, None


def layer_norm(x, gamma, beta, eps=1e-5):
    return _LayerNorm.apply(x, gamma, beta, eps)


# ------------------------------------------------------------------------------------------------
# Linear (+ fused skip-concat input, residual / DropPath epilogue)
# ------------------------------------------------------------------------------------------------
def _rows_per_sample(x):
    return x.numel() // (x.shape[0] * x.shape[-1])


# bf16 TWINS of fp32 residual-stream gradients (bf16 mode): a block's backward hands the gradient of its input to the next
# backward as fp32 (the residual path needs it) and leaves a rounded copy here for that block's GEMMs.  An entry is consumed
# once and only by the very tensor it was made for (same TensorImpl: the entry keeps the tensor alive, so its address cannot be
# reused); autograd sums fan-outs into new tensors, which then simply have no twin.
_twins = {}


def _twin_put(t, t16):
    _twins[t.data_ptr()] = (t, t16)


def _twin_take(t):
    e = _twins.pop(t.data_ptr(), None)
    return e[1] if e is not None and e[0]._cdata == t._cdata else None


def clear_twins():
    _twins.clear()


def _wsrc(w):
    """(pointer, io_bf16 bits) of a Linear weight for cswin_linear_fwd / _bwd_data: in the bf16 mode with bf16 storage, the bf16
    shadow the optimiser keeps of it (bit 2) when there is one -- same values as the GEMM's own rounding, half the bytes --,
    else the fp32 tensor."""
    if act_bf16() and w.shape[0] % 4 == 0 and w.shape[1] % 4 == 0:
        sp = shadow_ptr(w)
        if sp is not None:
            return sp, 4
    return ptr(w), 0


class _Linear(Function):
    @staticmethod
    def forward(ctx, x, w, b, x2, residual, row_scale):
        x, w = dev_f32(x, "linear input"), dev_f32(w, "linear weight")
        b, x2, residual, row_scale = dev_f32(b), dev_f32(x2), dev_f32(residual), dev_f32(row_scale)
        K1 = x.shape[-1]
        K = K1 + (x2.shape[-1] if x2 is not None else 0)
        N = w.shape[0]
        assert w.shape[1] == K, f"linear: weight {tuple(w.shape)} vs input features {K}"
        M = x.numel() // K1
        y = torch.empty(x.shape[:-1] + (N,), dtype=torch.float32, device=x.device)
        rps = _rows_per_sample(x) if row_scale is not None else 1
        pw, fw = _wsrc(w)
        call("cswin_linear_fwd", ptr(x), ptr(x2), K1 if x2 is not None else 0, pw, ptr(b), ptr(y), None, ptr(residual),
             ptr(row_scale), rps, M, N, K, precision(), fw, stream())
        ctx.save_for_backward(x, w, x2, row_scale)
        ctx.leaf = _all_leaf(w, b)
        ctx.has_bias, ctx.has_res, ctx.rps, ctx.bias_ptr = b is not None, residual is not None, rps, _pptr(b) if ctx.leaf else 0
        return y

    @staticmethod
    @once_differentiable
    def backward(ctx, dy):
        x, w, x2, row_scale = ctx.saved_tensors
        dy = dev_f32(dy)
        K1 = x.shape[-1]
        N, K = w.shape
        M = x.numel() // K1
        need = ctx.needs_input_grad
        dx = dx2 = dw = db = None
        if need[0] or (x2 is not None and need[3]):
            dx = torch.empty_like(x)
            dx2 = torch.empty_like(x2) if x2 is not None else None
            pw, fw = _wsrc(w)
            call("cswin_linear_bwd_data", ptr(dy), pw, ptr(dx), ptr(dx2), K1 if x2 is not None else 0, None,
                 ptr(row_scale), ctx.rps, None, M, N, K, precision(), fw, stream())
        if need[1]:
            if True:
                dw = _grad_like(w, ctx.leaf)
                db = _grad_at(ctx.bias_ptr, (N,), w.device) if ctx.has_bias else None
                nbytes = lib().cswin_linear_bwd_weight_workspace(M, N, K)
                ws = _ws(nbytes, w.device)
                job = (ReduceJob * 1)()
                call("cswin_linear_bwd_weight", ptr(dy), ptr(x), ptr(x2), K1 if x2 is not None else 0, ptr(row_scale), ctx.rps,
                     ptr(dw), ptr(db), ptr(ws), nbytes, M, N, K, ctypes.cast(job, ctypes.c_void_p), precision(), stream())
                _reduce_jobs(job, (ws,), ctx.leaf)
        dres = dy if ctx.has_res else None
        return dx, dw, db, dx2, dres, None


def linear(x, w, b=None, x2=None, residual=None, row_scale=None):
    """y = [x | x2] @ w^T + b;  with residual: y = residual + row_scale[sample] * (...)  (DropPath + skip add)."""
    return _Linear.apply(x, w, b, x2, residual, row_scale)


class _LinearPair(Function):
    """Two Linears of the same input (CARAFE: `down` and `out` 1x1 convs of x, cswin_unet.py:240,265).  As two separate
    autograd nodes their input gradients meet in an aten::add over (B, L, C); here the second data-gradient GEMM adds into
    the first one's result in its epilogue."""

    @staticmethod
    def forward(ctx, x, w1, b1, w2, b2):
        x, w1, b1, w2, b2 = (dev_f32(t) for t in (x, w1, b1, w2, b2))
        K = x.shape[-1]
        M = x.numel() // K
        ys = []
        for w, b in ((w1, b1), (w2, b2)):
            assert w.shape[1] == K
            y = torch.empty(x.shape[:-1] + (w.shape[0],), dtype=torch.float32, device=x.device)
            call("cswin_linear_fwd", ptr(x), None, 0, ptr(w), ptr(b), ptr(y), None, None, None, 1, M, w.shape[0], K, precision(), 0, stream())
            ys.append(y)
        ctx.save_for_backward(x, w1, w2)
        ctx.has_b = (b1 is not None, b2 is not None)
        ctx.leaf = _all_leaf(w1, b1, w2, b2)
        ctx.bias_ptrs = (_pptr(b1), _pptr(b2)) if ctx.leaf else (0, 0)
        return tuple(ys)

    @staticmethod
    @once_differentiable
    def backward(ctx, dy1, dy2):
        x, w1, w2 = ctx.saved_tensors
        dy1, dy2 = dev_f32(dy1), dev_f32(dy2)
        K = x.shape[-1]
        M = x.numel() // K
        st = stream()
        dx = None
        if ctx.needs_input_grad[0]:
            dx = torch.empty_like(x)
            call("cswin_linear_bwd_data", ptr(dy1), ptr(w1), ptr(dx), None, 0, None, None, 1, None, M, w1.shape[0], K, precision(), 0, st)
            call("cswin_linear_bwd_data", ptr(dy2), ptr(w2), ptr(dx), None, 0, None, None, 1, ptr(dx), M, w2.shape[0], K, precision(), 0, st)
        grads, keep = [], []
        wg, jobs = (WgradDesc * 2)(), (ReduceJob * 2)()
        for i, (dy, w, has_b) in enumerate(((dy1, w1, ctx.has_b[0]), (dy2, w2, ctx.has_b[1]))):
            N = w.shape[0]
            dw = _grad_like(w, ctx.leaf)
            db = _grad_at(ctx.bias_ptrs[i], (N,), w.device) if has_b else None
            nbytes = lib().cswin_linear_bwd_weight_workspace(M, N, K)
            ws = _ws(nbytes, w.device)
            keep.append(ws)
            wg[i].dy, wg[i].x, wg[i].row_scale, wg[i].dw = dy.data_ptr(), x.data_ptr(), None, dw.data_ptr()
            wg[i].dbias, wg[i].workspace, wg[i].ws_bytes = (db.data_ptr() if has_b else None), ws.data_ptr(), nbytes
            wg[i].rows_per_sample, wg[i].M, wg[i].N, wg[i].K, wg[i].precision = 1, M, N, K, precision()
            grads += [dw, db]
        call("cswin_linear_bwd_weight_batch", ctypes.cast(wg, ctypes.c_void_p), 2, ctypes.cast(jobs, ctypes.c_void_p), None, 0, st)
        _reduce_jobs(jobs, keep, ctx.leaf)
        return (dx,) + tuple(grads)


def linear_pair(x, w1, b1, w2, b2):
    """(x @ w1^T + b1, x @ w2^T + b2) with one fused input gradient."""
    return _LinearPair.apply(x, w1, b1, w2, b2)


class _Mlp(Function):
    """fc1 -> GELU(erf) -> drop -> fc2 -> drop (cswin_unet.py:22-28) with the optional residual/DropPath epilogue of :179.
    drop_p = 0 (every reference config): two GEMMs with fused epilogues.  drop_p > 0: the two nn.Dropouts are separate
    launches of cswin_dropout (masks regenerated from their seeds in backward; they commute with the GELU' factor)."""

    @staticmethod
    def forward(ctx, x, w1, b1, w2, b2, residual, row_scale, drop_p, seeds):
        x, w1, b1, w2, b2 = (dev_f32(t) for t in (x, w1, b1, w2, b2))
        residual, row_scale = dev_f32(residual), dev_f32(row_scale)
        K = x.shape[-1]
        Hd, N = w1.shape[0], w2.shape[0]
        M = x.numel() // K
        pre = torch.empty(x.shape[:-1] + (Hd,), dtype=torch.float32, device=x.device)
        act = torch.empty_like(pre)
        (p1, f1), (p2, f2) = _wsrc(w1), _wsrc(w2)
        call("cswin_linear_fwd", ptr(x), None, 0, p1, ptr(b1), ptr(pre), ptr(act), None, None, 1, M, Hd, K, precision(), f1, stream())
        y = torch.empty(x.shape[:-1] + (N,), dtype=torch.float32, device=x.device)
        rps = _rows_per_sample(x) if row_scale is not None else 1
        if drop_p > 0:
            call("cswin_dropout", ptr(act), None, None, ptr(act), act.numel(), act.numel() // act.shape[0], drop_p, seeds[0], ptr(dropout_epoch(act.device)), stream())
            call("cswin_linear_fwd", ptr(act), None, 0, p2, ptr(b2), ptr(y), None, None, None, 1, M, N, Hd, precision(), f2, stream())
            call("cswin_dropout", ptr(y), ptr(residual), ptr(row_scale), ptr(y), y.numel(), y.numel() // y.shape[0], drop_p, seeds[1], ptr(dropout_epoch(y.device)), stream())
        else:
            call("cswin_linear_fwd", ptr(act), None, 0, p2, ptr(b2), ptr(y), None, ptr(residual), ptr(row_scale), rps,
                 M, N, Hd, precision(), f2, stream())
        ctx.save_for_backward(x, w1, w2, pre, act, row_scale)
        ctx.has_res, ctx.rps, ctx.has_b1, ctx.has_b2 = residual is not None, rps, b1 is not None, b2 is not None
        ctx.leaf = _all_leaf(w1, b1, w2, b2)
        ctx.bias_ptrs = (_pptr(b1), _pptr(b2)) if ctx.leaf else (0, 0)
        ctx.drop = (float(drop_p), seeds)
        return y

    @staticmethod
    @once_differentiable
    def backward(ctx, dy):
        x, w1, w2, pre, act, row_scale = ctx.saved_tensors
        dy = dev_f32(dy)
        K = x.shape[-1]
        Hd, N = w1.shape[0], w2.shape[0]
        M = x.numel() // K
        dev = x.device
        st = stream()
        drop_p, seeds = ctx.drop
        dyl, rs_gemm = dy, row_scale                 # gradient of fc2's output, and the row factor still to be applied by the GEMMs
        if drop_p > 0:
            dyl = torch.empty_like(dy)
            call("cswin_dropout", ptr(dy), None, ptr(row_scale), ptr(dyl), dy.numel(), dy.numel() // dy.shape[0], drop_p, seeds[1], ptr(dropout_epoch(dy.device)), st)
            rs_gemm = None
        # d pre = (row_scale * dy @ w2) * gelu'(pre)   (GELU backward fused into the data-gradient epilogue)
        dpre = torch.empty_like(pre)
        call("cswin_linear_bwd_data", ptr(dyl), ptr(w2), ptr(dpre), None, 0, ptr(pre), ptr(rs_gemm), ctx.rps, None, M, N,
             Hd, precision(), 0, st)
        if drop_p > 0:
            call("cswin_dropout", ptr(dpre), None, None, ptr(dpre), dpre.numel(), dpre.numel() // dpre.shape[0], drop_p, seeds[0], ptr(dropout_epoch(dpre.device)), st)
        dx = None
        if ctx.needs_input_grad[0]:
            dx = torch.empty_like(x)
            call("cswin_linear_bwd_data", ptr(dpre), ptr(w1), ptr(dx), None, 0, None, None, 1, None, M, Hd, K, precision(), 0, st)
        if True:                                                # both weight gradients
            dw2 = _grad_like(w2, ctx.leaf)
            db2 = _grad_at(ctx.bias_ptrs[1], (N,), dev) if ctx.has_b2 else None
            nbytes = max(lib().cswin_linear_bwd_weight_workspace(M, N, Hd), lib().cswin_linear_bwd_weight_workspace(M, Hd, K))
            ws, ws1 = _ws(nbytes, dev), _ws(nbytes, dev)
            jobs = (ReduceJob * 2)()
            call("cswin_linear_bwd_weight", ptr(dyl), ptr(act), None, 0, ptr(rs_gemm), ctx.rps, ptr(dw2), ptr(db2), ptr(ws),
                 nbytes, M, N, Hd, ctypes.cast(ctypes.byref(jobs[0]), ctypes.c_void_p), precision(), stream())
            dw1 = _grad_like(w1, ctx.leaf)
            db1 = _grad_at(ctx.bias_ptrs[0], (Hd,), dev) if ctx.has_b1 else None
            call("cswin_linear_bwd_weight", ptr(dpre), ptr(x), None, 0, None, 1, ptr(dw1), ptr(db1), ptr(ws1), nbytes, M, Hd, K,
                 ctypes.cast(ctypes.byref(jobs[1]), ctypes.c_void_p), precision(), stream())
            _reduce_jobs(jobs, (ws, ws1), ctx.leaf)
        return dx, dw1, db1, dw2, db2, (dy if ctx.has_res else None), None, None, None


# Device-resident dropout epoch: every dropout launch adds its value to the seed it was given.  The seed itself comes from the host
# generator when the op is called -- under hipGraph replay that is ONCE, at capture -- so the engine advances this counter by one
# kernel inside the captured step and every replay draws fresh masks (forward and backward of one step see the same value).
_epoch = {}


def dropout_epoch(device):
    t = _epoch.get(device)
    if t is None:
        t = _epoch[device] = torch.zeros(1, dtype=torch.int64, device=device)
    return t


def advance_dropout_epoch(device):
    """epoch += 1 on the current stream (capturable)."""
    dropout_epoch(device).add_(1)


def _draw_seeds(n):
    return tuple(int(v) for v in torch.randint(0, 2 ** 62, (n,)).tolist())      # host RNG: torch.manual_seed controls it


def mlp(x, w1, b1, w2, b2, residual=None, row_scale=None, drop_p=0.0):
    return _Mlp.apply(x, w1, b1, w2, b2, residual, row_scale, float(drop_p), _draw_seeds(2) if drop_p > 0 else (0, 0))


class _MatmulNN(Function):
    """c (M, K) = a (M, N) @ b (N, K) -- used to compose the 1x1 `output` head with upsample1.out."""

    @staticmethod
    def forward(ctx, a, b):
        a, b = dev_f32(a), dev_f32(b)
        M, N = a.shape
        K = b.shape[1]
        c = torch.empty(M, K, dtype=torch.float32, device=a.device)
        call("cswin_linear_bwd_data", ptr(a), ptr(b), ptr(c), None, 0, None, None, 1, None, M, N, K, precision(), 0, stream())
        ctx.save_for_backward(a, b)
        return c

    @staticmethod
    @once_differentiable
    def backward(ctx, dc):
        a, b = ctx.saved_tensors
        dc = dev_f32(dc)
        M, N = a.shape
        K = b.shape[1]
        da = torch.empty_like(a)        # da = dc @ b^T
        call("cswin_linear_fwd", ptr(dc), None, 0, ptr(b), None, ptr(da), None, None, None, 1, M, N, K, precision(), 0, stream())
        db = torch.empty_like(b)        # db (N, K) = a^T @ dc
        nbytes = lib().cswin_linear_bwd_weight_workspace(M, N, K)
        ws = _ws(nbytes, a.device)
        job = (ReduceJob * 1)()
        call("cswin_linear_bwd_weight", ptr(a), ptr(dc), None, 0, None, 1, ptr(db), None, ptr(ws), nbytes, M, N, K,
             ctypes.cast(job, ctypes.c_void_p), precision(), stream())
        _reduce_jobs(job, (ws,), False)
        return da, db


def matmul_nn(a, b):
    return _MatmulNN.apply(a, b)


# ------------------------------------------------------------------------------------------------
# fused stripe attention
# ------------------------------------------------------------------------------------------------
class _StripeAttention(Function):
    @staticmethod
    def forward(ctx, qkv, reso, split, idx, heads, scale, drop, *wb):
        # qkv may arrive STORED as bf16 (the bf16 mode's activation storage): the kernel widens on load, dqkv comes back as bf16
        q16 = qkv.dtype == torch.bfloat16 and qkv.is_cuda
        qkv = qkv.contiguous() if q16 else dev_f32(qkv, "attention qkv")
        nb = len(idx)
        ws_ = [dev_f32(t).view(t.shape[0], 9) for t in wb[:nb]]
        bs_ = [dev_f32(t) for t in wb[nb:]]
        B, L, C3 = qkv.shape
        C = C3 // 3
        if L != reso * reso:
            raise ValueError("flatten img_tokens has wrong size")
        y = torch.empty(B, L, C, dtype=torch.float32, device=qkv.device)
        y0 = torch.empty_like(y) if any(ctx.needs_input_grad) else None          # P V without LePE: the backward's delta term
        lse = torch.empty(B, sum(heads), L, dtype=torch.float32, device=qkv.device)
        call("cswin_attn_fwd", ptr(qkv), _ptr_array(ws_), _ptr_array(bs_), ptr(y), ptr(y0), ptr(lse), B, reso, C, nb,
             _int_array(heads), _int_array(idx), split, float(scale or 0.0), drop[0], drop[1], ptr(dropout_epoch(qkv.device)) if drop[0] > 0 else None,
             int(q16), stream())
        ctx.save_for_backward(qkv, lse, y0, *ws_, *bs_)
        ctx.leaf = _all_leaf(*wb)
        ctx.meta = (reso, split, tuple(idx), tuple(heads), float(scale or 0.0), drop)
        return y

    @staticmethod
    @once_differentiable
    def backward(ctx, dy):
        qkv, lse, y0, *wb_ = ctx.saved_tensors
        reso, split, idx, heads, scale, drop = ctx.meta
        dy = dev_f32(dy)
        nb = len(idx)
        ws_, bs_ = wb_[:nb], wb_[nb:]
        B, L, C3 = qkv.shape
        C = C3 // 3
        dqkv = torch.empty_like(qkv)
        dws = [_grad_like(w, ctx.leaf) for w in ws_]
        dbs = [_grad_like(b, ctx.leaf) for b in bs_]
        ha, ia = _int_array(heads), _int_array(idx)
        nbytes = lib().cswin_attn_bwd_workspace(B, reso, C, nb, ha, ia, split)
        ws = _ws(nbytes, qkv.device)
        jobs = (ReduceJob * 2)()
        call("cswin_attn_bwd", ptr(qkv), _ptr_array(ws_), _ptr_array(bs_), ptr(lse), ptr(y0), ptr(dy), ptr(dqkv), _ptr_array(dws),
             _ptr_array(dbs), ptr(ws), nbytes, B, reso, C, nb, ha, ia, split, scale, ctypes.cast(jobs, ctypes.c_void_p), drop[0], drop[1],
             ptr(dropout_epoch(qkv.device)) if drop[0] > 0 else None, int(qkv.dtype == torch.bfloat16), stream())
        _reduce_jobs(jobs, [ws], ctx.leaf)
        return (dqkv, None, None, None, None, None, None) + tuple(d.view(d.shape[0], 1, 3, 3) for d in dws) + tuple(dbs)


def _attn_drop(p):
    """(p, seed) of one attention-probability dropout draw (nn.Dropout of cswin_unet.py:57,101); the seed comes from the host
    generator (torch.manual_seed controls it), the backward kernels regenerate the mask from it."""
    p = float(p)
    return (p, _draw_seeds(1)[0]) if p > 0 else (0.0, 0)


def stripe_attention(qkv, reso, split, idx, heads, lepe_w, lepe_b, scale=None, attn_drop=0.0):
    """qkv (B, L, 3C) -> (B, L, C).  idx/heads/lepe_w/lepe_b: one entry per branch.  attn_drop: dropout probability on the
    attention probabilities (training only; the caller passes 0 in eval mode)."""
    return _StripeAttention.apply(qkv, reso, split, tuple(idx), tuple(heads), scale, _attn_drop(attn_drop), *lepe_w, *lepe_b)


# ------------------------------------------------------------------------------------------------
# whole CSWinBlock as ONE autograd node
# ------------------------------------------------------------------------------------------------
class _CSWinBlock(Function):
    """x -> x + dp1(proj(attn(qkv(LN1 x)))) -> ... + dp2(fc2(gelu(fc1(LN2 .))))  (cswin_unet.py:160-181).

    Same kernels as the fine-grained ops; as one node the backward chains them by hand, so the two residual-fork
    gradient sums are the `dres` input of the LayerNorm backward kernel (no aten::add), and 14 autograd nodes per
    block become one."""

    @staticmethod
    def forward(ctx, x, reso, split, idx, heads, scale, eps1, eps2, rs1, rs2, drop, g1, b1, wqkv, bqkv, wp, bp, g2, b2, w1, bb1,
                w2, bb2, *lepe):
        x = dev_f32(x, "block input")
        B, L, C = x.shape
        if L != reso * reso:
            raise ValueError("flatten img_tokens has wrong size")
        M = B * L
        dev, st = x.device, stream()
        nb = len(idx)
        lw = [dev_f32(t).view(t.shape[0], 9) for t in lepe[:nb]]
        lb = [dev_f32(t) for t in lepe[nb:]]
        rs1, rs2 = dev_f32(rs1), dev_f32(rs2)
        E = lambda *shape: torch.empty(*shape, dtype=torch.float32, device=dev)
        # bf16 mode: every tensor of the block that only GEMMs and the attention kernel read -- both LayerNorm outputs, qkv, the
        # attention output, the MLP hidden pair and (backward) the gradients of qkv and the hidden layer -- is STORED as bf16, and
        # the GEMMs read the weights' bf16 shadow (_wsrc).  The residual stream (x, x1, y), its gradients, the LayerNorm / softmax
        # statistics, master weights and every accumulation stay fp32.
        s16 = act_bf16() and C % 4 == 0
        E16 = (lambda *shape: torch.empty(*shape, dtype=torch.bfloat16, device=dev)) if s16 else E
        (pq, fq), (pp, fp), (p1, f1), (p2, f2) = [_wsrc(w) if s16 else (ptr(w), 0) for w in (wqkv, wp, w1, w2)]
        io_x, io_xy = (1, 3) if s16 else (0, 0)                       # input / input and output stored as bf16
        h1, m1, r1 = E16(B, L, C), E(M), E(M)
        call("cswin_layernorm_fwd", ptr(x), ptr(g1), ptr(b1), ptr(h1), ptr(m1), ptr(r1), M, C, eps1, int(s16), st)
        qkv = E16(B, L, 3 * C)
        call("cswin_linear_fwd", ptr(h1), None, 0, pq, ptr(bqkv), ptr(qkv), None, None, None, 1, M, 3 * C, C, precision(), io_xy | fq, st)
        att, lse = E16(B, L, C), E(B, sum(heads), L)
        att0 = E16(B, L, C) if any(ctx.needs_input_grad) else None       # P V without LePE: the attention backward's delta term
        ha, ia = _int_array(heads), _int_array(idx)
        call("cswin_attn_fwd", ptr(qkv), _ptr_array(lw), _ptr_array(lb), ptr(att), ptr(att0), ptr(lse), B, reso, C, nb, ha, ia, split,
             float(scale or 0.0), drop[0], drop[1], ptr(dropout_epoch(dev)) if drop[0] > 0 else None, 7 if s16 else 0, st)
        x1 = torch.empty_like(x)
        call("cswin_linear_fwd", ptr(att), None, 0, pp, ptr(bp), ptr(x1), None, ptr(x), ptr(rs1), L, M, C, C, precision(), io_x | fp, st)
        h2, m2, r2 = E16(B, L, C), E(M), E(M)
        call("cswin_layernorm_fwd", ptr(x1), ptr(g2), ptr(b2), ptr(h2), ptr(m2), ptr(r2), M, C, eps2, int(s16), st)
        Hd = w1.shape[0]
        pre, act = E16(B, L, Hd), E16(B, L, Hd)
        call("cswin_linear_fwd", ptr(h2), None, 0, p1, ptr(bb1), ptr(pre), ptr(act), None, None, 1, M, Hd, C, precision(), io_xy | f1, st)
        y = torch.empty_like(x)
        call("cswin_linear_fwd", ptr(act), None, 0, p2, ptr(bb2), ptr(y), None, ptr(x1), ptr(rs2), L, M, C, Hd, precision(), io_x | f2, st)
        ctx.save_for_backward(x, m1, r1, h1, qkv, lse, att, att0, x1, m2, r2, h2, pre, act, rs1, rs2, g1, wqkv, wp, g2, w1, w2, *lw, *lb)
        ctx.meta = (reso, split, tuple(idx), tuple(heads), float(scale or 0.0), bqkv is not None, s16, drop)
        ctx.leaf = _all_leaf(g1, b1, wqkv, bqkv, wp, bp, g2, b2, w1, bb1, w2, bb2, *lepe)
        ctx.bptrs = tuple(_pptr(t) if ctx.leaf else 0 for t in (b1, bqkv, bp, b2, bb1, bb2))     # parameters that are not saved: where their gradients go
        return y

    @staticmethod
    @once_differentiable
    def backward(ctx, dy):
        (x, m1, r1, h1, qkv, lse, att, att0, x1, m2, r2, h2, pre, act, rs1, rs2, g1, wqkv, wp, g2, w1, w2, *lwb) = ctx.saved_tensors
        reso, split, idx, heads, scale, has_qkv_bias, s16, drop = ctx.meta
        lw, lb = lwb[:len(idx)], lwb[len(idx):]
        dy = dev_f32(dy)
        B, L, C = x.shape
        M, Hd, nb = B * L, w1.shape[0], len(idx)
        dev, st, h = x.device, stream(), lib()
        E = lambda *shape: torch.empty(*shape, dtype=torch.float32, device=dev)
        # the slab reductions (4 split-K weight gradients, 2 LayerNorm dgamma/dbeta, 1-2 LePE conv gradients) are deferred
        # and run as ONE launch
        sizes = [h.cswin_linear_bwd_weight_workspace(M, C, Hd), h.cswin_linear_bwd_weight_workspace(M, Hd, C),
                 h.cswin_layernorm_bwd_workspace(M, C), h.cswin_linear_bwd_weight_workspace(M, C, C),
                 h.cswin_linear_bwd_weight_workspace(M, 3 * C, C), h.cswin_layernorm_bwd_workspace(M, C)]
        sizes = [(n + 255) // 256 * 256 for n in sizes]
        ws = _ws(sum(sizes), dev)
        wsp = [ctypes.c_void_p(ws.data_ptr() + sum(sizes[:i])) for i in range(6)]
        jobs = (ReduceJob * 8)()
        J = lambda i: ctypes.cast(ctypes.byref(jobs[i]), ctypes.c_void_p)
        (pq, fq), (pp, fp), (p1, f1), (p2, f2) = [_wsrc(w) if s16 else (ptr(w), 0) for w in (wqkv, wp, w1, w2)]
        E16 = lambda *shape: torch.empty(*shape, dtype=torch.bfloat16, device=dev)
        dy16 = _twin_take(dy) if s16 else None          # rounded copy of dy left by the backward that produced it (see _twins)
        # ---- MLP branch ----
        dpre = torch.empty_like(pre)
        call("cswin_linear_bwd_data", ptr(dy16 if dy16 is not None else dy), p2, ptr(dpre), None, 0, ptr(pre), ptr(rs2), L, None, M, C, Hd,
             precision(), ((10 if s16 else 0) | f2) | (1 if dy16 is not None else 0), st)
        # the four weight gradients are off the critical path: they run as ONE batched launch once all operands exist
        wg = (WgradDesc * 4)()

        def defer_wgrad(slot, dy_, x_, rs_, dw_, db_, wsi, N_, K_, io=0):
            wg[slot].io_bf16 = io if s16 else 0
            wg[slot].dy, wg[slot].x, wg[slot].row_scale = dy_.data_ptr(), x_.data_ptr(), (rs_.data_ptr() if rs_ is not None else None)
            wg[slot].dw, wg[slot].dbias = dw_.data_ptr(), (db_.data_ptr() if db_ is not None else None)
            wg[slot].workspace, wg[slot].ws_bytes = wsp[wsi].value, sizes[wsi]
            wg[slot].rows_per_sample, wg[slot].M, wg[slot].N, wg[slot].K, wg[slot].precision = L, M, N_, K_, precision()

        pb1, pbqkv, pbp, pb2, pbb1, pbb2 = ctx.bptrs
        leaf = ctx.leaf
        dw2, db2 = _grad_like(w2, leaf), _grad_at(pbb2, (C,), dev)
        if dy16 is not None:
            defer_wgrad(0, dy16, act, rs2, dw2, db2, 0, C, Hd, io=3)   # dy's twin and x = act are stored as bf16
        else:
            defer_wgrad(0, dy, act, rs2, dw2, db2, 0, C, Hd, io=2)     # x = act is stored as bf16
        dw1, db1 = _grad_like(w1, leaf), _grad_at(pbb1, (Hd,), dev)
        defer_wgrad(1, dpre, h2, None, dw1, db1, 1, Hd, C, io=3)        # dy = dpre and x = h2 are stored as bf16
        dh2 = torch.empty_like(x)                                      # fp32 (x is)
        call("cswin_linear_bwd_data", ptr(dpre), p1, ptr(dh2), None, 0, None, None, 1, None, M, Hd, C, precision(), (1 if s16 else 0) | f1, st)
        dx1, dg2, dbt2 = torch.empty_like(x), _grad_like(g2, leaf), _grad_at(pb2, (C,), dev)
        dx1_16 = E16(B, L, C) if s16 else None                         # the GEMMs below read the twin, the residual path dx1
        call("cswin_layernorm_bwd", ptr(dh2), ptr(x1), ptr(m2), ptr(r2), ptr(g2), ptr(dy), ptr(dx1), ptr(dg2), ptr(dbt2), wsp[2],
             sizes[2], M, C, J(2), ptr(dx1_16), st)
        # ---- attention branch ----
        datt = dh2                                                     # reuse
        call("cswin_linear_bwd_data", ptr(dx1_16 if s16 else dx1), pp, ptr(datt), None, 0, None, ptr(rs1), L, None, M, C, C, precision(),
             fp | (1 if s16 else 0), st)
        dwp, dbp = _grad_like(wp, leaf), _grad_at(pbp, (C,), dev)
        defer_wgrad(2, dx1_16 if s16 else dx1, att, rs1, dwp, dbp, 3, C, C, io=3)      # dx1's twin and x = att are stored as bf16
        dqkv = torch.empty_like(qkv)
        dlw = [_grad_like(t, leaf) for t in lw]
        dlb = [_grad_like(t, leaf) for t in lb]
        ha, ia = _int_array(heads), _int_array(idx)
        naw = h.cswin_attn_bwd_workspace(B, reso, C, nb, ha, ia, split)
        aws = _ws(naw, dev)
        call("cswin_attn_bwd", ptr(qkv), _ptr_array(lw), _ptr_array(lb), ptr(lse), ptr(att0), ptr(datt), ptr(dqkv),
             _ptr_array(dlw), _ptr_array(dlb), ptr(aws), naw, B, reso, C, nb, ha, ia, split, scale, J(6), drop[0], drop[1],
             ptr(dropout_epoch(dev)) if drop[0] > 0 else None, 7 if s16 else 0, st)
        dwqkv = _grad_like(wqkv, leaf)
        dbqkv = _grad_at(pbqkv, (3 * C,), dev) if has_qkv_bias else None
        defer_wgrad(3, dqkv, h1, None, dwqkv, dbqkv, 4, 3 * C, C, io=3)  # dy = dqkv and x = h1 are stored as bf16
        wjobs = (ReduceJob * 4)()
        dh1 = datt                                                     # reuse again
        pend, npend = take_pending_reductions()              # the previous blocks' slab reductions ride at the end of this grid
        pend = ctypes.cast(pend, ctypes.c_void_p) if npend else None
        if precision() == 0:
            # fp32: the qkv data gradient rides in the weight-gradient batch's launch as well (both only wait for dqkv)
            call("cswin_linear_bwd_tail", ptr(dqkv), pq, ptr(dh1), M, 3 * C, C, ctypes.cast(wg, ctypes.c_void_p), 4,
                 ctypes.cast(wjobs, ctypes.c_void_p), pend, npend, st)
        else:
            call("cswin_linear_bwd_weight_batch", ctypes.cast(wg, ctypes.c_void_p), 4, ctypes.cast(wjobs, ctypes.c_void_p), pend, npend, st)
            call("cswin_linear_bwd_data", ptr(dqkv), pq, ptr(dh1), None, 0, None, None, 1, None, M, 3 * C, C, precision(), (1 if s16 else 0) | fq, st)
        for slot, ji in enumerate((0, 1, 3, 4)):
            jobs[ji] = wjobs[slot]
        dx, dg1, dbt1 = torch.empty_like(x), _grad_like(g1, leaf), _grad_at(pb1, (C,), dev)
        dx16 = E16(B, L, C) if s16 else None
        call("cswin_layernorm_bwd", ptr(dh1), ptr(x), ptr(m1), ptr(r1), ptr(g1), ptr(dx1), ptr(dx), ptr(dg1), ptr(dbt1), wsp[5],
             sizes[5], M, C, J(5), ptr(dx16), st)
        if s16:
            _twin_put(dx, dx16)
        _reduce_jobs(jobs, [ws, aws], leaf)
        grads = (dx, None, None, None, None, None, None, None, None, None, None, dg1, dbt1, dwqkv, dbqkv, dwp, dbp, dg2, dbt2, dw1, db1,
                 dw2, db2)
        return grads + tuple(d.view(d.shape[0], 1, 3, 3) for d in dlw) + tuple(dlb)


def cswin_block(x, reso, split, idx, heads, scale, norm1, qkv, proj, norm2, fc1, fc2, lepe_w, lepe_b, rs1=None, rs2=None, attn_drop=0.0):
    """Fused CSWinBlock forward/backward.  norm*/qkv/proj/fc*: nn.Modules holding the parameters.  attn_drop: dropout probability
    on the attention probabilities (pass 0 outside training)."""
    return _CSWinBlock.apply(x, reso, split, tuple(idx), tuple(heads), scale, norm1.eps, norm2.eps, rs1, rs2, _attn_drop(attn_drop),
                             norm1.weight, norm1.bias, qkv.weight, qkv.bias, proj.weight, proj.bias, norm2.weight,
                             norm2.bias, fc1.weight, fc1.bias, fc2.weight, fc2.bias, *lepe_w, *lepe_b)


# ------------------------------------------------------------------------------------------------
# convolutions on tokens
# ------------------------------------------------------------------------------------------------
def _permute_w(w, cpad, want_t):
    Cout, Cin, ks, _ = w.shape
    wp = torch.empty(Cout, ks * ks, cpad, dtype=torch.float32, device=w.device)
    wpt = torch.empty(ks * ks, Cout, cpad, dtype=torch.float32, device=w.device) if want_t else None
    call("cswin_conv_weight_permute", ptr(w), ptr(wp), ptr(wpt), Cout, Cin, ks, cpad, stream())
    return wp, wpt


class _ConvTokens(Function):
    @staticmethod
    def forward(ctx, x, w, b, H, W, stride, pad):
        x, w, b = dev_f32(x, "conv input"), dev_f32(w), dev_f32(b)
        B, L, Cin = x.shape
        assert L == H * W and w.shape[1] == Cin
        Cout, ks = w.shape[0], w.shape[2]
        OH, OW = (H + 2 * pad - ks) // stride + 1, (W + 2 * pad - ks) // stride + 1
        need_dx = ctx.needs_input_grad[0]
        # stride-1 "same" convolutions take their data gradient as a forward convolution of dy (see backward): no transposed image
        same = stride == 1 and 2 * pad == ks - 1 and Cout % 4 == 0
        wp, wpt = _permute_w(w, Cin, need_dx and not same)   # both weight images in one launch; the transposed one is kept
        y = torch.empty(B, OH * OW, Cout, dtype=torch.float32, device=x.device)
        call("cswin_conv_tok_fwd", ptr(x), ptr(wp), ptr(b), ptr(y), B, H, W, Cin, Cout, ks, stride, pad, precision(), stream())
        ctx.save_for_backward(x, w, wpt)
        ctx.meta = (H, W, stride, pad, b is not None)
        ctx.leaf = _all_leaf(w, b)
        ctx.bias_ptr = _pptr(b) if ctx.leaf else 0
        return y

    @staticmethod
    @once_differentiable
    def backward(ctx, dy):
        x, w, wpt = ctx.saved_tensors
        H, W, stride, pad, has_b = ctx.meta
        dy = dev_f32(dy)
        B, L, Cin = x.shape
        Cout, ks = w.shape[0], w.shape[2]
        st = stream()
        dx = None
        if ctx.needs_input_grad[0]:
            dx = torch.empty_like(x)
            if stride == 1 and 2 * pad == ks - 1 and Cout % 4 == 0:
                # dx = conv(dy, mirrored / transposed weights): both operands r-contiguous on the forward kernel, instead of
                # the generic transposed gather (CARAFE4 encoder, 16 -> 144 channels at 56 x 56: 13.41 -> 13.36 ms per step)
                wf = torch.empty(Cin, ks * ks, Cout, dtype=torch.float32, device=x.device)
                call("cswin_conv_weight_flipT", ptr(w), ptr(wf), Cout, Cin, ks, st)
                call("cswin_conv_tok_fwd", ptr(dy), ptr(wf), None, ptr(dx), B, H, W, Cout, Cin, ks, 1, pad, precision(), st)
            else:
                if wpt is None:
                    _, wpt = _permute_w(w, Cin, True)
                call("cswin_conv_tok_bwd_data", ptr(dy), ptr(wpt), ptr(dx), B, H, W, Cin, Cout, ks, stride, pad, precision(), st)
        if True:
            dw = _grad_like(w, ctx.leaf)                # written in the parameter layout by the slab reduction itself
            db = _grad_at(ctx.bias_ptr, (Cout,), x.device) if has_b else None
            nbytes = lib().cswin_conv_tok_bwd_weight_workspace(B, H, W, Cin, Cout, ks, stride, pad)
            ws = _ws(nbytes, x.device)
            job = (ReduceJob * 1)()
            call("cswin_conv_tok_bwd_weight", ptr(dy), ptr(x), ptr(dw), ptr(db), ptr(ws), nbytes, B, H, W, Cin, Cout, ks,
                 stride, pad, 1, ctypes.cast(job, ctypes.c_void_p), precision(), stream())
            _reduce_jobs(job, (ws,), ctx.leaf)
        return dx, dw, db, None, None, None, None


def conv_tokens(x, w, b, H, W, stride, pad):
    """nn.Conv2d(w, b, stride, pad) applied to tokens x (B, H*W, Cin) -> (B, OH*OW, Cout)."""
    return _ConvTokens.apply(x, w, b, H, W, stride, pad)


class _PatchEmbedConv(Function):
    """Conv2d(in_chans, E, 7, 4, 2) on an NCHW image -> tokens (cswin_unet.py:339-340); no input gradient."""

    @staticmethod
    def forward(ctx, img, w, b, stride, pad):
        img, w, b = dev_f32(img, "image"), dev_f32(w), dev_f32(b)
        B, Cin, H, W = img.shape
        Cout, ks = w.shape[0], w.shape[2]
        cpad = (Cin + 3) // 4 * 4
        st = stream()
        x = torch.empty(B, H * W, cpad, dtype=torch.float32, device=img.device)
        call("cswin_nchw_to_tokens", ptr(img), ptr(x), B, Cin, H, W, cpad, st)
        wp, _ = _permute_w(w, cpad, False)
        OH, OW = (H + 2 * pad - ks) // stride + 1, (W + 2 * pad - ks) // stride + 1
        y = torch.empty(B, OH * OW, Cout, dtype=torch.float32, device=img.device)
        call("cswin_conv_tok_fwd", ptr(x), ptr(wp), ptr(b), ptr(y), B, H, W, cpad, Cout, ks, stride, pad, precision(), st)
        ctx.save_for_backward(x, w)
        ctx.meta = (H, W, stride, pad, cpad, b is not None)
        ctx.leaf = _all_leaf(w, b)
        ctx.bias_ptr = _pptr(b) if ctx.leaf else 0
        return y

    @staticmethod
    @once_differentiable
    def backward(ctx, dy):
        x, w = ctx.saved_tensors
        H, W, stride, pad, cpad, has_b = ctx.meta
        dy = dev_f32(dy)
        B = x.shape[0]
        Cout, Cin, ks, _ = w.shape
        st = stream()
        dwp = torch.empty(Cout, ks * ks, cpad, dtype=torch.float32, device=x.device)
        db = _grad_at(ctx.bias_ptr, (Cout,), x.device) if has_b else None
        nbytes = lib().cswin_conv_tok_bwd_weight_workspace(B, H, W, cpad, Cout, ks, stride, pad)
        ws = _ws(nbytes, x.device)
        call("cswin_conv_tok_bwd_weight", ptr(dy), ptr(x), ptr(dwp), ptr(db), ptr(ws), nbytes, B, H, W, cpad, Cout, ks, stride,
             pad, 0, None, precision(), st)                          # channel-padded image (3 -> 4): unpermuted separately, right away
        dw = _grad_like(w, ctx.leaf)
        call("cswin_conv_weight_unpermute", ptr(dwp), ptr(dw), Cout, Cin, ks, cpad, st)
        return None, dw, db, None, None


def patch_embed_conv(img, w, b, stride=4, pad=2):
    return _PatchEmbedConv.apply(img, w, b, stride, pad)


# ------------------------------------------------------------------------------------------------
# CARAFE reassembly, layout adapters
# ------------------------------------------------------------------------------------------------
class _CarafeReassemble(Function):
    @staticmethod
    def forward(ctx, e, z, bias, H, W, S):
        e, z, bias = dev_f32(e, "carafe kernel logits"), dev_f32(z, "carafe features"), dev_f32(bias)
        B, L, Cz = z.shape
        assert L == H * W and e.shape == (B, L, 9 * S * S)
        out = torch.empty(B, L * S * S, Cz, dtype=torch.float32, device=z.device)
        wt = torch.empty_like(e)
        call("cswin_carafe_fwd", ptr(e), ptr(z), ptr(bias), ptr(out), ptr(wt), B, H, W, Cz, S, stream())
        ctx.save_for_backward(z, wt)
        ctx.meta = (H, W, S, bias is not None)
        ctx.bias_ptr = _pptr(bias) if _all_leaf(bias) else 0
        return out

    @staticmethod
    @once_differentiable
    def backward(ctx, dout):
        z, wt = ctx.saved_tensors
        H, W, S, has_b = ctx.meta
        dout = dev_f32(dout)
        B, L, Cz = z.shape
        de = torch.empty_like(wt)
        dz = torch.empty_like(z)
        db = _grad_at(ctx.bias_ptr, (Cz,), z.device) if has_b else None
        nbytes = lib().cswin_carafe_bwd_workspace(B, H, W, Cz, S)
        ws = _ws(nbytes, z.device)
        job = (ReduceJob * 1)()
        call("cswin_carafe_bwd", ptr(dout), ptr(z), ptr(wt), ptr(de), ptr(dz), ptr(db), ptr(ws), nbytes, B, H, W, Cz, S,
             ctypes.cast(job, ctypes.c_void_p), stream())
        _reduce_jobs(job, [ws], leaf=ctx.bias_ptr != 0)      # the bias gradient's partial sums: queued when the bias is a leaf parameter
        return de, dz, db, None, None, None


def carafe_reassemble(e, z, bias, H, W, S):
    return _CarafeReassemble.apply(e, z, bias, H, W, S)


class _TokensToNchw(Function):
    @staticmethod
    def forward(ctx, x, C, H, W):
        x = dev_f32(x)
        B, L, Cpad = x.shape
        assert L == H * W and C <= Cpad
        y = torch.empty(B, C, H, W, dtype=torch.float32, device=x.device)
        call("cswin_tokens_to_nchw", ptr(x), ptr(y), B, C, H, W, Cpad, stream())
        ctx.meta = (C, H, W, Cpad)
        return y

    @staticmethod
    @once_differentiable
    def backward(ctx, dy):
        C, H, W, Cpad = ctx.meta
        dy = dev_f32(dy)
        B = dy.shape[0]
        dx = torch.empty(B, H * W, Cpad, dtype=torch.float32, device=dy.device)
        call("cswin_nchw_to_tokens", ptr(dy), ptr(dx), B, C, H, W, Cpad, stream())
        return dx, None, None, None


def tokens_to_nchw(x, C, H, W):
    """(B, H*W, Cpad) tokens -> (B, C, H, W) taking the first C channels."""
    return _TokensToNchw.apply(x, C, H, W)


def img2windows(img, H_sp, W_sp):
    """img (B, C, H, W) -> (B*nH*nW, H_sp*W_sp, C)   (cswin_unet.py:184-191); index-only."""
    img = dev_f32(img, "img2windows input")
    B, C, H, W = img.shape
    out = torch.empty(B * (H // H_sp) * (W // W_sp), H_sp * W_sp, C, dtype=torch.float32, device=img.device)
    call("cswin_img2windows", ptr(img), ptr(out), B, C, H, W, H_sp, W_sp, stream())
    return out


def windows2img(img_splits_hw, H_sp, W_sp, H, W):
    """(B', H_sp*W_sp, C) windows -> (B, H, W, C)   (cswin_unet.py:194-202); index-only."""
    x = dev_f32(img_splits_hw, "windows2img input")
    C = x.shape[-1]
    B = int(x.shape[0] / (H * W / H_sp / W_sp))
    out = torch.empty(B, H, W, C, dtype=torch.float32, device=x.device)
    call("cswin_windows2img", ptr(x), ptr(out), B, C, H, W, H_sp, W_sp, stream())
    return out


# ------------------------------------------------------------------------------------------------
# loss
# ------------------------------------------------------------------------------------------------
class _CeDiceLoss(Function):
    @staticmethod
    def forward(ctx, logits, labels, w_ce, w_dice, group, probs, class_weight):
        logits = dev_f32(logits, "logits")
        class_weight = dev_f32(class_weight)
        labels = labels.contiguous()
        if labels.dtype != torch.int64:
            labels = labels.long()
        B, ncls = logits.shape[:2]
        HW = logits.numel() // (B * ncls)
        dev = logits.device
        st = stream()
        nbytes = lib().cswin_loss_workspace(B, ncls, HW)
        ws = _ws(nbytes, dev)
        sums = torch.empty(1 + 3 * ncls, dtype=torch.float32, device=dev)
        call("cswin_loss_sums", ptr(logits), ptr(labels), ptr(sums), ptr(ws), nbytes, B, ncls, HW, int(probs), st)
        world = 1
        if group is not None:
            import torch.distributed as dist
            world = dist.get_world_size(group)
            if world > 1:
                dist.all_reduce(sums, group=group)      # 1 + 3*ncls floats: the reference's global-batch Dice
        out = torch.empty(3, dtype=torch.float32, device=dev)
        coef = torch.empty(2 * ncls, dtype=torch.float32, device=dev)
        call("cswin_loss_finalize", ptr(sums), ptr(out), ptr(coef), float(B * HW * world), ncls, w_ce, w_dice, ptr(class_weight),
             stream())
        ctx.save_for_backward(logits, labels, coef)
        # gradients are averaged over ranks afterwards: local CE mean -> ce/(B*HW); global Dice -> * world
        ctx.meta = (w_ce / float(B * HW), w_dice / ncls * world, int(probs))
        loss = out[0].clone()
        ctx.mark_non_differentiable(out)
        return loss, out

    @staticmethod
    @once_differentiable
    def backward(ctx, gloss, _gout):
        logits, labels, coef = ctx.saved_tensors
        ce_scale, dice_scale, probs = ctx.meta
        B, ncls = logits.shape[:2]
        HW = logits.numel() // (B * ncls)
        gloss = dev_f32(gloss.reshape(1))
        dlogits = torch.empty_like(logits)
        call("cswin_loss_bwd", ptr(logits), ptr(labels), ptr(coef), ptr(gloss), ptr(dlogits), ce_scale, dice_scale, B, ncls,
             HW, probs, stream())
        return dlogits, None, None, None, None, None, None


def ce_dice_loss(logits, labels, w_ce=0.4, w_dice=0.6, group=None, inputs_are_probs=False, class_weight=None):
    """0.4*CE + 0.6*Dice (trainer.py:55-57).  Returns (loss, stats) with stats = [loss, ce, dice] (no host sync).
    With a process group the 1+3*ncls Dice/CE sums are all-reduced so Dice is the global-batch Dice.
    inputs_are_probs: the input holds probabilities (DiceLoss(softmax=False)); class_weight: (ncls,) device tensor or None."""
    if inputs_are_probs and w_ce != 0.0:
        raise ValueError("ce_dice_loss: cross entropy needs logits; with inputs_are_probs pass w_ce=0")
    return _CeDiceLoss.apply(logits, labels, float(w_ce), float(w_dice), group, bool(inputs_are_probs), class_weight)


# ------------------------------------------------------------------------------------------------
# dropout (+ residual add and DropPath factor)
# ------------------------------------------------------------------------------------------------
class _Dropout(Function):
    @staticmethod
    def forward(ctx, x, residual, row_scale, p, seed):
        x, residual, row_scale = dev_f32(x, "dropout input"), dev_f32(residual), dev_f32(row_scale)
        y = torch.empty_like(x)
        eps = x.numel() // x.shape[0]
        call("cswin_dropout", ptr(x), ptr(residual), ptr(row_scale), ptr(y), x.numel(), eps, float(p), int(seed), ptr(dropout_epoch(x.device)), stream())
        ctx.save_for_backward(row_scale)
        ctx.meta = (float(p), int(seed), eps, residual is not None)
        return y

    @staticmethod
    @once_differentiable
    def backward(ctx, dy):
        (row_scale,) = ctx.saved_tensors
        p, seed, eps, has_res = ctx.meta
        dy = dev_f32(dy)
        dx = torch.empty_like(dy)
        call("cswin_dropout", ptr(dy), None, ptr(row_scale), ptr(dx), dy.numel(), eps, p, seed, ptr(dropout_epoch(dy.device)), stream())
        return dx, (dy if has_res else None), None, None, None


def dropout(x, p, residual=None, row_scale=None, seed=None):
    """residual + row_scale[sample] * dropout_p(x)  (nn.Dropout followed by the block's residual add, cswin_unet.py:27,135,178-179).
    The keep mask is a counter-based hash of (seed, element index); backward regenerates it."""
    if seed is None:
        seed = int(torch.randint(0, 2 ** 62, (1,)).item())        # host RNG (torch.manual_seed controls it); not capturable
    return _Dropout.apply(x, residual, row_scale, p, seed)
