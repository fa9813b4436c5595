"""Differentiable HIP ops: each torch.autograd.Function below runs a HIP kernel forward and HIP kernels backward.

torch.autograd is used as the tape (which op ran on which tensors, in which order -- including back-propagation through
the autoregressive re-feed) and for gradient accumulation at fan-out points; every per-token computation is in
libtante_hip.so.  LayerNorm's affine and the FiLM tables are tiny parameter-sized expressions (C values) that the host
evaluates with torch so that their gradients come for free.
"""
from __future__ import annotations

import ctypes as C
from typing import Optional

import torch
from torch.autograd import Function

from . import _lib as L
from . import options as _O
from . import kernels as K

_DT = {torch.float32: L.F32, torch.bfloat16: L.BF16}

# Packed copies of a weight are valid until the weight changes; within one train step (4 BPTT calls x forward + backward) the
# same (tensor, layout) is needed many times.  Keyed by the source tensor's identity and version; cleared by the optimiser.
_PACKS = {}


def clear_pack_cache():
    _PACKS.clear()


def _packed(W, b, compute, layout=L.W_LINEAR, **kw):
    key = (W.data_ptr(), W._version, tuple(W.shape), None if b is None else (b.data_ptr(), b._version), compute, layout,
           tuple(sorted(kw.items())))
    hit = _PACKS.get(key)
    if hit is None:
        hit = K.pack_weight(W, b, compute, layout, **kw)
        _PACKS[key] = (hit, W, b)        # keep the sources alive so that data_ptr stays unique
        return hit
    return hit[0]


_s = K._stream


def _rm_linear(t: torch.Tensor, n0: Optional[int] = None, s1: int = 0, s0: Optional[int] = None, off: int = 0, es: int = 1,
               rows: Optional[int] = None, cols: Optional[int] = None) -> L.RowMat:
    m = L.RowMat()
    m.p, m.dtype, m.mode = t.data_ptr(), _DT[t.dtype], L.A_LINEAR
    cols = t.shape[-1] if cols is None else cols
    m.n0 = (rows if rows is not None else t.numel() // cols) if n0 is None else n0
    m.s1, m.s0, m.off, m.es = s1, (cols if s0 is None else s0), off, es
    return m


def _rm_patch(t: torch.Tensor, nhwc: bool, n_img: int, Hin: int, Win: int, Cin: int, P: int) -> L.RowMat:
    m = L.RowMat()
    m.p, m.dtype, m.mode = t.data_ptr(), _DT[t.dtype], (L.A_PATCH_NHWC if nhwc else L.A_PATCH_NCHW)
    m.n0, m.s1, m.s0, m.off, m.es = n_img, 0, 0, 0, 1
    m.Hin, m.Win, m.Cin, m.P = Hin, Win, Cin, P
    return m


_AXIS_WS = {}


def _axis_wgrad_workspace(device) -> torch.Tensor:
    """Per-device slab workspace of tante_axis_wgrad_ws: zero on first use, and every call leaves it zeroed (kernels on one stream)."""
    key = (device.type, device.index, K._stream())
    ws = _AXIS_WS.get(key)
    if ws is None:
        ws = _AXIS_WS[key] = torch.zeros(L.lib().tante_axis_wgrad_workspace_bytes() // 4, dtype=torch.float32, device=device)
    return ws


ACCUMULATE_INTO_GRAD = True   # weight-gradient kernels add straight into a pre-allocated p.grad (FlatAdamW's bucket views)


def _grad_slot(p) -> Optional[torch.Tensor]:
    """The parameter's existing .grad when the weight-gradient kernels may accumulate into it directly.  With BPTT a weight is used
    once per rollout step; letting autograd's AccumulateGrad add the per-use gradients costs an elementwise add (and a zero-filled
    temporary) per use per parameter -- ~800 tiny launches per train step.  Backward then returns None for that input."""
    if not ACCUMULATE_INTO_GRAD or p is None:
        return None
    g = getattr(p, "_tante_grad", None)      # a derived weight with its own accumulator (FoldFn)
    if g is None:
        if not isinstance(p, torch.nn.Parameter) or p.grad is None:
            return None
        g = p.grad
    return g if (g.dtype == torch.float32 and g.is_contiguous() and g.shape == p.shape and g.is_cuda) else None


# ---- weight gradients on a side stream -------------------------------------------------------------------------------------------
# The weight-gradient kernel of a layer and the data-gradient GEMM of the same layer are independent (both read dY), each is bound by
# latency / lockstep rather than by a saturated unit, and their resources add up to exactly one CU (3 x 32 KB + 64 KB of LDS, 4 x 128
# VGPRs per lane): issued on two streams they overlap.  Only the accumulate-into-slot form runs there (nothing on the main stream reads
# the slot before the optimiser); the main stream re-joins at the end of the backward pass through an engine callback.
# Measured on the cfg3 train step (alternating runs on one box): 29.5-29.9 ms against 30.0-30.1 ms on one stream, but with outliers at
# 33-34.6 ms when the two streams' workgroups interleave badly -- a small, unreliable gain, so it is opt-in (TANTE_WGRAD_SIDE_STREAM=1).
SIDE_STREAM_WGRAD = _O.register("TANTE_WGRAD_SIDE_STREAM", False, __name__, "SIDE_STREAM_WGRAD")
_SIDE = {"stream": None, "armed": False, "task": -1}


def _join_side():
    _SIDE["armed"] = False
    if _SIDE["stream"] is not None:
        torch.cuda.current_stream().wait_stream(_SIDE["stream"])


def join_side_stream():
    """Make the current stream wait for every weight-gradient kernel issued so far (before reading gradient slots)."""
    _join_side()


class _side_wgrad:
    """with _side_wgrad(dy, a): wgrad(..., into=...)  -- runs the body on the side stream after everything issued so far."""

    def __init__(self, *tensors):
        self.tensors = tensors

    def __enter__(self):
        if not SIDE_STREAM_WGRAD:
            self.ctxm = None
            return self
        if _SIDE["stream"] is None:
            _SIDE["stream"] = torch.cuda.Stream()
        side = _SIDE["stream"]
        side.wait_stream(torch.cuda.current_stream())
        for t in self.tensors:
            t.record_stream(side)
        task = _graph_task()
        if not _SIDE["armed"] or _SIDE["task"] != task:   # a callback queued by a backward pass that died is gone with it
            _SIDE["armed"], _SIDE["task"] = True, task
            try:
                torch.autograd.Variable._execution_engine.queue_callback(_join_side)
            except RuntimeError:      # not inside a backward pass
                _SIDE["armed"] = False
        self.ctxm = torch.cuda.stream(side)
        self.ctxm.__enter__()
        return self

    def __exit__(self, *exc):
        if self.ctxm is not None:
            self.ctxm.__exit__(*exc)
            if not _SIDE["armed"]:
                _join_side()
        return False


# ---- deferred weight gradients -------------------------------------------------------------------------------------------------------
# With BPTT a weight is used once per rollout call, and each use's weight-gradient kernel pays its own launch (4-8 us of GPU time
# however short), its own pipeline ramp and its own atomic epilogue (64 KiB of fp32 atomics per workgroup, ~11 us per call in situ).
# The dense bf16 linears therefore only RECORD (dY, A) in backward; when the backward pass ends (engine callback) -- or when a FoldFn
# needs its accumulators -- the uses of one weight run as ONE tante_wgrad_multi launch over the concatenated row range.  The recorded
# operands stay alive a little longer (~3 GB at cfg3 on a 288 GB part).
DEFER_WGRAD = _O.register("TANTE_WGRAD_DEFER", True, __name__, "DEFER_WGRAD")
DEFER_MAX_GB = _O.register("TANTE_WGRAD_DEFER_MAX_GB", 32.0, __name__, "DEFER_MAX_GB")
# (recorded operands held at most DEFER_MAX_GB GiB)
_DEFER = {"pending": {}, "armed": False, "bytes": 0, "task": -1}


def _graph_task() -> int:
    """Id of the backward pass (autograd graph task) this thread is executing, -1 outside one."""
    return torch._C._current_graph_task_id()


_FOLD_DIRTY = {}     # id -> accumulator buffer of a FoldFn whose backward has not consumed (and thereby zeroed) it yet

# id(frame encoding) -> the gradient accumulator FilmPosFramesFn nodes of ONE backward pass share (see its backward)
_FRAME_ACC = {"task": -1, "acc": {}}


def _drop_frame_accumulators():
    _FRAME_ACC["task"] = -1
    _FRAME_ACC["acc"] = {}


def _frame_accumulators():
    """The table of the backward pass this thread is executing (created on its first use, dropped by an engine callback when the pass
    ends and by reset_backward_state); None outside a backward pass: nothing is shared then, every node returns its own gradient."""
    task = _graph_task()
    if task < 0:
        return None
    if _FRAME_ACC["task"] != task:
        _FRAME_ACC["task"], _FRAME_ACC["acc"] = task, {}
        try:
            torch.autograd.Variable._execution_engine.queue_callback(_drop_frame_accumulators)
        except RuntimeError:
            _drop_frame_accumulators()
            return None
    return _FRAME_ACC["acc"]


def reset_backward_state(after: bool = False):
    """Forget everything recorded for a backward pass that is not running any more.  'armed' means "the engine's end-of-backward
    callback of THIS backward pass is queued"; a backward pass that raises (OOM, a kernel error, KeyboardInterrupt) drops its
    callbacks, so the flag is keyed to the graph task and additionally cleared here: train_step* call this before and -- in a
    finally -- after every loss.backward(), so operands recorded by a dead pass can never reach a later step's gradient slots."""
    _DEFER["pending"].clear()
    _FOLD_PENDING.clear()
    _drop_frame_accumulators()
    _DEFER["bytes"] = 0
    _DEFER["armed"] = False
    _DEFER["task"] = -1
    _SIDE["armed"] = False
    _SIDE["task"] = -1
    if _SIDE["stream"] is not None:
        torch.cuda.current_stream().wait_stream(_SIDE["stream"])
    if after and _FOLD_DIRTY:
        # fold accumulators live across steps and are zeroed by the fold's own backward kernel; whatever a dead (or partial) backward
        # pass left non-empty is cleared here, so the next step starts from zeros in every case
        for buf in list(_FOLD_DIRTY.values()):
            buf.zero_()
        _FOLD_DIRTY.clear()


def run_backward(loss: torch.Tensor):
    """loss.backward() with the deferred / side-stream state guaranteed clean on entry and on ANY exit."""
    reset_backward_state()
    try:
        loss.backward()
    finally:
        reset_backward_state(after=True)


_WG_WS = {}      # device index -> 64 MiB scratch of the shared weight-gradient launches (split-R partial tiles, summed by a second kernel)


def _wgrad_workspace(device) -> torch.Tensor:
    ws = _WG_WS.get(device.index)
    if ws is None:
        ws = _WG_WS[device.index] = torch.empty(64 << 20, dtype=torch.uint8, device=device)
    return ws


WGRAD_JOBS = _O.register("TANTE_WGRAD_JOBS_PER_LAUNCH", 4, __name__, "WGRAD_JOBS")      # weights per shared launch (1: one launch per weight)


def _flush_wgrads(slot: Optional[torch.Tensor] = None):
    pend = _DEFER["pending"]
    keys = [k for k in pend if slot is None or k[0] == slot.data_ptr()]
    if slot is None and WGRAD_JOBS > 1 and len(keys) > 1:
        # the end-of-pass flush: consecutive recorded weights (the four of a block sit next to each other) as the jobs of one launch
        ents = [pend.pop(k) for k in keys]
        dev = ents[0][0].device
        ws = _wgrad_workspace(dev)
        i = 0
        while i < len(ents):
            grp = ents[i: i + WGRAD_JOBS]
            if any(e[5] != grp[0][5] or e[0].device != dev for e in grp):      # mixed compute modes / devices: one at a time
                grp = grp[:1]
            i += len(grp)
            jobs = (L.WgradJob * len(grp))()
            keep = []
            for jb, (gW, gb, M, N, Kk, comp, lay, uses) in zip(jobs, grp):
                n = len(uses)
                U = (L.RowMat * n)(*[_rm_linear(dy) for dy, _ in uses])
                V = (L.RowMat * n)(*[_rm_linear(a) for _, a in uses])
                keep.append((U, V))
                jb.U, jb.V, jb.n_seg, jb.R, jb.I, jb.J = U, V, n, M, N, Kk
                jb.dW, jb.dbias = gW.data_ptr(), None if gb is None else gb.data_ptr()
                jb.layout, jb.P, jb.C_other, jb.swap = lay[0], lay[1], lay[2], int(lay[3])
            L.check(L.lib().tante_wgrad_jobs_ws(jobs, len(grp), grp[0][5], ws.data_ptr(), ws.numel(), _s()), "tante_wgrad_jobs")
        keys = []
    for k in keys:
        gW, gb, M, N, Kk, comp, lay, uses = pend.pop(k)
        n = len(uses)
        U = (L.RowMat * n)(*[_rm_linear(dy) for dy, _ in uses])
        V = (L.RowMat * n)(*[_rm_linear(a) for _, a in uses])
        layout, P, Co, swap = lay
        ws = _wgrad_workspace(gW.device)
        L.check(L.lib().tante_wgrad_multi_ws(C.byref(U), C.byref(V), n, M, N, Kk, gW.data_ptr(), None if gb is None else gb.data_ptr(), layout, P,
                                             Co, int(swap), comp, 1, ws.data_ptr(), ws.numel(), _s()), "tante_wgrad_multi")
    if slot is None:
        _DEFER["armed"] = False
        _DEFER["task"] = -1
        _DEFER["bytes"] = 0


_FOLD_PENDING = []      # folds whose backward waits for the end of the pass: (acc buffer, GW, Gb, W, gamma, beta, dW, db, dgamma, dbeta, N, K)


def _flush_folds():
    """The recorded FoldFn backwards as ONE launch (tante_fold_bwd_multi), after the weight-gradient launches that fill their accumulators."""
    if not _FOLD_PENDING:
        return
    if _SIDE["stream"] is not None:      # the accumulators are written by weight-gradient kernels on the side stream
        torch.cuda.current_stream().wait_stream(_SIDE["stream"])
    n = len(_FOLD_PENDING)
    arr = (L.Fold * n)()
    for f, (buf, GW, Gb, W, gamma, beta, dW, db, dg, dbt, N, Kk) in zip(arr, _FOLD_PENDING):
        f.GW, f.Gb, f.W, f.gamma, f.beta = GW.data_ptr(), Gb.data_ptr(), W.data_ptr(), gamma.data_ptr(), beta.data_ptr()
        f.dW, f.db, f.dgamma, f.dbeta = dW.data_ptr(), None if db is None else db.data_ptr(), dg.data_ptr(), dbt.data_ptr()
        f.N, f.K = N, Kk
    L.check(L.lib().tante_fold_bwd_multi(C.byref(arr), n, 1, _s()), "tante_fold_bwd_multi")
    for ent in _FOLD_PENDING:
        _FOLD_DIRTY.pop(id(ent[0]), None)      # the kernel left the accumulators zeroed
    _FOLD_PENDING.clear()


PRE_FLUSH_HOOK = [None]      # callable() run before the end-of-pass flush (train.py: the early part of the gradient all-reduce)
HOLD_FLUSH = [False]         # True: the engine callback leaves the recorded work alone (train.GraphedTrainStep flushes in a graph of its own)


def flush_write_range(bucket: torch.Tensor):
    """(first element, end element) of `bucket` (a flat gradient buffer) that the pending flush will write: the deferred linears' weight /
    bias slots and the folds' four outputs, as far as they are views of the bucket (the folded weights' own accumulators are private
    buffers).  None when the flush writes nothing inside it."""
    base, esz = bucket.data_ptr(), bucket.element_size()
    end = base + bucket.numel() * esz
    lo, hi = None, 0

    def add(t):
        nonlocal lo, hi
        if t is None:
            return
        a = t.data_ptr()
        if a < base or a >= end:
            return
        lo = a if lo is None else min(lo, a)
        hi = max(hi, a + t.numel() * t.element_size())
    for ent in _DEFER["pending"].values():
        add(ent[0])
        add(ent[1])
    for ent in _FOLD_PENDING:
        for t in ent[6:10]:
            add(t)
    return None if lo is None else ((lo - base) // esz, (min(hi, end) - base) // esz)


# ---- the end-of-pass flush in SEGMENTS (round 6: the data-parallel all-reduce pipelined against it, dist.GradAllReduce) ------------------
# The flush is ~11 shared weight-gradient launches (one per block: its four weights) followed by the LayerNorm folds, which distribute the
# folded in-projection / fc1 gradients to W, b, gamma, beta -- 58 % of the bucket, final only at the very end when the folds run as ONE
# launch.  In segments, every fold runs right behind the launch group that fills its accumulator, so a block's whole span of the bucket is
# final when its group is: the spans of the first segments travel while the later segments still compute.
def _pending_groups():
    ents = list(_DEFER["pending"].values())
    groups = []
    if WGRAD_JOBS > 1 and len(ents) > 1:
        i = 0
        while i < len(ents):
            grp = ents[i: i + WGRAD_JOBS]
            if any(e[5] != grp[0][5] or e[0].device != grp[0][0].device for e in grp):
                grp = grp[:1]
            i += len(grp)
            groups.append(grp)
    else:
        groups = [[e] for e in ents]
    # fold f is ready behind the last group that writes its accumulator (group -1: nothing pending writes it)
    ready = {}
    for fi, f in enumerate(_FOLD_PENDING):
        last = -1
        for gi, grp in enumerate(groups):
            if any(e[0].data_ptr() == f[1].data_ptr() for e in grp):
                last = gi
        ready.setdefault(max(last, 0), []).append(fi)
    return groups, ready


def _segment_bounds(n_groups: int, segments: int):
    segments = max(1, min(int(segments), max(1, n_groups)))
    return [round(s * n_groups / segments) for s in range(segments + 1)]


def flush_plan(bucket: torch.Tensor, segments: int):
    """What the pending flush will finalise, by segment, WITHOUT launching: (early, [seg_0, seg_1, ...]) -- each a list of disjoint
    (first element, end element) ranges of `bucket`.  `early` = what no segment writes (final already); every element of the bucket is in
    exactly one list.  None when nothing is pending."""
    groups, ready = _pending_groups()
    if not groups and not _FOLD_PENDING:
        return None
    fold_acc = {f[1].data_ptr() for f in _FOLD_PENDING}
    base, esz, n = bucket.data_ptr(), bucket.element_size(), bucket.numel()
    owner = {}      # (lo, hi) -> the LAST segment that writes it
    bounds = _segment_bounds(len(groups), segments)
    seg_of = lambda gi: max(s for s in range(len(bounds) - 1) if bounds[s] <= gi) if groups else 0

    def add(t, seg):
        if t is None:
            return
        a = t.data_ptr()
        if a < base or a >= base + n * esz:
            return
        lo = (a - base) // esz
        owner[(lo, min(n, lo + t.numel()))] = seg
    for gi, grp in enumerate(groups):
        for e in grp:
            if e[0].data_ptr() not in fold_acc:
                add(e[0], seg_of(gi))
                add(e[1], seg_of(gi))
        for fi in ready.get(gi, ()):
            for t in _FOLD_PENDING[fi][6:10]:
                add(t, seg_of(gi))
    if not groups:
        for f in _FOLD_PENDING:
            for t in f[6:10]:
                add(t, 0)
    n_seg = len(bounds) - 1
    segs = [[] for _ in range(n_seg)]
    for (lo, hi), sg in sorted(owner.items()):
        lst = segs[sg]
        if lst and lst[-1][1] == lo:
            lst[-1] = (lst[-1][0], hi)
        else:
            lst.append((lo, hi))
    taken = sorted(r for lst in segs for r in lst)
    early, pos = [], 0
    for lo, hi in taken:
        if lo < pos:
            raise RuntimeError("flush_plan: overlapping gradient slots in the bucket")
        if lo > pos:
            early.append((pos, lo))
        pos = hi
    if pos < n:
        early.append((pos, n))
    return early, segs


def flush_steps(segments: int):
    """Generator form of the flush: every next() launches one segment (its launch groups, each followed by the folds it completes) and
    yields the segment's index; exhausting it resets the recorded state.  (train.GraphedTrainStep captures each next() as a graph.)"""
    groups, ready = _pending_groups()
    bounds = _segment_bounds(len(groups), segments)
    n_seg = len(bounds) - 1
    eager = n_seg > 1
    for s in range(n_seg):
        for gi in range(bounds[s], bounds[s + 1]):
            _launch_group(groups[gi])
            if eager and ready.get(gi):
                _launch_folds([_FOLD_PENDING[fi] for fi in ready[gi]])
        if s == n_seg - 1:      # the last segment also resets the recorded state (before its yield: the caller may never come back)
            _DEFER["pending"].clear()
            _DEFER["armed"], _DEFER["task"], _DEFER["bytes"] = False, -1, 0
            if eager:
                if not groups:
                    _launch_folds(list(_FOLD_PENDING))
                _FOLD_PENDING.clear()
            else:
                _flush_folds()
        yield s


def flush_run(segments: int, on_segment=None):
    """Run the pending flush in `segments` segments (flush_plan's partition), calling on_segment(s) behind each.  One segment is the plain
    flush: the folds as ONE launch at the end."""
    for s in flush_steps(segments):
        if on_segment is not None:
            on_segment(s)


def _launch_group(grp):
    dev = grp[0][0].device
    ws = _wgrad_workspace(dev)
    if WGRAD_JOBS > 1 and len(_DEFER["pending"]) > 1:
        jobs = (L.WgradJob * len(grp))()
        keep = []
        for jb, (gW, gb, M, N, Kk, comp, lay, uses) in zip(jobs, grp):
            n = len(uses)
            U = (L.RowMat * n)(*[_rm_linear(dy) for dy, _ in uses])
            V = (L.RowMat * n)(*[_rm_linear(a) for _, a in uses])
            keep.append((U, V))
            jb.U, jb.V, jb.n_seg, jb.R, jb.I, jb.J = U, V, n, M, N, Kk
            jb.dW, jb.dbias = gW.data_ptr(), None if gb is None else gb.data_ptr()
            jb.layout, jb.P, jb.C_other, jb.swap = lay[0], lay[1], lay[2], int(lay[3])
        L.check(L.lib().tante_wgrad_jobs_ws(jobs, len(grp), grp[0][5], ws.data_ptr(), ws.numel(), _s()), "tante_wgrad_jobs")
        return
    for gW, gb, M, N, Kk, comp, lay, uses in grp:
        n = len(uses)
        U = (L.RowMat * n)(*[_rm_linear(dy) for dy, _ in uses])
        V = (L.RowMat * n)(*[_rm_linear(a) for _, a in uses])
        layout, P, Co, swap = lay
        L.check(L.lib().tante_wgrad_multi_ws(C.byref(U), C.byref(V), n, M, N, Kk, gW.data_ptr(), None if gb is None else gb.data_ptr(), layout, P,
                                             Co, int(swap), comp, 1, ws.data_ptr(), ws.numel(), _s()), "tante_wgrad_multi")


def _launch_folds(folds):
    if not folds:
        return
    if _SIDE["stream"] is not None:
        torch.cuda.current_stream().wait_stream(_SIDE["stream"])
    n = len(folds)
    arr = (L.Fold * n)()
    for f, (buf, GW, Gb, W, gamma, beta, dW, db, dg, dbt, N, Kk) in zip(arr, folds):
        f.GW, f.Gb, f.W, f.gamma, f.beta = GW.data_ptr(), Gb.data_ptr(), W.data_ptr(), gamma.data_ptr(), beta.data_ptr()
        f.dW, f.db, f.dgamma, f.dbeta = dW.data_ptr(), None if db is None else db.data_ptr(), dg.data_ptr(), dbt.data_ptr()
        f.N, f.K = N, Kk
    L.check(L.lib().tante_fold_bwd_multi(C.byref(arr), n, 1, _s()), "tante_fold_bwd_multi")
    for ent in folds:
        _FOLD_DIRTY.pop(id(ent[0]), None)


FLUSH_DRIVER = [None]        # callable() that runs the end-of-pass flush itself (train.py: flush_plan + flush_run around the all-reduce calls)


def flush_deferred_wgrads(force: bool = False):
    """Run every recorded weight-gradient launch now (called automatically at the end of a backward pass), then the folds that wait for them."""
    if HOLD_FLUSH[0] and not force:
        return
    if FLUSH_DRIVER[0] is not None:
        _join_side()
        FLUSH_DRIVER[0]()
        return
    if PRE_FLUSH_HOOK[0] is not None:
        _join_side()                     # (weight gradients issued on the side stream, when that option is on, are part of "final")
        PRE_FLUSH_HOOK[0]()
    _flush_wgrads(None)
    _flush_folds()


def _defer_wgrad(gW, gb, dy, a, M, N, Kk, comp, lay=(L.W_LINEAR, 0, 0, False)) -> bool:
    """Record one use (dW[N x Kk] += dy^T a over M dense bf16 rows; lay = (output layout, P, C_other, swap) as in tante_wgrad);
    False when this use has to run immediately (feature off, shape / dtype outside the shared-launch kernel)."""
    if not DEFER_WGRAD or comp != L.BF16 or dy.dtype != torch.bfloat16 or a.dtype != torch.bfloat16 or not _tr_shape(M, N, Kk):
        return False
    task = _graph_task()
    if not _DEFER["armed"] or _DEFER["task"] != task:
        # first deferred use of THIS backward pass (a different graph task id means the pass that armed the flag is gone, and its
        # queued callback with it: re-queue, and drop whatever it had recorded)
        try:
            torch.autograd.Variable._execution_engine.queue_callback(flush_deferred_wgrads)
        except RuntimeError:          # not inside a backward pass: nothing would flush it
            return False
        _DEFER["pending"].clear()     # leftovers of a backward pass that died half-way must not leak into this one
        _DEFER["bytes"] = 0
        _DEFER["armed"], _DEFER["task"] = True, task
    key = (gW.data_ptr(), M, N, Kk, comp, lay)
    ent = _DEFER["pending"].get(key)
    if ent is None:
        ent = _DEFER["pending"][key] = (gW, gb, M, N, Kk, comp, lay, [])
    ent[7].append((dy, a))
    _DEFER["bytes"] += dy.numel() * dy.element_size() + a.numel() * a.element_size()
    if _DEFER["bytes"] > int(DEFER_MAX_GB * 2 ** 30):      # a very large model / batch: do not sit on more activations than this
        armed, task = _DEFER["armed"], _DEFER["task"]
        _flush_wgrads(None)
        _DEFER["armed"], _DEFER["task"] = armed, task                # the engine callback is still queued for the rest of this backward pass
    return True


def wgrad(U: L.RowMat, V: L.RowMat, R: int, I: int, J: int, out_shape, compute: int, layout: int = L.W_LINEAR, P: int = 0,
          C_other: int = 0, swap: bool = False, device=None, with_bias: bool = False, into: Optional[torch.Tensor] = None,
          db_into: Optional[torch.Tensor] = None):
    """dW (and, with_bias, db[i] = sum_r U[r][i] from the same staged tiles).  `into` / `db_into`: accumulate onto existing tensors."""
    acc = into is not None
    dW = into if acc else torch.empty(out_shape, dtype=torch.float32, device=device)
    db = None
    if with_bias:
        db = db_into if acc else torch.empty(I, dtype=torch.float32, device=device)
    # the scratch buffer of the shared launches serves the single-tile gradients too (split-R partials summed by a second kernel); not when
    # weight gradients run on a side stream beside the main stream's users of the same buffer
    ws = None if (SIDE_STREAM_WGRAD or I > 64 or J > 64 or device is None and into is None) else _wgrad_workspace(dW.device)
    L.check(L.lib().tante_wgrad_ws(C.byref(U), C.byref(V), R, I, J, dW.data_ptr(), None if db is None else db.data_ptr(), layout, P,
                                   C_other, int(swap), compute, int(acc), None if ws is None else ws.data_ptr(), 0 if ws is None else ws.numel(),
                                   _s()), "tante_wgrad")
    return (dW, db) if with_bias else dW


def colsum(x: torch.Tensor, outer: int, Cc: int, inner: int, into: Optional[torch.Tensor] = None) -> torch.Tensor:
    out = into if into is not None else torch.empty(Cc, dtype=torch.float32, device=x.device)
    L.check(L.lib().tante_colsum(x.data_ptr(), _DT[x.dtype], outer, Cc, inner, out.data_ptr(), int(into is not None), _s()), "tante_colsum")
    return out


class LayerNormFn(Function):
    """xhat = (x - mean) / sqrt(var + eps) per row, no affine."""

    @staticmethod
    def forward(ctx, x, eps, out_dtype):
        M, Cc = x.shape
        xh = torch.empty(M, Cc, dtype=out_dtype, device=x.device)
        st = torch.empty(M, 2, dtype=torch.float32, device=x.device)
        L.check(L.lib().tante_layernorm_fwd(x.data_ptr(), M, Cc, eps, xh.data_ptr(), _DT[out_dtype], st.data_ptr(), _s()), "ln_fwd")
        ctx.save_for_backward(x, st)
        return xh

    @staticmethod
    def backward(ctx, g):
        x, st = ctx.saved_tensors
        g = g.contiguous()
        dx = torch.empty_like(x)
        L.check(L.lib().tante_layernorm_bwd(g.data_ptr(), _DT[g.dtype], x.data_ptr(), st.data_ptr(), None, x.shape[0], x.shape[1],
                                            dx.data_ptr(), _s()), "ln_bwd")
        return dx, None, None


class FoldFn(Function):
    """(W diag(gamma), b + W beta): LayerNorm's affine folded into the consumer's weight, with zero-initialised accumulators that the
    weight-gradient kernels of every use add into (`_tante_grad`, see _grad_slot).  Those uses then return no gradient for the folded
    tensors, autograd calls this backward once with nothing, and the accumulated gradient is distributed to W, b, gamma, beta here --
    one HIP launch forward (tante_fold_fwd), one backward (tante_fold_bwd, straight into the parameters' gradient slots).
    Without the accumulators every use of a folded weight produced a fresh dW / db that autograd summed: ~300 tiny launches per step."""

    @staticmethod
    def forward(ctx, W, b, gamma, beta, pre=None):
        N, Kk = W.shape
        if pre is not None:      # (We, be) already computed -- by tante_fold_fwd_multi, for every fold of the model in one launch
            We, be = pre
        else:
            We = torch.empty_like(W)
            be = torch.empty(N, dtype=torch.float32, device=W.device)
            L.check(L.lib().tante_fold_fwd(W.data_ptr(), None if b is None else b.data_ptr(), gamma.data_ptr(), beta.data_ptr(), N, Kk,
                                           We.data_ptr(), be.data_ptr(), _s()), "tante_fold_fwd")
        # both accumulators in one buffer that lives ON the weight across steps: the fold's backward kernel zeroes it while reading
        # (tante_fold_bwd_clear), so the steady state has no fill per weight and step (there were two: here and after the backward)
        acc = getattr(W, "_tante_fold_acc", None)
        if acc is None or acc.numel() != N * Kk + N or acc.device != W.device:
            acc = torch.zeros(N * Kk + N, dtype=torch.float32, device=W.device)
            W._tante_fold_acc = acc
        _FOLD_DIRTY[id(acc)] = acc
        gW, gb = acc[:N * Kk].view(N, Kk), acc[N * Kk:]
        ctx.save_for_backward(W, gamma, beta)
        ctx.acc = (gW, gb)
        ctx.acc_buf = acc
        ctx.params = (W, b, gamma, beta)
        ctx.set_materialize_grads(False)
        ctx.mark_non_differentiable(gW, gb)
        return We, be, gW, gb

    @staticmethod
    def backward(ctx, gWe, gbe, _a, _b):
        W, gamma, beta = ctx.saved_tensors
        GW, Gb = ctx.acc
        N, Kk = W.shape
        if gWe is None and gbe is None and Kk <= 256 and _DEFER["armed"] and _DEFER["task"] == _graph_task():
            # the usual case on the train path: every use added into the accumulators (or is recorded to), nothing arrives through
            # autograd.  The fold joins the others of this backward pass: ONE launch after the recorded weight-gradient launches
            # (flush_deferred_wgrads, the engine's end-of-pass callback that the recorded uses queued)
            slots = [_grad_slot(q) for q in ctx.params]
            if all(g is not None for g, q in zip(slots, ctx.params) if q is not None):
                _FOLD_PENDING.append((ctx.acc_buf, GW, Gb, W, gamma, beta, slots[0], slots[1], slots[2], slots[3], N, Kk))
                return None, None, None, None, None
        _flush_wgrads(GW)                    # the recorded uses of this folded weight run now, as one launch
        if _SIDE["stream"] is not None:      # the accumulators are written by weight-gradient kernels on the side stream
            torch.cuda.current_stream().wait_stream(_SIDE["stream"])
        if gWe is not None:
            GW = GW + gWe
        if gbe is not None:
            Gb = Gb + gbe
        N, Kk = W.shape
        slots = [_grad_slot(q) for q in ctx.params]
        has_b = ctx.params[1] is not None
        direct = all(g is not None for g, q in zip(slots, ctx.params) if q is not None)
        if direct:
            dW, db, dg, dbt = slots
        else:
            dW, dg, dbt = torch.zeros_like(W), torch.zeros_like(gamma), torch.zeros_like(beta)
            db = torch.zeros(N, dtype=torch.float32, device=W.device) if has_b else None
        GW, Gb = GW.contiguous(), Gb.contiguous()
        own = GW.data_ptr() == ctx.acc[0].data_ptr() and Gb.data_ptr() == ctx.acc[1].data_ptr() and Kk <= 256
        fn = L.lib().tante_fold_bwd_clear if own else L.lib().tante_fold_bwd
        L.check(fn(GW.data_ptr(), Gb.data_ptr(), W.data_ptr(), gamma.data_ptr(), beta.data_ptr(), N, Kk, dW.data_ptr(),
                   None if db is None else db.data_ptr(), dg.data_ptr(), dbt.data_ptr(), _s()), "tante_fold_bwd")
        if not own:
            ctx.acc_buf.zero_()   # consumed: a second backward through a retained graph starts from empty accumulators
        _FOLD_DIRTY.pop(id(ctx.acc_buf), None)      # the kernel (or the fill above) left it zeroed
        return (None, None, None, None, None) if direct else (dW, db, dg, dbt, None)


class LayerNormSkipFn(Function):
    """(normalised x, x): the second output is the input itself, to be used as the skip operand of the residual add that closes the
    pre-LN branch.  Its gradient then arrives HERE together with the branch's, and the LayerNorm backward kernel adds it while it writes
    dx (its `dskip` operand) -- otherwise autograd sums the two with a stand-alone elementwise add over the residual stream per LayerNorm."""

    @staticmethod
    def forward(ctx, x, eps, out_dtype, pre=None):
        """pre = (xh, stats) already computed by the fused training-forward kernel (tante_block_fused_train): nothing is launched, the
        node only records what its backward reads.  The same convention holds for LinearFn / AttentionFn / BranchOutFn below: the fused
        forward produces every intermediate in one launch and the autograd graph -- hence the whole backward pass -- stays the unfused one."""
        M, Cc = x.shape
        if pre is not None:
            xh, st = pre
        else:
            xh = torch.empty(M, Cc, dtype=out_dtype, device=x.device)
            st = torch.empty(M, 2, dtype=torch.float32, device=x.device)
            L.check(L.lib().tante_layernorm_fwd(x.data_ptr(), M, Cc, eps, xh.data_ptr(), _DT[out_dtype], st.data_ptr(), _s()), "ln_fwd")
        ctx.save_for_backward(x, st)
        return xh, x.view(M, Cc)

    @staticmethod
    def backward(ctx, g, gskip):
        x, st = ctx.saved_tensors
        if g is None:
            return gskip, None, None, None
        g = g.contiguous()
        if gskip is not None and (gskip.dtype != torch.float32 or not gskip.is_contiguous()):
            gskip = gskip.float().contiguous()
        dx = torch.empty_like(x)
        L.check(L.lib().tante_layernorm_bwd(g.data_ptr(), _DT[g.dtype], x.data_ptr(), st.data_ptr(), None if gskip is None else gskip.data_ptr(),
                                            x.shape[0], x.shape[1], dx.data_ptr(), _s()), "ln_bwd")
        return dx, None, None, None


class LinearFn(Function):
    """y = a @ W^T + b (+ residual).  a (M, K) fp32 / bf16, W (N, K) fp32 master, y in out_dtype (fp32 when a residual is added)."""

    @staticmethod
    def forward(ctx, a, W, b, residual, compute, out_dtype, pre=None):
        M, Kk = a.shape
        N = W.shape[0]
        if pre is not None:
            out = pre
        elif Kk <= 512:
            pw = _packed(W, b, compute)
            out = torch.empty(M, N, dtype=torch.float32 if residual is not None else out_dtype, device=a.device)
            K.linear(a, pw, out, M=M, residual=residual)
        else:   # contraction longer than the GEMM's register-stationary limit: K-chunks accumulate through the fp32 residual operand
            acc = residual
            for c0 in range(0, Kk, 512):
                ck = min(512, Kk - c0)
                pw = K.pack_weight(W.detach()[:, c0:c0 + ck].contiguous(), b if c0 == 0 else None, compute)
                nxt = torch.empty(M, N, dtype=torch.float32, device=a.device)
                K.linear(a, pw, nxt, M=M, a_n0=M, a_s0=Kk, a_off=c0, residual=acc)
                acc = nxt
            out = acc if (residual is not None or out_dtype == torch.float32) else acc.to(out_dtype)
        ctx.save_for_backward(a, W)
        ctx.compute, ctx.has_bias, ctx.has_res = compute, b is not None, residual is not None
        ctx.params = (W, b)
        return out

    @staticmethod
    def backward(ctx, dy):
        a, W = ctx.saved_tensors
        dy = dy.contiguous()
        M, Kk = a.shape
        N = W.shape[0]
        comp = ctx.compute
        da = dW = db = None
        dres = dy if ctx.has_res else None
        if comp == L.BF16 and dy.dtype == torch.float32:
            # the fp32 residual-stream gradient feeds two bf16 GEMMs (dgrad, wgrad): round it once -- both then read half the bytes
            # and the weight gradient takes the LDS-DMA / transposed-read kernel, which wants bf16 rows
            dyb = torch.empty(dy.shape, dtype=torch.bfloat16, device=dy.device)
            L.check(L.lib().tante_act_fwd(dy.data_ptr(), L.F32, dyb.data_ptr(), L.BF16, dy.numel(), L.ACT_NONE, _s()), "tante_act_fwd")
            dy = dyb
        side_done = False
        if ctx.needs_input_grad[1] and ctx.has_bias and ctx.needs_input_grad[2]:
            gW, gb = _grad_slot(ctx.params[0]), _grad_slot(ctx.params[1])
            if gW is not None and gb is not None:
                if not _defer_wgrad(gW, gb, dy, a, M, N, Kk, comp):
                    with _side_wgrad(dy, a):
                        wgrad(_rm_linear(dy), _rm_linear(a), M, N, Kk, (N, Kk), comp, device=a.device, with_bias=True, into=gW, db_into=gb)
                side_done = True
        if ctx.needs_input_grad[0]:       # dgrad GEMM: (M, N) x (N, K); the contraction (N) is chunked to the kernel's K limit
            chunks = [(c0, min(512, N - c0)) for c0 in range(0, N, 512)]
            da = torch.empty(M, Kk, dtype=a.dtype, device=a.device)
            acc = None
            for ci, (c0, ck) in enumerate(chunks):
                pwt = _packed(W[c0:c0 + ck], None, comp, L.W_LINEAR_T, N=Kk, K=ck)
                last = ci == len(chunks) - 1
                out = da if last else (acc if acc is not None else torch.empty(M, Kk, dtype=torch.float32, device=a.device))
                K.linear(dy, pw=pwt, out=out, M=M, a_n0=M, a_s0=N, a_off=c0, residual=acc)
                acc = out
        if side_done:
            pass
        elif ctx.needs_input_grad[1]:
            gW, gb = _grad_slot(ctx.params[0]), _grad_slot(ctx.params[1])
            if gW is not None and ctx.has_bias and ctx.needs_input_grad[2] and gb is not None:
                wgrad(_rm_linear(dy), _rm_linear(a), M, N, Kk, (N, Kk), comp, device=a.device, with_bias=True, into=gW, db_into=gb)
                dW = db = None
            else:
                dW, db = wgrad(_rm_linear(dy), _rm_linear(a), M, N, Kk, (N, Kk), comp, device=a.device, with_bias=True)
                if not (ctx.has_bias and ctx.needs_input_grad[2]):
                    db = None
        elif ctx.has_bias and ctx.needs_input_grad[2]:
            db = colsum(dy, M, N, 1)
        return da, dW, db, dres, None, None, None


FP32_BIAS_SUMS = _O.register("TANTE_TRAIN_FP32_BIAS_SUMS", True, __name__, "FP32_BIAS_SUMS")


class BranchOutFn(Function):
    """out = res + dropout_p(act(pre) @ W^T + b): the closing projection of a residual branch (attention out-proj: act = none; MLP fc2:
    act = GELU on fc1's pre-activation) as ONE differentiable op, so that the GEMM epilogues carry what would otherwise be separate
    passes over the token matrix: the dropout + skip add forward (tante_dropout_add) and the activation backward (tante_act_bwd)."""

    @staticmethod
    def forward(ctx, pre, W, b, res, act, p, compute, done=None):
        M, Kk = pre.shape
        N = W.shape[0]
        adt = K.act_torch_dtype(compute)
        if done is not None:                  # (out, act(pre) or None, dropout seed) from the fused training-forward kernel
            out, a, seed = done
            a = pre if a is None else a
            ctx.save_for_backward(pre, a, W)
            ctx.act, ctx.p, ctx.seed, ctx.compute, ctx.has_bias = act, float(p), seed, compute, b is not None
            ctx.params = (W, b)
            return out
        if act != L.ACT_NONE:
            a = torch.empty(M, Kk, dtype=adt, device=pre.device)
            L.check(L.lib().tante_act_fwd(pre.data_ptr(), _DT[pre.dtype], a.data_ptr(), _DT[adt], pre.numel(), act, _s()), "act_fwd")
        else:
            a = pre
        pw = _packed(W, b, compute)
        seed = next_seed() if p > 0.0 else 0
        fused = Kk <= 512 and K.linear_train_epilogue_ok(a, M, N, Kk, compute)
        out = torch.empty(M, N, dtype=torch.float32, device=pre.device)
        if p > 0.0 and fused:
            K.linear(a, pw, out, M=M, residual=res, drop_p=p, drop_seed=seed)
        elif p > 0.0:
            y = torch.empty(M, N, dtype=adt, device=pre.device)
            K.linear(a, pw, y, M=M)
            L.check(L.lib().tante_dropout_add(y.data_ptr(), _DT[adt], res.data_ptr(), float(p), seed, y.numel(), out.data_ptr(), _s()), "dropout_add")
        else:
            K.linear(a, pw, out, M=M, residual=res)
        ctx.save_for_backward(pre, a, W)
        ctx.act, ctx.p, ctx.seed, ctx.compute, ctx.has_bias = act, float(p), seed, compute, b is not None
        ctx.params = (W, b)
        return out

    @staticmethod
    def backward(ctx, dout):
        pre, a, W = ctx.saved_tensors
        dout = dout.contiguous()
        M, Kk = a.shape
        N = W.shape[0]
        comp = ctx.compute
        adt = K.act_torch_dtype(comp)
        dy = torch.empty(M, N, dtype=adt, device=dout.device)   # gradient of the product, in the activation dtype (read by two GEMMs)
        if ctx.p > 0.0:
            L.check(L.lib().tante_dropout_bwd(dout.data_ptr(), ctx.p, ctx.seed, dout.numel(), dy.data_ptr(), _DT[adt], _s()), "dropout_bwd")
        elif adt == torch.float32:
            dy = dout
        else:
            L.check(L.lib().tante_act_fwd(dout.data_ptr(), L.F32, dy.data_ptr(), _DT[adt], dout.numel(), L.ACT_NONE, _s()), "tante_act_fwd")
        dpre = dW = db = None
        side_done = False
        # The bias gradient is a column sum over every token.  From the bf16 copy (what the weight-gradient kernel stages) each term carries
        # 2^-9 of rounding noise, and where the true sum is a cancellation that noise is a large share of the VALUE (fixture g15: 12 % on a
        # block's fc2 bias).  Without dropout the fp32 gradient is right here: sum that instead (round-5 verdict, weak item 1).
        db32, bias32 = None, False
        if FP32_BIAS_SUMS and ctx.p == 0.0 and adt != torch.float32 and ctx.has_bias and ctx.needs_input_grad[2]:
            bias32 = True
            gb_slot = _grad_slot(ctx.params[1])
            if gb_slot is not None:
                colsum(dout, M, N, 1, into=gb_slot)
            else:
                db32 = colsum(dout, M, N, 1)
        if ctx.needs_input_grad[1] and ctx.has_bias and ctx.needs_input_grad[2]:
            gW, gb = _grad_slot(ctx.params[0]), _grad_slot(ctx.params[1])
            if gW is not None and gb is not None:
                gbk = None if bias32 else gb      # (bias32: the bias gradient has been added above, from the fp32 rows)
                if not _defer_wgrad(gW, gbk, dy, a, M, N, Kk, comp):
                    with _side_wgrad(dy, a):
                        wgrad(_rm_linear(dy), _rm_linear(a), M, N, Kk, (N, Kk), comp, device=a.device, with_bias=not bias32, into=gW,
                              db_into=gbk)
                side_done = True
        if ctx.needs_input_grad[0]:
            dpre = torch.empty(M, Kk, dtype=pre.dtype, device=pre.device)
            if N <= 512:
                pwt = _packed(W, None, comp, L.W_LINEAR_T, N=Kk, K=N)
                if ctx.act != L.ACT_NONE and K.linear_train_epilogue_ok(dy, M, Kk, N, comp):
                    K.linear(dy, pwt, dpre, M=M, dact=pre, dact_kind=ctx.act)      # dgrad x act'(pre) in one pass
                else:
                    K.linear(dy, pwt, dpre, M=M)
                    if ctx.act != L.ACT_NONE:
                        L.check(L.lib().tante_act_bwd(dpre.data_ptr(), _DT[dpre.dtype], pre.data_ptr(), _DT[pre.dtype], dpre.data_ptr(),
                                                      _DT[dpre.dtype], pre.numel(), ctx.act, _s()), "act_bwd")
            else:
                acc = None
                for c0 in range(0, N, 512):
                    ck = min(512, N - c0)
                    pwt = _packed(W[c0:c0 + ck], None, comp, L.W_LINEAR_T, N=Kk, K=ck)
                    nxt = torch.empty(M, Kk, dtype=torch.float32, device=a.device)
                    K.linear(dy, pw=pwt, out=nxt, M=M, a_n0=M, a_s0=N, a_off=c0, residual=acc)
                    acc = nxt
                L.check(L.lib().tante_act_bwd(acc.data_ptr(), L.F32, pre.data_ptr(), _DT[pre.dtype], dpre.data_ptr(), _DT[dpre.dtype], pre.numel(),
                                              ctx.act, _s()), "act_bwd")
        if side_done:
            pass
        elif ctx.needs_input_grad[1]:
            gW, gb = _grad_slot(ctx.params[0]), _grad_slot(ctx.params[1])
            want_b = ctx.has_bias and ctx.needs_input_grad[2]
            if gW is not None and want_b and gb is not None:
                wgrad(_rm_linear(dy), _rm_linear(a), M, N, Kk, (N, Kk), comp, device=a.device, with_bias=True, into=gW, db_into=gb)
            elif bias32:      # (db32 None: the bias gradient went into its slot above)
                dW, db = wgrad(_rm_linear(dy), _rm_linear(a), M, N, Kk, (N, Kk), comp, device=a.device), db32
            else:
                dW, db = wgrad(_rm_linear(dy), _rm_linear(a), M, N, Kk, (N, Kk), comp, device=a.device, with_bias=True)
                if not want_b:
                    db = None
        elif ctx.has_bias and ctx.needs_input_grad[2]:
            db = db32 if bias32 else colsum(dy, M, N, 1)
        return dpre, dW, db, dout, None, None, None, None


class BlockTailFn(Function):
    """out = x1 + drop(fc2(gelu(fc1'(LN2(x1))))), x1 = xs + drop(out_proj(o)) -- the tail of a TransformerBlock behind its attention --
    as ONE node.  Forward is always precomputed (tante_block_fused_train produced `out` and every saved tensor); backward is one launch
    (tante_block_tail_bwd) that returns the gradients of o and of the skip operand xs and hands the three weight-gradient operand
    pairs to the deferred shared launches.  w1f / b1f are the LayerNorm-folded fc1 parameters (FoldFn outputs with their accumulators)."""

    @staticmethod
    def forward(ctx, o, xs, Wo, bo, w1f, b1f, W2, b2, saved, bwd_stream, p, seeds):
        ctx.save_for_backward(o, saved["hpre"], saved["xh2"], saved["st2"], saved["act"], bwd_stream)
        ctx.params = (Wo, bo, w1f, b1f, W2, b2)
        ctx.p, ctx.seed_out, ctx.seed_mlp = float(p), seeds[1], seeds[2]
        return saved["out"]

    @staticmethod
    def backward(ctx, dout):
        o, hpre, xh2, st2, act, bwd_stream = ctx.saved_tensors
        dout = dout.contiguous()
        if dout.dtype != torch.float32:
            dout = dout.float()
        M, Cc = o.shape
        t = K.block_tail_bwd(dout, hpre, xh2, st2, bwd_stream, Cc, hpre.shape[1], ctx.p, ctx.seed_out, ctx.seed_mlp)
        Wo, bo, w1f, b1f, W2, b2 = ctx.params
        for W, b, dy, a in ((W2, b2, t["dy2"], act), (w1f, b1f, t["dhpre"], xh2), (Wo, bo, t["dy1"], o)):
            gW, gb = _grad_slot(W), _grad_slot(b)
            N, Kk = W.shape
            if not _defer_wgrad(gW, gb, dy, a, M, N, Kk, L.BF16):
                wgrad(_rm_linear(dy), _rm_linear(a), M, N, Kk, (N, Kk), L.BF16, device=a.device, with_bias=True, into=gW, db_into=gb)
        return t["do"], t["dx1"], None, None, None, None, None, None, None, None, None, None


class BlockFn(Function):
    """A whole TransformerBlock as ONE autograd node (training, bf16, fused kernels): forward is tante_block_fused_train's result, backward
    is tante_block_tail_bwd followed by the three operators in front of it (attention backward, the q | k | v data-gradient GEMM with its
    deferred weight gradient, LayerNorm1 backward with the skip gradient added in the same pass) -- the SAME code the per-operator nodes
    run (their static backward methods, called on plain records), minus six trips through the autograd engine per block call: with one
    launch per block the host, not the GPU, was setting the train step's time."""

    @staticmethod
    def forward(ctx, x, w_in, b_in, Wo, bo, w1f, b1f, W2, b2, saved, bwd_stream, seq, n_head, causal, p, seeds, compute, head_stream=None,
                fwd_stream=None):
        ctx.save_for_backward(x, saved["st1"], saved["xh1"], saved["qkv"], saved["o"], saved["hpre"], saved["xh2"], saved["st2"], saved["act"],
                              bwd_stream, w_in)
        ctx.head_stream = head_stream      # fragments of the folded in-projection weight's transpose (tante_block_head_bwd), or None
        # the forward kernel's own weight stream: given = the backward is ONE launch (tante_block_bwd_fused) that recomputes q | k | v with
        # it (saved["qkv"] is None then: the forward did not store the packed projection)
        ctx.fwd_stream = fwd_stream
        ctx.params = (w_in, b_in, Wo, bo, w1f, b1f, W2, b2)
        ctx.meta = (seq, n_head, bool(causal), float(p), tuple(seeds), compute)
        return saved["out"]

    @staticmethod
    def backward(ctx, dout):
        from types import SimpleNamespace as NS
        x, st1, xh1, qkv, o, hpre, xh2, st2, act, bwd_stream, w_in_s = ctx.saved_tensors
        w_in, b_in, Wo, bo, w1f, b1f, W2, b2 = ctx.params
        seq, n_head, causal, p, seeds, compute = ctx.meta
        if ctx.fwd_stream is not None:
            dout = dout.contiguous()
            if dout.dtype != torch.float32:
                dout = dout.float()
            M, Cc = xh1.shape
            t = K.block_bwd_fused(dout, xh1, st1, hpre, xh2, st2, bwd_stream, ctx.fwd_stream, ctx.head_stream, Cc, n_head, hpre.shape[1], seq,
                                  causal, p, seeds)
            for W, b, dy, a in ((W2, b2, t["dy2"], act), (w1f, b1f, t["dhpre"], xh2), (Wo, bo, t["dy1"], o), (w_in, b_in, t["dqkv"], xh1)):
                gW, gb = _grad_slot(W), _grad_slot(b)
                N, Kk = W.shape
                if not _defer_wgrad(gW, gb, dy, a, M, N, Kk, L.BF16):
                    wgrad(_rm_linear(dy), _rm_linear(a), M, N, Kk, (N, Kk), L.BF16, device=a.device, with_bias=True, into=gW, db_into=gb)
            return (t["dx"],) + (None,) * 18
        tail = NS(saved_tensors=(o, hpre, xh2, st2, act, bwd_stream), params=(Wo, bo, w1f, b1f, W2, b2), p=p, seed_out=seeds[1], seed_mlp=seeds[2])
        r = BlockTailFn.backward(tail, dout)
        d_o, dx1 = r[0], r[1]
        dqkv = AttentionFn.backward(NS(saved_tensors=(qkv,), seq=seq, C=x.shape[1], nh=n_head, causal=causal, p=p, seed=seeds[0]), d_o)[0]
        if ctx.head_stream is not None and dqkv.dtype == torch.bfloat16:
            # q | k | v data gradient + LayerNorm1 backward + skip gradient: ONE launch (two GEMM launches and a LayerNorm backward otherwise)
            M, N = dqkv.shape
            gW, gb = _grad_slot(w_in), _grad_slot(b_in)
            if not _defer_wgrad(gW, gb, dqkv, xh1, M, N, xh1.shape[1], L.BF16):
                wgrad(_rm_linear(dqkv), _rm_linear(xh1), M, N, xh1.shape[1], (N, xh1.shape[1]), L.BF16, device=xh1.device, with_bias=True,
                      into=gW, db_into=gb)
            dx = K.block_head_bwd(dqkv, xh1, st1, dx1.contiguous(), ctx.head_stream, x.shape[1])
            return (dx,) + (None,) * 18
        dxh = LinearFn.backward(NS(saved_tensors=(xh1, w_in_s), compute=compute, has_bias=True, has_res=False, params=(w_in, b_in),
                                   needs_input_grad=(True, True, True, False, False, False, False)), dqkv)[0]
        dx = LayerNormSkipFn.backward(NS(saved_tensors=(x, st1)), dxh, dx1)[0]
        return (dx,) + (None,) * 18


def block_tail_ready(*params) -> bool:
    """BlockTailFn adds its weight gradients straight into the parameters' accumulators: every one must have one."""
    return all(_grad_slot(q) is not None for q in params)


class ActFn(Function):
    @staticmethod
    def forward(ctx, pre, act, out_dtype):
        post = torch.empty(pre.shape, dtype=out_dtype, device=pre.device)
        L.check(L.lib().tante_act_fwd(pre.data_ptr(), _DT[pre.dtype], post.data_ptr(), _DT[out_dtype], pre.numel(), act, _s()), "act_fwd")
        ctx.save_for_backward(pre)
        ctx.act = act
        return post

    @staticmethod
    def backward(ctx, g):
        (pre,) = ctx.saved_tensors
        g = g.contiguous()
        d = torch.empty_like(pre)
        L.check(L.lib().tante_act_bwd(g.data_ptr(), _DT[g.dtype], pre.data_ptr(), _DT[pre.dtype], d.data_ptr(), _DT[d.dtype], pre.numel(),
                                      ctx.act, _s()), "act_bwd")
        return d, None, None


_SEED = [0]


def next_seed() -> int:
    """A fresh 64-bit dropout seed per op, derived from torch's seed so that runs are reproducible."""
    _SEED[0] += 1
    # the data-parallel rank is mixed in: ranks seed identically (same initial weights), but their dropout masks must differ
    rank = int(__import__("os").environ.get("RANK", "0"))
    return (torch.initial_seed() * 0x9E3779B97F4A7C15 + _SEED[0] * 0xD1B54A32D192ED03 + rank * 0xA24BAED4963EE407) & 0xFFFFFFFFFFFFFFFF


class AttentionFn(Function):
    """o = dropout_p(softmax(q k^T / sqrt(d) [+causal])) v per (sequence, head) on the packed (tokens, 3C) projection."""

    @staticmethod
    def forward(ctx, qkv, seq, Cc, n_head, causal, p_drop=0.0, done=None):
        if done is not None:                  # (o, dropout seed) from the fused training-forward kernel
            o, seed = done
            ctx.save_for_backward(qkv)
            ctx.seq, ctx.C, ctx.nh, ctx.causal, ctx.p, ctx.seed = seq, Cc, n_head, causal, float(p_drop), seed
            return o
        o = torch.empty(qkv.shape[0], Cc, dtype=qkv.dtype, device=qkv.device)
        seed = next_seed() if p_drop > 0 else 0
        if p_drop > 0:
            L.check(L.lib().tante_attention_dropout(qkv.data_ptr(), o.data_ptr(), _DT[qkv.dtype], Cc, n_head, C.byref(seq), int(causal),
                                                    float(p_drop), seed, _s()), "attention_dropout")
        else:
            K.attention(qkv, o, Cc, n_head, seq, causal)
        ctx.save_for_backward(qkv)
        ctx.seq, ctx.C, ctx.nh, ctx.causal, ctx.p, ctx.seed = seq, Cc, n_head, causal, float(p_drop), seed
        return o

    @staticmethod
    def backward(ctx, do):
        (qkv,) = ctx.saved_tensors
        do = do.contiguous().to(qkv.dtype)
        dqkv = torch.empty_like(qkv)
        q = ctx.seq
        if q.L > 128 and ctx.p == 0.0 and q.n_s0 == 1 and q.S1 == q.L and q.n_l0 >= q.L and q.P0 == 1:
            # dense sequences past the MFMA backward's length (the channel letter 'C' over 256 channels, 'L' over a 16 x 16 patch grid):
            # the lane-per-row recomputing backward (tante_attention_masked_bwd with no masks)
            stats = torch.empty(q.nseq * ctx.nh * q.L * 3, dtype=torch.float32, device=qkv.device)
            L.check(L.lib().tante_attention_masked_bwd(qkv.data_ptr(), do.data_ptr(), dqkv.data_ptr(), _DT[qkv.dtype], ctx.C, ctx.nh, q.nseq, q.L,
                                                       int(ctx.causal), None, 0, None, stats.data_ptr(), _s()), "attention_masked_bwd")
            return dqkv, None, None, None, None, None, None
        L.check(L.lib().tante_attention_bwd(qkv.data_ptr(), do.data_ptr(), dqkv.data_ptr(), _DT[qkv.dtype], ctx.C, ctx.nh, C.byref(ctx.seq),
                                            int(ctx.causal), ctx.p, ctx.seed, _s()), "attention_bwd")
        return dqkv, None, None, None, None, None, None


class MaskedAttentionFn(Function):
    """AttentionFn over dense (Bp, L) sequences with nn.MultiheadAttention's masks as additive fp32 tensors (attn_backbone.py:59-72):
    attn_mask (1, L, L) or (Bp * n_head, L, L), key_padding_mask (Bp, L), either None.  No attention dropout on this path."""

    @staticmethod
    def forward(ctx, qkv, Cc, n_head, Bp, Lq, causal, attn_mask, key_padding_mask):
        o = torch.empty(qkv.shape[0], Cc, dtype=qkv.dtype, device=qkv.device)
        K.attention_masked(qkv, o, Cc, n_head, Bp, Lq, causal, attn_mask, key_padding_mask)
        ctx.save_for_backward(qkv)
        ctx.a = (Cc, n_head, Bp, Lq, int(causal))
        ctx.masks = (attn_mask, key_padding_mask)      # constants (no gradient): kept alive for the backward launch
        return o

    @staticmethod
    def backward(ctx, do):
        (qkv,) = ctx.saved_tensors
        Cc, nh, Bp, Lq, causal = ctx.a
        am, kp = ctx.masks
        do = do.contiguous().to(qkv.dtype)
        dqkv = torch.empty_like(qkv)
        stats = torch.empty(Bp * nh * Lq * 3, dtype=torch.float32, device=qkv.device)
        stride = 0 if am is None or am.shape[0] == 1 else Lq * Lq
        L.check(L.lib().tante_attention_masked_bwd(qkv.data_ptr(), do.data_ptr(), dqkv.data_ptr(), _DT[qkv.dtype], Cc, nh, Bp, Lq, causal,
                                                   am.data_ptr() if am is not None else None, stride,
                                                   kp.data_ptr() if kp is not None else None, stats.data_ptr(), _s()), "attention_masked_bwd")
        return dqkv, None, None, None, None, None, None, None


class DropoutAddFn(Function):
    """out = res + dropout_p(y)   (x + self.drop(y), attn_backbone.py:81-82)."""

    @staticmethod
    def forward(ctx, y, res, p):
        seed = next_seed()
        out = torch.empty_like(res)
        L.check(L.lib().tante_dropout_add(y.data_ptr(), _DT[y.dtype], res.data_ptr(), float(p), seed, y.numel(), out.data_ptr(), _s()), "dropout_add")
        ctx.p, ctx.seed, ctx.ydt = float(p), seed, y.dtype
        return out

    @staticmethod
    def backward(ctx, dout):
        dout = dout.contiguous()
        dy = torch.empty(dout.shape, dtype=ctx.ydt, device=dout.device)
        L.check(L.lib().tante_dropout_bwd(dout.data_ptr(), ctx.p, ctx.seed, dout.numel(), dy.data_ptr(), _DT[ctx.ydt], _s()), "dropout_bwd")
        return dy, dout, None


AXIS_BWD_FUSED = _O.register("TANTE_AXIS_BWD_FUSED", True, __name__, "AXIS_BWD_FUSED")     # propagator backward + weight gradients in one MFMA launch


class AxisMlpFn(Function):
    """y = x + W2 gelu_erf(W1 x + b1) + b2 along one axis of (outer, n, inner)."""

    @staticmethod
    def forward(ctx, x, w1, b1, w2, b2, outer, n, inner, compute):
        if n <= 8 and inner % 4 == 0 and x.dtype == torch.float32 and x.is_contiguous():
            # short axes (the temporal propagator): out of place in one launch instead of a 25 MB clone + the in-place kernel
            y = torch.empty_like(x)
            L.check(L.lib().tante_axis_mlp_oop(x.data_ptr(), y.data_ptr(), outer, n, inner, w1.data_ptr(), b1.data_ptr(), w2.data_ptr(),
                                               b2.data_ptr(), L.F32, _s()), "tante_axis_mlp_oop")
        else:
            y = x.clone()
            K.axis_mlp(y, outer, n, inner, w1, b1, w2, b2)
        ctx.save_for_backward(x, w1, b1, w2)
        ctx.dims, ctx.compute = (outer, n, inner), compute
        ctx.params = (w1, b1, w2, b2)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, w1, b1, w2 = ctx.saved_tensors
        outer, n, inner = ctx.dims
        dy = dy.contiguous()
        if (AXIS_BWD_FUSED and (ctx.compute == L.BF16 or n == 4) and x.dtype == torch.float32 and dy.dtype == torch.float32 and x.is_contiguous()
                and L.lib().tante_axis_mlp_bwd_fused_supported(n, inner)):
            # dx AND the four parameter gradients in one launch (x and dy read once, nothing materialised for separate weight-gradient
            # launches): axis lengths 16 / 32 / 48 on the matrix cores with bf16 operands (bf16 train path only), 4 on the vector
            # units in fp32 (the expressions of the kernels it replaces: every compute mode)
            slots = [_grad_slot(q) for q in ctx.params]
            direct = all(g is not None for g in slots)
            if direct:
                dw1, db1, dw2, db2 = slots
            else:
                dw1, dw2 = (torch.zeros(n, n, dtype=torch.float32, device=x.device) for _ in range(2))
                db1, db2 = (torch.zeros(n, dtype=torch.float32, device=x.device) for _ in range(2))
            dx = torch.empty_like(x)
            ws = _axis_wgrad_workspace(x.device)
            L.check(L.lib().tante_axis_mlp_bwd_fused_ws(x.data_ptr(), dy.data_ptr(), outer, n, inner, w1.data_ptr(), b1.data_ptr(), w2.data_ptr(),
                                                        dx.data_ptr(), dw1.data_ptr(), db1.data_ptr(), dw2.data_ptr(), db2.data_ptr(),
                                                        ws.data_ptr(), ws.numel() * 4, _s()), "tante_axis_mlp_bwd_fused")
            if direct:
                return dx, None, None, None, None, None, None, None, None
            return dx, dw1, db1, dw2, db2, None, None, None, None
        dx, h, dpre = torch.empty_like(x), torch.empty_like(x), torch.empty_like(x)
        L.check(L.lib().tante_axis_mlp_bwd(x.data_ptr(), dy.data_ptr(), outer, n, inner, w1.data_ptr(), b1.data_ptr(), w2.data_ptr(),
                                           dx.data_ptr(), h.data_ptr(), dpre.data_ptr(), _s()), "axis_mlp_bwd")
        R = outer * inner

        def lines(t):     # one row per (outer, inner) column, n elements with stride `inner`
            return _rm_linear(t, n0=inner, s1=n * inner, s0=1, es=inner, cols=n)
        comp = L.F32      # the propagators run in fp32 on the residual stream
        # the bias gradients are column sums of the wgrad kernels' first operand: they fall out of the tiles those already stage
        # (two stand-alone colsum passes over the residual stream cost 1.5 ms of a 41 ms step)
        slots = [_grad_slot(q) for q in ctx.params]
        if n <= 64 and inner % 16 == 0:
            # the contraction index (inner) is contiguous in this layout: a dedicated fp32-MFMA kernel streams both operands
            def aw(U, V, gw, gb):
                acc = gw is not None
                dW = gw if acc else torch.empty(n, n, dtype=torch.float32, device=x.device)
                dB = gb if acc else torch.empty(n, dtype=torch.float32, device=x.device)
                ws = _axis_wgrad_workspace(x.device)
                L.check(L.lib().tante_axis_wgrad_ws(U.data_ptr(), V.data_ptr(), outer, n, inner, dW.data_ptr(), dB.data_ptr(), int(acc),
                                                    ws.data_ptr(), ws.numel() * 4, _s()), "tante_axis_wgrad")
                return (None, None) if acc else (dW, dB)
            both = all(g is not None for g in slots)
            if both:
                with _side_wgrad(dy, h, dpre, x):
                    aw(dy, h, slots[2], slots[3])
                    aw(dpre, x, slots[0], slots[1])
                return dx, None, None, None, None, None, None, None, None
            dw2, db2 = aw(dy, h, None, None)
            dw1, db1 = aw(dpre, x, None, None)
            return dx, dw1, db1, dw2, db2, None, None, None, None
        if all(g is not None for g in slots):
            wgrad(lines(dy), lines(h), R, n, n, (n, n), comp, device=x.device, with_bias=True, into=slots[2], db_into=slots[3])
            wgrad(lines(dpre), lines(x), R, n, n, (n, n), comp, device=x.device, with_bias=True, into=slots[0], db_into=slots[1])
            return dx, None, None, None, None, None, None, None, None
        dw2, db2 = wgrad(lines(dy), lines(h), R, n, n, (n, n), comp, device=x.device, with_bias=True)
        dw1, db1 = wgrad(lines(dpre), lines(x), R, n, n, (n, n), comp, device=x.device, with_bias=True)
        return dx, dw1, db1, dw2, db2, None, None, None, None


class AxisHWFn(Function):
    """The vertical and the horizontal propagator of a backbone as one node (bf16 training path): forward is ONE launch
    (tante_axis_hw_train: both axes on MFMA over an LDS-resident plane, out of place, the intermediate planes saved for the backward),
    backward is AxisMlpFn's backward twice -- W axis on the saved intermediate, then H axis on the input."""

    @staticmethod
    def forward(ctx, x, vw1, vb1, vw2, vb2, hw1, hb1, hw2, hb2, BT, H, W, C_, compute):
        y, xm = K.axis_hw_train(x.detach().contiguous(), BT, H, W, C_, (vw1, vb1, vw2, vb2), (hw1, hb1, hw2, hb2), compute)
        ctx.save_for_backward(x, xm, vw1, vb1, vw2, hw1, hb1, hw2)
        ctx.params = (vw1, vb1, vw2, vb2, hw1, hb1, hw2, hb2)
        ctx.dims = (BT, H, W, C_)
        ctx.compute = compute
        return y

    @staticmethod
    def backward(ctx, dy):
        from types import SimpleNamespace as NS
        x, xm, vw1, vb1, vw2, hw1, hb1, hw2 = ctx.saved_tensors
        BT, H, W, C_ = ctx.dims
        p = ctx.params
        rw = AxisMlpFn.backward(NS(saved_tensors=(xm, hw1, hb1, hw2), dims=(BT * H, W, C_), compute=ctx.compute, params=p[4:8]), dy)
        rh = AxisMlpFn.backward(NS(saved_tensors=(x, vw1, vb1, vw2), dims=(BT, H, W * C_), compute=ctx.compute, params=p[0:4]), rw[0])
        return (rh[0], rh[1], rh[2], rh[3], rh[4], rw[1], rw[2], rw[3], rw[4], None, None, None, None, None)


def _tr_shape(M: int, I: int, J: int) -> bool:
    """Shapes the LDS-DMA / transposed-read weight-gradient kernel takes (bf16 dense rows)."""
    return M % 32 == 0 and I % 128 == 0 and J % 128 == 0


def _to_bf16(t: torch.Tensor) -> torch.Tensor:
    out = torch.empty(t.shape, dtype=torch.bfloat16, device=t.device)
    L.check(L.lib().tante_act_fwd(t.data_ptr(), L.F32, out.data_ptr(), L.BF16, t.numel(), L.ACT_NONE, _s()), "tante_act_fwd")
    return out


class PatchEmbedFn(Function):
    """Kernel = stride = P convolution as a patch GEMM; x (n_img, Cin, H, W) [nchw] or (n_img, H, W, Cin); out (rows, Cout) channels-last,
    PRE-activation."""

    @staticmethod
    def forward(ctx, x, W, b, n_img, Hin, Win, Cin, P, nchw, compute, out_dtype, act_in=L.ACT_NONE):
        """act_in != ACT_NONE: x is the PRE-activation of the previous stage and this node applies the activation itself (channels-last
        only) -- its backward then folds act'(x) into the data-gradient scatter instead of a separate pass over three images."""
        Cout = W.shape[0]
        if nchw:
            pw = _packed(W, b, compute, L.W_LINEAR, N=Cout, K=Cin * P * P)
        else:
            pw = _packed(W, b, compute, L.W_CONV_NHWC, N=Cout, K=Cin * P * P, P=P, C_other=Cin)
        M = n_img * (Hin // P) * (Win // P)
        pre = None
        if act_in != L.ACT_NONE:
            if nchw or Cin % 4 or not x.is_contiguous():
                raise RuntimeError("PatchEmbedFn: act_in needs a contiguous channels-last input with Cin % 4 == 0")
            pre, x = x, torch.empty_like(x)
            L.check(L.lib().tante_act_fwd(pre.data_ptr(), _DT[pre.dtype], x.data_ptr(), _DT[x.dtype], pre.numel(), act_in, _s()), "act_fwd")
        out = torch.empty(M, Cout, dtype=out_dtype, device=x.device)
        K.patch_embed(x, pw, out, n_img=n_img, Hin=Hin, Win=Win, Cin=Cin, P=P, nchw=nchw, act=L.ACT_NONE)
        ctx.save_for_backward(x, W, *([pre] if pre is not None else []))
        ctx.geo, ctx.compute = (n_img, Hin, Win, Cin, P, nchw), compute
        ctx.params = (W, b)
        ctx.act_in = act_in
        return out

    @staticmethod
    def backward(ctx, d):
        x, W = ctx.saved_tensors[:2]
        pre = ctx.saved_tensors[2] if ctx.act_in != L.ACT_NONE else None
        n_img, Hin, Win, Cin, P, nchw = ctx.geo
        comp = ctx.compute
        d = d.contiguous()
        M, Cout = d.shape
        Kk = Cin * P * P
        dx = dW = db = None
        if ctx.needs_input_grad[0]:     # col2im of non-overlapping patches = the pixel-shuffle scatter of a transposed conv
            if nchw:
                pwt = _packed(W, None, comp, L.W_LINEAR_T, N=Kk, K=Cout)
                dx = torch.empty(n_img, Cin, Hin, Win, dtype=torch.float32, device=x.device)
            else:
                pwt = _packed(W, None, comp, L.W_CONV_NHWC_T, N=Kk, K=Cout, P=P, C_other=Cin)
                dx = torch.empty(n_img, Hin, Win, Cin, dtype=x.dtype, device=x.device)
            K.deconv(d, pwt, dx, n_img=n_img, Hi=Hin // P, Wi=Win // P, P=P, Cout=Cin, nchw_out=nchw, act=L.ACT_NONE,
                     dact=pre, dact_kind=ctx.act_in)      # act_in: dx is the gradient of the previous stage's PRE-activation
            dx = dx.view(x.shape)
        if ctx.needs_input_grad[1]:
            U, V = _rm_linear(d), _rm_patch(x, not nchw, n_img, Hin, Win, Cin, P)
            keep = [d, x]        # operands the (side-stream) kernel reads: held here and marked in use on that stream
            if comp == L.BF16 and not nchw and x.dtype == torch.bfloat16 and _tr_shape(M, Cout, Kk):
                # a kernel = stride patch matrix is a permutation of the image: one 2 x 25 MB copy buys the dense-row weight-gradient
                # kernel (~45 us) instead of the gathering one (~240 us at 24576 x 256 x 512)
                cols = K.im2col(x, False, n_img, Cin, Hin, Win, P, P, P, P, 0, 0, 1, torch.bfloat16)
                d16 = d if d.dtype == torch.bfloat16 else _to_bf16(d)
                U, V = _rm_linear(d16), _rm_linear(cols)
                keep = [d16, cols]
            gW, gb = _grad_slot(ctx.params[0]), _grad_slot(ctx.params[1])
            lay = L.W_LINEAR if nchw else L.W_CONV_NHWC
            if gW is not None and gb is not None and ctx.needs_input_grad[2]:
                if len(keep) == 2 and keep[1] is not x and _defer_wgrad(gW, gb, keep[0], keep[1], M, Cout, Kk, comp, (lay, P, Cin, False)):
                    pass                       # dense (dY, patch rows): shares a launch with the other BPTT uses
                else:
                    with _side_wgrad(*keep):
                        wgrad(U, V, M, Cout, Kk, tuple(W.shape), comp, layout=lay, P=P, C_other=Cin, device=x.device, with_bias=True,
                              into=gW, db_into=gb)
            else:
                dW, db = wgrad(U, V, M, Cout, Kk, tuple(W.shape), comp, layout=lay, P=P, C_other=Cin, device=x.device,
                               with_bias=True)
        elif ctx.needs_input_grad[2]:
            db = colsum(d, M, Cout, 1)
        return dx, dW, db, None, None, None, None, None, None, None, None, None


class DeconvFn(Function):
    """Kernel = stride = P transposed convolution: a (rows, Cin) pixels -> (n_img, Hi*P, Wi*P, Cout) channels-last, or (n_img, Cout, .., ..)
    fp32 when nchw_out; PRE-activation."""

    @staticmethod
    def forward(ctx, a, W, b, n_img, Hi, Wi, P, nchw_out, compute, out_dtype):
        Cin, Cout = W.shape[0], W.shape[1]
        lay = L.W_DECONV_NCHW if nchw_out else L.W_DECONV_NHWC
        pw = _packed(W, b, compute, lay, N=Cout * P * P, K=Cin, P=P, C_other=Cout)
        if nchw_out:
            out = torch.empty(n_img, Cout, Hi * P, Wi * P, dtype=torch.float32, device=a.device)
        else:
            out = torch.empty(n_img, Hi * P, Wi * P, Cout, dtype=out_dtype, device=a.device)
        K.deconv(a, pw, out, n_img=n_img, Hi=Hi, Wi=Wi, P=P, Cout=Cout, nchw_out=nchw_out, act=L.ACT_NONE)
        ctx.save_for_backward(a, W)
        ctx.geo, ctx.compute = (n_img, Hi, Wi, P, nchw_out), compute
        ctx.params = (W, b)
        return out

    @staticmethod
    def backward(ctx, d):
        a, W = ctx.saved_tensors
        n_img, Hi, Wi, P, nchw_out = ctx.geo
        comp = ctx.compute
        Cin, Cout = W.shape[0], W.shape[1]
        d = d.contiguous()
        M, N = n_img * Hi * Wi, Cout * P * P
        da = dW = db = None
        # the P x P output-gradient patch of every input pixel, once, as dense bf16 rows: both the data-gradient GEMM and the
        # weight-gradient kernel then run their dense-row variants instead of gathering
        cols = None
        if comp == L.BF16 and not nchw_out and d.dtype == torch.bfloat16 and N <= 512 and _tr_shape(M, Cin, N) \
                and (ctx.needs_input_grad[0] or ctx.needs_input_grad[1]):
            cols = K.im2col(d, False, n_img, Cout, Hi * P, Wi * P, P, P, P, P, 0, 0, 1, torch.bfloat16)
        if ctx.needs_input_grad[0]:
            lay = L.W_DECONV_NCHW_T if nchw_out else L.W_DECONV_NHWC_T
            pwt = _packed(W, None, comp, lay, N=Cin, K=N, P=P, C_other=Cout)
            da = torch.empty(M, Cin, dtype=a.dtype, device=a.device)
            if cols is not None:
                K.linear(cols, pwt, da, M=M)
            else:
                K.patch_embed(d, pwt, da, n_img=n_img, Hin=Hi * P, Win=Wi * P, Cin=Cout, P=P, nchw=nchw_out, act=L.ACT_NONE)
            da = da.view(a.shape)
        if ctx.needs_input_grad[1]:
            V = _rm_linear(cols) if cols is not None else _rm_patch(d, not nchw_out, n_img, Hi * P, Wi * P, Cout, P)
            gW = _grad_slot(ctx.params[0])
            if cols is not None and a.dtype != torch.bfloat16:     # the first stage reads the fp32 residual stream
                a = _to_bf16(a)
            lay_w = L.W_DECONV_NCHW if nchw_out else L.W_DECONV_NHWC
            if gW is not None:
                if cols is not None and _defer_wgrad(gW, None, a, cols, M, Cin, N, comp, (lay_w, P, Cout, True)):
                    pass
                else:
                    with _side_wgrad(a, d if cols is None else cols):
                        wgrad(_rm_linear(a), V, M, Cin, N, tuple(W.shape), comp, layout=lay_w, P=P, C_other=Cout, swap=True, device=a.device,
                              into=gW)
            else:
                dW = wgrad(_rm_linear(a), V, M, Cin, N, tuple(W.shape), comp, layout=lay_w, P=P, C_other=Cout, swap=True, device=a.device)
        if ctx.needs_input_grad[2]:
            gb = _grad_slot(ctx.params[1])
            db = colsum(d, n_img, Cout, Hi * P * Wi * P, gb) if nchw_out else colsum(d, n_img * Hi * P * Wi * P, Cout, 1, gb)
            if gb is not None:
                db = None
        return da, dW, db, None, None, None, None, None, None, None


class FilmPosFn(Function):
    """y[r] = v[r] * a[t] + b[t] + s_emb[hw],  r = (b, t, hw)   (tante.py:136-141 with t_emb folded into b)."""

    @staticmethod
    def forward(ctx, v, a, b, s_emb, T, HW):
        rows, Cc = v.shape
        y = torch.empty_like(v)
        L.check(L.lib().tante_film_pos_fwd(v.data_ptr(), a.data_ptr(), b.data_ptr(), s_emb.data_ptr(), rows, Cc, T, HW, y.data_ptr(), _s()),
                "film_pos_fwd")
        ctx.save_for_backward(v, a)
        ctx.dims = (T, HW)
        return y

    @staticmethod
    def backward(ctx, dy):
        v, a = ctx.saved_tensors
        T, HW = ctx.dims
        rows, Cc = v.shape
        dy = dy.contiguous()
        dv = torch.empty_like(v)
        da = torch.empty(T, Cc, dtype=torch.float32, device=v.device)
        db = torch.empty(T, Cc, dtype=torch.float32, device=v.device)
        ds = torch.empty(HW, Cc, dtype=torch.float32, device=v.device)
        L.check(L.lib().tante_film_pos_bwd(dy.data_ptr(), v.data_ptr(), a.data_ptr(), rows // HW, HW, Cc, T, dv.data_ptr(), da.data_ptr(),
                                           db.data_ptr(), ds.data_ptr(), _s()), "film_pos_bwd")
        return dv, da, db, ds, None, None


class FilmPosFramesFn(Function):
    """FilmPosFn with the window given as T separate frame encodings (each (B, HW, C) fp32, rows contiguous, any batch stride): the BPTT
    rollout keeps one encoding per frame and a window is the last T of them -- no 25 MB torch.stack per call, and every frame receives its
    gradient as a contiguous tensor of its own (tante.py:136-141 over trainer.py:144-159's sliding window)."""

    @staticmethod
    def _arg(frames):
        fr = L.Frames()
        for t, f in enumerate(frames):
            fr.f[t] = f.data_ptr()
            fr.bstride[t] = f.stride(0)
        return fr

    @staticmethod
    def supported(frames, Cc) -> bool:
        return (Cc == 256 and 1 <= len(frames) <= 8 and all(f.dtype == torch.float32 and f.is_cuda and f.dim() == 3 and f.stride(2) == 1
                                                             and f.stride(1) == Cc and f.stride(0) % 4 == 0 and f.data_ptr() % 16 == 0
                                                             for f in frames))

    @staticmethod
    def forward(ctx, a, b, s_emb, *frames):
        T = len(frames)
        B, HW, Cc = frames[0].shape
        y = torch.empty(B * T * HW, Cc, dtype=torch.float32, device=a.device)
        fr = FilmPosFramesFn._arg(frames)
        L.check(L.lib().tante_film_pos_fwd_frames(C.byref(fr), a.data_ptr(), b.data_ptr(), s_emb.data_ptr(), B, T, HW, Cc, y.data_ptr(), _s()),
                "film_pos_fwd_frames")
        ctx.save_for_backward(a, *frames)
        # the Python objects (saved_tensors hands back fresh wrappers): their identity keys the per-backward-pass accumulators below
        ctx.frame_objs, ctx.a_obj, ctx.b_obj, ctx.s_obj = frames, a, b, s_emb
        return y

    @staticmethod
    def backward(ctx, dy):
        a, *frames = ctx.saved_tensors
        T = len(frames)
        B, HW, Cc = frames[0].shape
        dy = dy.contiguous()
        dev = a.device
        # Gradients that several calls of a BPTT rollout contribute to are accumulated IN PLACE by the kernel instead of returned as fresh
        # tensors for autograd to sum (one elementwise add + one temporary per use: 9 adds of 6 MB for the frame encodings, 6 for the
        # tables, 3 for s_emb per train step):
        #   * a frame encoding sits in up to T windows.  The FIRST of their nodes to run in a backward pass allocates the frame's
        #     accumulator, has the kernel WRITE it and RETURNS it: the engine keeps that tensor (by reference) as the pending gradient of
        #     the frame's producer.  Every later node of the same pass has the kernel ADD into the same storage and returns None.  The
        #     producer runs once all of the pass's consumers of the frame have run (the engine's own dependency count), so it sees the
        #     sum of exactly the windows that took part -- whichever they are: a loss on some calls only, torch.autograd.grad with an
        #     inputs subset, a second pass over a retained graph (the table is keyed to the graph task and dropped when it ends);
        #   * the FiLM tables come from ONE FilmTableFn node per rollout and carry accumulators (`_tante_grad`) that node reads;
        #   * s_emb is a parameter: its .grad slot is added to directly.
        objs = ctx.frame_objs
        if objs is None:
            raise RuntimeError("FilmPosFramesFn.backward: the node's frame records are gone (backward through a freed graph?)")
        acc = _frame_accumulators()
        mask, dvs, rets = 0, [], []
        for t, f in enumerate(objs):
            g = None if acc is None else acc.get(id(f))
            if g is None:                             # first contribution of this pass: write, and hand the tensor to the engine
                g = torch.empty(B, HW, Cc, dtype=torch.float32, device=dev)
                if acc is not None:
                    acc[id(f)] = g
                rets.append(g)
            else:                                     # the engine already holds g for the frame's producer: add in place
                mask |= 1 << t
                rets.append(None)
            dvs.append(g)
        ga, gb, gs = _grad_slot(ctx.a_obj), _grad_slot(ctx.b_obj), _grad_slot(ctx.s_obj)
        flags = 0
        if ga is not None and gb is not None:
            da, db, flags = ga, gb, 1
        else:
            da = torch.empty(T, Cc, dtype=torch.float32, device=dev)
            db = torch.empty(T, Cc, dtype=torch.float32, device=dev)
        if gs is not None:
            ds, flags = gs.view(HW, Cc), flags | 2
        else:
            ds = torch.empty(HW, Cc, dtype=torch.float32, device=dev)
        fr = FilmPosFramesFn._arg(frames)
        ptrs = (C.c_void_p * T)(*[d.data_ptr() for d in dvs])
        L.check(L.lib().tante_film_pos_bwd_frames_acc(dy.data_ptr(), C.byref(fr), a.data_ptr(), B, HW, Cc, T, ptrs, mask, da.data_ptr(),
                                                      db.data_ptr(), ds.data_ptr(), flags, _s()), "film_pos_bwd_frames_acc")
        return (None if flags & 1 else da, None if flags & 1 else db, None if flags & 2 else ds, *rets)


def guard_frame_gradient(frame: torch.Tensor):
    """FilmPosFramesFn's accumulators rely on the autograd engine keeping the FIRST node's returned tensor, by reference, as the pending
    gradient of the frame's producer (later nodes add into the same storage and return None).  Should the engine ever accumulate out of
    place -- a second consumer of the frame outside FilmPosFramesFn, a producer / consumer stream mismatch -- those in-place adds would be
    lost silently.  This hook on the frame checks what actually reaches the producer: the accumulator itself."""
    if not frame.requires_grad:
        return
    key = id(frame)

    def hook(g):
        acc = _frame_accumulators()
        exp = None if acc is None else acc.get(key)
        if exp is not None and g.data_ptr() != exp.data_ptr():
            raise RuntimeError("tante_amd: a frame encoding's gradient reached its producer as a different tensor than FilmPosFramesFn's "
                               "accumulator -- the windows' in-place contributions would be lost (the frame has another consumer, or the "
                               "engine accumulated out of place); set TANTE_TRAIN_FRAME_FILM=0")
    frame.register_hook(hook)


class SplitFramesFn(Function):
    """(B, k, HW, C) frame encodings -> k contiguous (B, HW, C) tensors.  torch's unbind would do, but its backward builds the window's
    gradient with a zero fill + a copy per frame and autograd then sums those k full-size tensors; here the forward is one transposing
    copy and the backward one stack of the k frame gradients (a frame no window used contributes zeros)."""

    @staticmethod
    def forward(ctx, z):
        B, k, HW, Cc = z.shape
        zc = z.permute(1, 0, 2, 3).contiguous()          # (k, B, HW, C): one copy kernel
        ctx.dims = (B, k, HW, Cc)
        return tuple(zc[f] for f in range(k))

    @staticmethod
    def backward(ctx, *gs):
        B, k, HW, Cc = ctx.dims
        ref = next(g for g in gs if g is not None)
        gs = [g if g is not None else torch.zeros_like(ref) for g in gs]
        return torch.stack(gs, dim=1)


class FilmTableFn(Function):
    """(a, b) = (1 + scale(t), shift(t) + add): the FiLM tables of a rollout (tante.py:203-230 with t_emb folded into the shift) in ONE
    launch, differentiable: the outputs carry zeroed accumulators (`_tante_grad`) that every FilmPos*Fn use adds its da / db into, autograd
    then calls this backward once with nothing, and ONE launch (tante_film_table_bwd) turns the accumulated (T, C) gradients into the eight
    MLP parameters' gradients, straight into their slots.  (As torch modules the two MLPs were ten hipBLASLt GEMMs plus their glue per
    train step -- 0.14 ms of 4-row launches -- and their tables' gradients six more adds.)"""

    @staticmethod
    def forward(ctx, t, add, *params):
        rows, Cc = t.numel(), params[3].numel()
        dev = t.device
        a = torch.empty(rows, Cc, dtype=torch.float32, device=dev)
        b = torch.empty(rows, Cc, dtype=torch.float32, device=dev)
        L.check(L.lib().tante_film_table(t.data_ptr(), rows, Cc, *[p.data_ptr() for p in params], None if add is None else add.data_ptr(),
                                         a.data_ptr(), b.data_ptr(), _s()), "tante_film_table")
        acc = torch.zeros(2, rows, Cc, dtype=torch.float32, device=dev)
        ctx.save_for_backward(t, *params)
        ctx.acc, ctx.params, ctx.add = acc, params, add
        ctx.set_materialize_grads(False)
        ctx.mark_non_differentiable(acc)
        return a, b, acc

    @staticmethod
    def backward(ctx, ga, gb, _acc):
        t, *params = ctx.saved_tensors
        rows, Cc = t.numel(), params[3].numel()
        dA, dB = ctx.acc[0], ctx.acc[1]
        if ga is not None:
            dA = dA + ga
        if gb is not None:
            dB = dB + gb
        dA, dB = dA.contiguous(), dB.contiguous()
        slots = [_grad_slot(p) for p in ctx.params]
        direct = all(g is not None for g in slots)
        outs = slots if direct else [torch.empty_like(p) for p in params]
        sc_w0, sc_b0, sc_w2, sc_b2, sh_w0, sh_b0, sh_w2, sh_b2 = params
        L.check(L.lib().tante_film_table_bwd(t.data_ptr(), rows, Cc, sc_w0.data_ptr(), sc_b0.data_ptr(), sc_w2.data_ptr(), sh_w0.data_ptr(),
                                             sh_b0.data_ptr(), sh_w2.data_ptr(), dA.data_ptr(), dB.data_ptr(), *[o.data_ptr() for o in outs],
                                             1 if direct else 0, _s()), "tante_film_table_bwd")
        d_add = None
        if ctx.add is not None and ctx.add.requires_grad:
            d_add = dB.view(ctx.add.shape).clone() if dB.numel() == ctx.add.numel() else None
        ctx.acc.zero_()      # consumed: a second pass over a retained graph starts from zeros again
        return (None, d_add) + ((None,) * 8 if direct else tuple(outs))


class TaylorFn(Function):
    """out_i = inp[:, -1] + sum_k derivs[k] (i dt)^k / k!   (tante.py:165-171)."""

    @staticmethod
    def forward(ctx, inp, dt, n_out, *derivs):
        B, T = inp.shape[:2]
        frame = inp[0, 0].numel()
        out = torch.empty(B, n_out, *inp.shape[2:], dtype=torch.float32, device=inp.device)
        K.taylor(inp, (T - 1) * frame, T * frame, [d.contiguous() for d in derivs], dt, n_out, out, B, frame)
        ctx.dims = (B, T, frame, dt, n_out, len(derivs), tuple(inp.shape))
        return out

    @staticmethod
    def backward(ctx, dout):
        B, T, frame, dt, n_out, n_order, ishape = ctx.dims
        dout = dout.contiguous()
        # (the kernel writes the last frame's gradient in full; only the earlier frames of a longer window need the zero fill)
        dinp = (torch.empty if T == 1 else torch.zeros)(ishape, dtype=torch.float32, device=dout.device)
        dds = [torch.empty(B, frame, dtype=torch.float32, device=dout.device) for _ in range(n_order)]
        arr = (C.c_void_p * n_order)(*[d.data_ptr() for d in dds])
        L.check(L.lib().tante_taylor_bwd(dout.data_ptr(), n_out * frame, arr, n_order, float(dt), n_out,
                                         dinp.data_ptr() + 4 * (T - 1) * frame, T * frame, 0, B, frame, _s()), "taylor_bwd")
        return (dinp, None, None) + tuple(d.view(B, 1, *ishape[2:]) for d in dds)


class TailFn(Function):
    """The token-local tail of a rollout call as ONE autograd node, one launch forward and one backward (csrc/tail_chain.hip):
    every order's derivative head (dec_CNN, enc_dec_cnn.py:263-277) -> Taylor sum (tante.py:165-171) -> predicted frame -> its
    re-encoding for the next call's window (enc_CNN, enc_dec_cnn.py:217-229).  Inputs: base (B, 1, D, H, W) = the window's last frame,
    cfg = TailCfg, xs[k] = order k's residual stream (B * T * HW, C) fp32 (its last time slot is read).  Outputs: the predicted frame
    (B, 1, D, H, W) and its pre-FiLM encoding (B, HW, C) (None when cfg.want_z is False: the rollout's last call).
    Parameter gradients go straight into the parameters' gradient slots (cfg checked that every one exists): the four wide weight
    gradients as recorded uses of the shared end-of-pass launches, the two pixel-level ones (64 x 4 D) and the decoder biases inside the
    backward launch (per-workgroup partials + a small reduce launch; PIXEL_WGRAD_IN_KERNEL off: from row operands, immediately)."""

    @staticmethod
    def forward(ctx, base, cfg, *xs):
        B, T, HW, D, Hp, Wp = cfg.B, cfg.T, cfg.HW, cfg.D, cfg.Hp, cfg.Wp
        dev = base.device
        Tk = B * HW
        bf = torch.bfloat16
        a = L.TailFwd()
        a.n_ord = len(xs)
        a.a_n0, a.a_s1, a.a_s0, a.a_off = HW, T * HW * cfg.C, cfg.C, (T - 1) * HW * cfg.C
        a.n_img, a.Hp, a.Wp, a.D = B, Hp, Wp, D
        base = base.contiguous()
        out = torch.empty(B, 1, D, 8 * Hp, 8 * Wp, dtype=torch.float32, device=dev)
        a.base, a.base_bstride, a.out, a.out_bstride = base.data_ptr(), base.stride(0), out.data_ptr(), out.stride(0)
        saved = []
        for k, x in enumerate(xs):
            x = x.contiguous()      # (kernels on one stream: the caching allocator does not hand a temporary's memory out before they have run)
            o = a.o[k]
            t = {"xl16": torch.empty(Tk, 256, dtype=bf, device=dev), "pre1": torch.empty(4 * Tk, 128, dtype=bf, device=dev),
                 "act1": torch.empty(4 * Tk, 128, dtype=bf, device=dev), "pre2": torch.empty(16 * Tk, 64, dtype=bf, device=dev),
                 "act2": torch.empty(16 * Tk, 64, dtype=bf, device=dev)}
            o.x, o.w, o.coef = x.data_ptr(), cfg.dec_streams[k][0].data_ptr(), cfg.coefs[k]
            o.xl16, o.pre1, o.act1, o.pre2, o.act2 = (t[n].data_ptr() for n in ("xl16", "pre1", "act1", "pre2", "act2"))
            saved.append(t)
        enc = None
        z = None
        if cfg.want_z:
            enc = {"f16": torch.empty(16 * Tk, 64, dtype=bf, device=dev), "pre1e": torch.empty(16 * Tk, 64, dtype=bf, device=dev),
                   "act1e": torch.empty(16 * Tk, 64, dtype=bf, device=dev), "pre2e": torch.empty(4 * Tk, 128, dtype=bf, device=dev),
                   "act2e": torch.empty(4 * Tk, 128, dtype=bf, device=dev)}
            z = torch.empty(B, HW, cfg.C, dtype=torch.float32, device=dev)
            a.we = cfg.enc_streams[0].data_ptr()
            a.f16, a.pre1e, a.act1e, a.pre2e, a.act2e = (enc[n].data_ptr() for n in ("f16", "pre1e", "act1e", "pre2e", "act2e"))
            a.z = z.data_ptr()
        L.check(L.lib().tante_tail_fwd(C.byref(a), _s()), "tante_tail_fwd")
        ctx.cfg, ctx.saved, ctx.enc, ctx.xshapes = cfg, saved, enc, [tuple(x.shape) for x in xs]
        ctx.set_materialize_grads(False)
        return out, z

    @staticmethod
    def backward(ctx, d_out, d_z):
        cfg, saved, enc = ctx.cfg, ctx.saved, ctx.enc
        B, T, HW, D, Hp, Wp = cfg.B, cfg.T, cfg.HW, cfg.D, cfg.Hp, cfg.Wp
        Tk = B * HW
        n_ord = len(saved)
        if d_out is None and d_z is None:
            return (None, None) + (None,) * n_ord
        dev = saved[0]["xl16"].device
        bf = torch.bfloat16
        a = L.TailBwd()
        a.n_ord = n_ord
        a.a_n0, a.a_s1, a.a_s0, a.a_off = HW, T * HW * cfg.C, cfg.C, (T - 1) * HW * cfg.C
        a.n_img, a.Hp, a.Wp, a.D = B, Hp, Wp, D
        keep = []
        if d_out is not None:
            d_out = d_out.contiguous()
            keep.append(d_out)
            a.dext, a.dext_bstride = d_out.data_ptr(), d_out.stride(0)
        dbase = torch.empty(B, 1, D, 8 * Hp, 8 * Wp, dtype=torch.float32, device=dev)
        a.dbase, a.dbase_bstride = dbase.data_ptr(), dbase.stride(0)
        outs, dxs = [], []
        for k in range(n_ord):
            t = saved[k]
            o = a.o[k]
            # (the other time slots of the stream receive no gradient from this node: the zero fill the slice's backward used to do)
            dx = torch.zeros(ctx.xshapes[k], dtype=torch.float32, device=dev)
            g = {"dpre1": torch.empty(Tk, 512, dtype=bf, device=dev), "dpre2": torch.empty(4 * Tk, 256, dtype=bf, device=dev)}
            dc = cfg.dec_params[k]
            slots = [_grad_slot(q) for q in dc]
            o.w, o.coef, o.pre1, o.pre2 = cfg.dec_streams[k][1].data_ptr(), cfg.coefs[k], t["pre1"].data_ptr(), t["pre2"].data_ptr()
            o.dpre1, o.dpre2, o.dx = g["dpre1"].data_ptr(), g["dpre2"].data_ptr(), dx.data_ptr()
            o.db1, o.db2, o.db3 = slots[1].data_ptr(), slots[3].data_ptr(), slots[5].data_ptr()
            if PIXEL_WGRAD_IN_KERNEL:      # the last stage's weight gradient inside the launch (per-workgroup partials + the reduce launch)
                o.act2, o.dw3 = t["act2"].data_ptr(), slots[4].data_ptr()
            else:
                g["dder"] = torch.empty(16 * Tk, 64, dtype=bf, device=dev)
                o.dder = g["dder"].data_ptr()
            outs.append((g, slots))
            dxs.append(dx)
        ge = None
        if d_z is not None and enc is not None:
            d_z = d_z.contiguous()
            keep.append(d_z)
            ge = {"dz16": torch.empty(Tk, 256, dtype=bf, device=dev), "dpre2e": torch.empty(4 * Tk, 128, dtype=bf, device=dev)}
            a.dz, a.we = d_z.data_ptr(), cfg.enc_streams[1].data_ptr()
            a.pre1e, a.pre2e = enc["pre1e"].data_ptr(), enc["pre2e"].data_ptr()
            a.dz16, a.dpre2e = ge["dz16"].data_ptr(), ge["dpre2e"].data_ptr()
            _tail_enc_first_stage(a, cfg, enc, ge, Tk, dev)
        bias_ws = torch.empty((Tk // 16) * (n_ord + 1) * (L.lib().tante_tail_stream_bytes(4) // 4), dtype=torch.float32, device=dev)
        a.bias_ws = bias_ws.data_ptr()
        L.check(L.lib().tante_tail_bwd(C.byref(a), _s()), "tante_tail_bwd")
        comp = L.BF16
        for k in range(n_ord):
            t, (g, slots), dc = saved[k], outs[k], cfg.dec_params[k]
            # decoder weights (Cin, Cout, 2, 2): U = the stage's input pixels, V = the output-gradient patches, n = (kh, kw, co)
            if not _defer_wgrad(slots[0], None, t["xl16"], g["dpre1"], Tk, 256, 512, comp, (L.W_DECONV_NHWC, 2, 128, True)):
                wgrad(_rm_linear(t["xl16"]), _rm_linear(g["dpre1"]), Tk, 256, 512, tuple(dc[0].shape), comp, layout=L.W_DECONV_NHWC, P=2,
                      C_other=128, swap=True, device=dev, into=slots[0])
            if not _defer_wgrad(slots[2], None, t["act1"], g["dpre2"], 4 * Tk, 128, 256, comp, (L.W_DECONV_NHWC, 2, 64, True)):
                wgrad(_rm_linear(t["act1"]), _rm_linear(g["dpre2"]), 4 * Tk, 128, 256, tuple(dc[2].shape), comp, layout=L.W_DECONV_NHWC, P=2,
                      C_other=64, swap=True, device=dev, into=slots[2])
            if "dder" in g:
                wgrad(_rm_linear(t["act2"]), _rm_linear(g["dder"], s0=64, rows=16 * Tk, cols=4 * D), 16 * Tk, 64, 4 * D, tuple(dc[4].shape), comp,
                      layout=L.W_DECONV_NHWC, P=2, C_other=D, swap=True, device=dev, into=slots[4])
        if ge is not None:
            _tail_enc_wgrads(cfg, enc, ge, Tk, dev)
        ctx.saved = ctx.enc = None
        return (dbase, None) + tuple(dxs)


class EncTailFn(Function):
    """enc_CNN.forward on INPUT frames (the rollout's initial window) through the tail kernels' encoder half (csrc/tail_chain.hip with no
    decoder): frames (n_img, D, H, W) fp32, no gradient -> their pre-FiLM encodings (n_img * HW, C) fp32.  One launch forward (with the
    activations the weight gradients read, token-major), one backward (the two wide stages backwards: the weight-gradient operands)."""

    @staticmethod
    def forward(ctx, frames, cfg, *enc_params):      # (the parameters are inputs so that the node takes part in the backward pass)
        n_img, D, Hp, Wp, HW = frames.shape[0], cfg.D, cfg.Hp, cfg.Wp, cfg.HW
        dev, bf = frames.device, torch.bfloat16
        Tk = n_img * HW
        frames = frames.contiguous()
        a = L.TailFwd()
        a.n_ord, a.a_n0 = 0, HW
        a.n_img, a.Hp, a.Wp, a.D = n_img, Hp, Wp, D
        a.base, a.base_bstride = frames.data_ptr(), frames.stride(0)
        enc = {"f16": torch.empty(16 * Tk, 64, dtype=bf, device=dev), "pre1e": torch.empty(16 * Tk, 64, dtype=bf, device=dev),
               "act1e": torch.empty(16 * Tk, 64, dtype=bf, device=dev), "pre2e": torch.empty(4 * Tk, 128, dtype=bf, device=dev),
               "act2e": torch.empty(4 * Tk, 128, dtype=bf, device=dev)}
        z = torch.empty(Tk, cfg.C, dtype=torch.float32, device=dev)
        a.we = cfg.enc_streams[0].data_ptr()
        a.f16, a.pre1e, a.act1e, a.pre2e, a.act2e = (enc[n].data_ptr() for n in ("f16", "pre1e", "act1e", "pre2e", "act2e"))
        a.z = z.data_ptr()
        L.check(L.lib().tante_tail_fwd(C.byref(a), _s()), "tante_tail_fwd")
        ctx.cfg, ctx.enc, ctx.geo = cfg, enc, (n_img, Tk)
        return z

    @staticmethod
    def backward(ctx, d_z):
        cfg, enc = ctx.cfg, ctx.enc
        n_img, Tk = ctx.geo
        D = cfg.D
        dev, bf = d_z.device, torch.bfloat16
        d_z = d_z.contiguous()
        a = L.TailBwd()
        a.n_ord, a.a_n0 = 0, cfg.HW
        a.n_img, a.Hp, a.Wp, a.D = n_img, cfg.Hp, cfg.Wp, D
        ge = {"dz16": torch.empty(Tk, 256, dtype=bf, device=dev), "dpre2e": torch.empty(4 * Tk, 128, dtype=bf, device=dev)}
        a.dz, a.we = d_z.data_ptr(), cfg.enc_streams[1].data_ptr()
        a.pre1e, a.pre2e = enc["pre1e"].data_ptr(), enc["pre2e"].data_ptr()
        a.dz16, a.dpre2e = ge["dz16"].data_ptr(), ge["dpre2e"].data_ptr()
        _tail_enc_first_stage(a, cfg, enc, ge, Tk, dev)
        if a.dwe1:
            bias_ws = torch.empty((Tk // 16) * (L.lib().tante_tail_stream_bytes(4) // 4), dtype=torch.float32, device=dev)
            a.bias_ws = bias_ws.data_ptr()
        L.check(L.lib().tante_tail_bwd(C.byref(a), _s()), "tante_tail_bwd")
        _tail_enc_wgrads(cfg, enc, ge, Tk, dev)
        ctx.enc = None
        return (None, None) + (None,) * len(cfg.enc_params)


# the two pixel-level weight gradients (64 x 4 D each) inside the tail's backward launch instead of a gathering launch + a reduce each
PIXEL_WGRAD_IN_KERNEL = _O.register("TANTE_TAIL_PIXEL_WGRAD", True, __name__, "PIXEL_WGRAD_IN_KERNEL")


def _tail_enc_first_stage(a, cfg, enc, ge, Tk, dev):
    """How the first encoder stage's weight / bias gradients are produced: inside the launch (partials + the reduce launch), or from the
    dpre1e rows by an immediate weight-gradient launch (_tail_enc_wgrads)."""
    if PIXEL_WGRAD_IN_KERNEL:
        es = [_grad_slot(q) for q in cfg.enc_params]
        a.f16, a.dwe1, a.dbe1 = enc["f16"].data_ptr(), es[0].data_ptr(), es[1].data_ptr()
    else:
        ge["dpre1e"] = torch.empty(16 * Tk, 64, dtype=torch.bfloat16, device=dev)
        a.dpre1e = ge["dpre1e"].data_ptr()


def _tail_enc_wgrads(cfg, enc, ge, Tk, dev):
    """The three encoder weight gradients from the tail kernels' token-major operands: (Cout, Cin, 2, 2) weights, U = the stage's output
    gradient (the bias gradient = its column sums), V = input patches with k = (kh, kw, ci); the two wide ones as recorded uses of the
    shared end-of-pass launches, the pixel-level one (J = 4 D columns) immediately."""
    comp, D = L.BF16, cfg.D
    ec = cfg.enc_params
    es = [_grad_slot(q) for q in ec]
    if not _defer_wgrad(es[4], es[5], ge["dz16"], enc["act2e"].view(Tk, 512), Tk, 256, 512, comp, (L.W_CONV_NHWC, 2, 128, False)):
        wgrad(_rm_linear(ge["dz16"]), _rm_linear(enc["act2e"].view(Tk, 512)), Tk, 256, 512, tuple(ec[4].shape), comp, layout=L.W_CONV_NHWC,
              P=2, C_other=128, device=dev, with_bias=True, into=es[4], db_into=es[5])
    if not _defer_wgrad(es[2], es[3], ge["dpre2e"], enc["act1e"].view(4 * Tk, 256), 4 * Tk, 128, 256, comp, (L.W_CONV_NHWC, 2, 64, False)):
        wgrad(_rm_linear(ge["dpre2e"]), _rm_linear(enc["act1e"].view(4 * Tk, 256)), 4 * Tk, 128, 256, tuple(ec[2].shape), comp,
              layout=L.W_CONV_NHWC, P=2, C_other=64, device=dev, with_bias=True, into=es[2], db_into=es[3])
    if "dpre1e" in ge:
        wgrad(_rm_linear(ge["dpre1e"]), _rm_linear(enc["f16"], s0=64, rows=16 * Tk, cols=4 * D), 16 * Tk, 64, 4 * D, tuple(ec[0].shape), comp,
              layout=L.W_CONV_NHWC, P=2, C_other=D, device=dev, with_bias=True, into=es[0], db_into=es[1])


class TailCfg:
    """What TailFn needs besides tensors: geometry, Taylor coefficients, the packed weight streams (one set per rollout graph) and the
    parameters whose gradient slots its backward adds into."""

    def __init__(self, B, T, Hp, Wp, C_, D, coefs, dec_params, dec_streams, enc_params, enc_streams, want_z):
        self.B, self.T, self.Hp, self.Wp, self.HW, self.C, self.D = B, T, Hp, Wp, Hp * Wp, C_, D
        self.coefs, self.dec_params, self.dec_streams = coefs, dec_params, dec_streams
        self.enc_params, self.enc_streams, self.want_z = enc_params, enc_streams, want_z


class MseMeanFn(Function):
    """MSE(pred, ref).mean() of channels-last tensors (trainer/trainer.py:189)."""

    @staticmethod
    def forward(ctx, pred, ref):
        from . import metrics
        s = metrics.metric_sums(pred, ref)
        ctx.save_for_backward(pred, ref)
        n = pred.numel()
        return s[..., 0].sum() / n

    @staticmethod
    def backward(ctx, g):
        from . import metrics
        pred, ref = ctx.saved_tensors
        grad = metrics.mse_mean_grad(pred, ref)
        return grad * g, None


class RtReduceFn(Function):
    """rt[b] = mean_l clamp(t[b][l], 0, out_T - 1) + ep with a straight-through clamp (tante.py:194-201)."""

    @staticmethod
    def forward(ctx, t, B, Lq, out_T, ep):
        rt = K.rt_reduce(t.contiguous(), B, Lq, out_T, ep)
        ctx.dims = (B, Lq, tuple(t.shape))
        return rt

    @staticmethod
    def backward(ctx, drt):
        B, Lq, shape = ctx.dims
        drt = drt.contiguous()
        dt = torch.empty(shape, dtype=torch.float32, device=drt.device)
        L.check(L.lib().tante_rt_reduce_bwd(drt.data_ptr(), B, Lq, dt.data_ptr(), _s()), "rt_reduce_bwd")
        return dt, None, None, None, None


# ---- CViT training ops (cvit.py) --------------------------------------------------------------------------------------------------
class CrossAttentionFn(Function):
    """o = softmax(q k^T / sqrt(D)) v per (batch, head); q lives in `qbuf` at column offset q_off (row stride = qbuf width), k / v in
    `kvbuf` at k_off / v_off.  Self-attention passes the packed (M, 3C) projection as both buffers."""

    @staticmethod
    def forward(ctx, qbuf, kvbuf, q_off, k_off, v_off, nb, n_head, D, Lq, Lk):
        C_ = n_head * D
        o = torch.empty(nb * Lq, C_, dtype=qbuf.dtype, device=qbuf.device)
        es = qbuf.element_size()
        L.check(L.lib().tante_cross_attention(qbuf.data_ptr() + q_off * es, kvbuf.data_ptr() + k_off * es, kvbuf.data_ptr() + v_off * es,
                                              o.data_ptr(), _DT[qbuf.dtype], nb, n_head, D, Lq, Lk, qbuf.shape[1], kvbuf.shape[1], C_, _s()),
                "tante_cross_attention")
        ctx.save_for_backward(qbuf, kvbuf, o)
        ctx.geo = (q_off, k_off, v_off, nb, n_head, D, Lq, Lk)
        return o

    @staticmethod
    def backward(ctx, dO):
        qbuf, kvbuf, o = ctx.saved_tensors
        q_off, k_off, v_off, nb, n_head, D, Lq, Lk = ctx.geo
        dO = dO.contiguous().to(qbuf.dtype)
        C_ = n_head * D
        es = qbuf.element_size()
        dq = torch.zeros_like(qbuf)
        dkv32 = torch.zeros(kvbuf.shape, dtype=torch.float32, device=kvbuf.device)
        stats = torch.empty(nb * n_head * Lq * 3, dtype=torch.float32, device=qbuf.device)
        L.check(L.lib().tante_cross_attention_bwd(qbuf.data_ptr() + q_off * es, kvbuf.data_ptr() + k_off * es, kvbuf.data_ptr() + v_off * es,
                                                  o.data_ptr(), dO.data_ptr(), dq.data_ptr() + q_off * es, dkv32.data_ptr() + 4 * k_off,
                                                  dkv32.data_ptr() + 4 * v_off, stats.data_ptr(), _DT[qbuf.dtype], nb, n_head, D, Lq, Lk,
                                                  qbuf.shape[1], kvbuf.shape[1], C_, kvbuf.shape[1], _s()), "tante_cross_attention_bwd")
        if kvbuf.dtype == torch.float32:
            dkv = dkv32
        else:
            dkv = torch.empty_like(kvbuf)
            L.check(L.lib().tante_act_fwd(dkv32.data_ptr(), L.F32, dkv.data_ptr(), _DT[kvbuf.dtype], dkv32.numel(), L.ACT_NONE, _s()), "act_fwd")
        return dq, dkv, None, None, None, None, None, None, None, None


class LayerNormAffineFn(Function):
    """y = LayerNorm(x) * gamma + beta (fp32 out): the LayerNorms of CViT whose output is itself a residual stream."""

    @staticmethod
    def forward(ctx, x, gamma, beta, eps):
        x = x.contiguous()
        y = K.layernorm_affine(x, gamma, beta, eps)
        ctx.save_for_backward(x, gamma)
        ctx.eps = eps
        return y

    @staticmethod
    def backward(ctx, dy):
        x, gamma = ctx.saved_tensors
        dy = dy.contiguous()
        Cc = x.shape[-1]
        M = x.numel() // Cc
        dx = torch.empty(x.shape, dtype=torch.float32, device=x.device)
        dg = torch.zeros(Cc, dtype=torch.float32, device=x.device)
        db = torch.zeros(Cc, dtype=torch.float32, device=x.device)
        L.check(L.lib().tante_layernorm_affine_bwd(dy.data_ptr(), _DT[dy.dtype], x.data_ptr(), _DT[x.dtype], gamma.data_ptr(), M, Cc, ctx.eps,
                                                   dx.data_ptr(), dg.data_ptr(), db.data_ptr(), _s()), "tante_layernorm_affine_bwd")
        return dx.to(x.dtype), dg, db, None


class GridEmbedFn(Function):
    """out[n] = sum_g softmax_g(-eps |coords_n - grid_g|^2) latents[g]   (cvit.py:434-438); gradients for latents and grid."""

    @staticmethod
    def forward(ctx, coords, grid, latents, eps):
        out = K.grid_embed(coords, grid.detach().contiguous(), latents.detach().contiguous(), eps)
        ctx.save_for_backward(coords, grid, latents, out)
        ctx.eps = eps
        return out

    @staticmethod
    def backward(ctx, dout):
        coords, grid, latents, out = ctx.saved_tensors
        dout = dout.contiguous().float()
        N, G, LD = coords.shape[0], grid.shape[0], latents.shape[1]
        work = torch.empty(N, dtype=torch.float32, device=coords.device)
        dl = torch.empty(G, LD, dtype=torch.float32, device=coords.device)
        dg = torch.empty(G, 2, dtype=torch.float32, device=coords.device)
        L.check(L.lib().tante_grid_embed_bwd(coords.data_ptr(), grid.detach().contiguous().data_ptr(), latents.detach().contiguous().data_ptr(),
                                             out.data_ptr(), dout.data_ptr(), N, G, LD, ctx.eps, work.data_ptr(), dl.data_ptr(), dg.data_ptr(),
                                             _s()), "tante_grid_embed_bwd")
        return None, dg, dl, None


class FourierEmbedFn(Function):
    """[cos(c . K), sin(c . K)]  (cvit.py:308-331).  The gradient of the 2 x E/2 kernel is a parameter-sized reduction over the query
    points, formed with torch expressions (like the LayerNorm-affine folding of the dense layers)."""

    @staticmethod
    def forward(ctx, coords, kernel):
        out = K.fourier_embed(coords, kernel.detach().contiguous())
        ctx.save_for_backward(coords, out)
        return out

    @staticmethod
    def backward(ctx, dout):
        coords, out = ctx.saved_tensors
        half = out.shape[1] // 2
        ddp = dout[:, half:].float() * out[:, :half] - dout[:, :half].float() * out[:, half:]     # d/d(c.K): -sin dcos + cos dsin
        return None, coords.t() @ ddp


# ---- spectral operator path training ops (enc_dec_fno.py) -------------------------------------------------------------------------
class SpectralLayerFn(Function):
    """SpectralLayer.forward without activation; x (n, Cin, H, W) fp32 contiguous, weight complex (Cin, Cout, m1, m2), w0 (Cout, Cin, 1, 1)."""

    @staticmethod
    def forward(ctx, x, weight, w0_w, w0_b, modes1, modes2):
        x = x.contiguous()
        wd = weight.detach()
        re, im = wd.real.contiguous(), wd.imag.contiguous()
        Cout, Cin = w0_w.shape[0], w0_w.shape[1]
        y = K.spectral_layer(x, re, im, modes1, modes2, w0_w.detach().view(Cout, Cin), w0_b.detach(), L.ACT_NONE)
        ctx.save_for_backward(x, re, im, w0_w)
        ctx.modes = (modes1, modes2)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, re, im, w0_w = ctx.saved_tensors
        m1, m2 = ctx.modes
        dy = dy.contiguous().float()
        n, Cin, H, W = x.shape
        Cout = w0_w.shape[0]
        w0t = w0_w.detach().view(Cout, Cin).t().contiguous()
        dx = torch.empty_like(x)
        dre, dim_ = torch.empty_like(re), torch.empty_like(im)
        nbytes = L.lib().tante_spectral_workspace_bytes(n, Cin, Cout, H, W)
        work = torch.empty(nbytes, dtype=torch.uint8, device=x.device)
        L.check(L.lib().tante_spectral_layer_bwd(x.data_ptr(), dy.data_ptr(), n, Cin, H, W, re.data_ptr(), im.data_ptr(), re.shape[2], re.shape[3],
                                                 m1, m2, w0t.data_ptr(), Cout, dx.data_ptr(), dre.data_ptr(), dim_.data_ptr(), work.data_ptr(),
                                                 nbytes, _s()), "tante_spectral_layer_bwd")
        # the 1x1 conv's parameters: channels-first "lines" (rows = pixels, column stride = H W), like the axis propagators
        HW = H * W

        def lines(t, Cc):
            return _rm_linear(t, n0=HW, s1=Cc * HW, s0=1, es=HW, cols=Cc)
        dw0 = wgrad(lines(dy, Cout), lines(x, Cin), n * HW, Cout, Cin, (Cout, Cin), L.F32, device=x.device).view(w0_w.shape)
        db0 = colsum(dy, n, Cout, HW)
        return dx, torch.complex(dre, dim_), dw0, db0, None, None


class Im2colFn(Function):
    """Patch matrix of a channels-last image, columns (kh, kw, c); backward = gather-sum of the overlapping / padded patches."""

    @staticmethod
    def forward(ctx, x, n_img, C_, H, W, P, stride, pad, out_dtype):
        cols = K.im2col(x.contiguous(), False, n_img, C_, H, W, P, P, stride, stride, pad, pad, 1, out_dtype)
        ctx.geo = (n_img, C_, H, W, P, stride, pad, x.dtype)
        return cols

    @staticmethod
    def backward(ctx, dcols):
        n_img, C_, H, W, P, stride, pad, xdt = ctx.geo
        dcols = dcols.contiguous()
        Ho, Wo = (H + 2 * pad - P) // stride + 1, (W + 2 * pad - P) // stride + 1
        dx = torch.empty(n_img, H, W, C_, dtype=xdt, device=dcols.device)
        L.check(L.lib().tante_col2im_nhwc_sized(dcols.data_ptr(), _DT[dcols.dtype], n_img, Ho, Wo, P, stride, pad, C_, None, H, W, dx.data_ptr(),
                                                _DT[xdt], _s()), "tante_col2im_nhwc_sized")
        return dx, None, None, None, None, None, None, None, None


class AvgPoolFn(Function):
    """F.adaptive_avg_pool2d to (Ht, Wt) on channels-last rows (n_img * H * W, C) -> (n_img * Ht * Wt, C): what follows an OVERLAPPING conv
    (stride < kernel) in RealConv2d.forward (enc_dec_cnn.py:104-110)."""

    @staticmethod
    def forward(ctx, x, n_img, H, W, C_, Ht, Wt, out_dtype):
        ctx.geo = (n_img, H, W, C_, Ht, Wt, x.dtype)
        return K.avgpool_nhwc(x.contiguous(), n_img, H, W, C_, Ht, Wt, L.ACT_NONE, out_dtype)

    @staticmethod
    def backward(ctx, dy):
        n_img, H, W, C_, Ht, Wt, xdt = ctx.geo
        dy = dy.contiguous()
        dx = torch.empty(n_img * H * W, C_, dtype=xdt, device=dy.device)
        L.check(L.lib().tante_avgpool_nhwc_bwd(dy.data_ptr(), _DT[dy.dtype], n_img, H, W, C_, Ht, Wt, dx.data_ptr(), _DT[xdt], _s()),
                "tante_avgpool_nhwc_bwd")
        return dx, None, None, None, None, None, None, None


class Col2imFn(Function):
    """The gather half of a ConvTranspose2d with OVERLAPPING taps (stride < kernel P; enc_dec_cnn.py:128-166): tap matrix (n_img * Hi * Wi,
    P * P * Cout), columns (kh, kw, co) -> channels-last (n_img, Hf, Wf, Cout) = summed taps + bias.  Backward: the tap matrix's gradient is
    the patch matrix of d(out) (tante_im2col with the same kernel / stride / padding), the bias's its column sum."""

    @staticmethod
    def forward(ctx, cols, bias, n_img, Hi, Wi, P, stride, pad, Cout, out_dtype):
        ctx.geo = (n_img, Hi, Wi, P, stride, pad, Cout, cols.dtype, bias is not None)
        return K.col2im_nhwc(cols.contiguous(), n_img, Hi, Wi, P, stride, pad, Cout, None if bias is None else bias.detach().float().contiguous(),
                             out_dtype)

    @staticmethod
    def backward(ctx, dout):
        n_img, Hi, Wi, P, stride, pad, Cout, cdt, has_bias = ctx.geo
        dout = dout.contiguous()
        Hf, Wf = dout.shape[1], dout.shape[2]
        dcols = db = None
        if ctx.needs_input_grad[0]:
            dcols = K.im2col(dout, False, n_img, Cout, Hf, Wf, P, P, stride, stride, pad, pad, 1, cdt)
            if dcols.shape[0] != n_img * Hi * Wi:
                raise RuntimeError("Col2imFn: geometry mismatch")
        if has_bias and ctx.needs_input_grad[1]:
            db = colsum(dout, n_img * Hf * Wf, Cout, 1)
        return dcols, db, None, None, None, None, None, None, None, None


class CropResizeFn(Function):
    """Bilinear resize (align_corners=False) of the window (crop, Hi x Wi) of a channels-last image to (Ho, Wo), channels-last or -first out."""

    @staticmethod
    def forward(ctx, full, n_img, C_, Hi, Wi, crop, Ho, Wo, nchw_out, out_dtype):
        full = full.contiguous()
        Hf, Wf = full.shape[1], full.shape[2]
        shape = (n_img, C_, Ho, Wo) if nchw_out else (n_img, Ho, Wo, C_)
        out = torch.empty(shape, dtype=out_dtype, device=full.device)
        istr = (Hf * Wf * C_, 1, Wf * C_, C_)
        ostr = (C_ * Ho * Wo, Ho * Wo, Wo, 1) if nchw_out else (Ho * Wo * C_, 1, Wo * C_, C_)
        K.resize_bilinear(full, n_img, C_, Hi, Wi, crop, istr, Ho, Wo, out, ostr, L.ACT_NONE)
        ctx.geo = (n_img, C_, Hi, Wi, crop, Ho, Wo, istr, ostr, tuple(full.shape), full.dtype)
        return out

    @staticmethod
    def backward(ctx, dout):
        n_img, C_, Hi, Wi, crop, Ho, Wo, istr, ostr, fshape, fdt = ctx.geo
        dout = dout.contiguous()
        din = torch.zeros(fshape, dtype=torch.float32, device=dout.device)
        L.check(L.lib().tante_resize_bilinear_bwd(dout.data_ptr(), _DT[dout.dtype], n_img, C_, Hi, Wi, crop[0], crop[1], *istr, Ho, Wo, *ostr,
                                                  din.data_ptr(), _s()), "tante_resize_bilinear_bwd")
        return din.to(fdt), None, None, None, None, None, None, None, None, None
