"""torch.autograd.Function wrappers over the C ABI (libmdvit_hip.so).

PyTorch is used here for device memory (caching allocator), streams and the autograd tape only;
every arithmetic op below is a HIP kernel.  Tensors are fp32, contiguous, on a CUDA(HIP) device,
activations token-major NHWC.  Saved tensors are never modified in place, so a graph can be
back-propagated twice (`retain_graph=True`, multi_train_MDViT.py:201-207).
"""
from __future__ import annotations

import ctypes as C
import itertools
import os
import weakref
from typing import Optional, Tuple

import torch

from . import _lib
from ._lib import GemmDesc, call

_key_counter = itertools.count(1)
_POISON = os.environ.get("MDVIT_POISON", "0") == "1"      # debug: NaN-fill every buffer an op allocates


def _empty(*a, **k):
    t = torch.empty(*a, **k)
    if _POISON and t.is_floating_point():
        t.fill_(float("nan"))
    return t


def _empty_like(x):
    t = torch.empty_like(x)
    if _POISON and t.is_floating_point():
        t.fill_(float("nan"))
    return t


def _next_key() -> Tuple[int, int]:
    """A fresh (key0, key1) pair for one dropout site; derived from torch's seed so that
    torch.manual_seed makes runs repeatable."""
    n = next(_key_counter)
    seed = torch.initial_seed() & 0xFFFFFFFF
    k0 = (seed * 2654435761 + n * 0x9E3779B1) & 0xFFFFFFFF
    k1 = ((n * 0x85EBCA6B) ^ (seed >> 3) ^ 0xC2B2AE35) & 0xFFFFFFFF
    return k0, k1


# ---- device-side dropout seed -------------------------------------------------------------------------
# Dropout keys are baked into kernel arguments; under HIP-graph replay they would repeat.  With the device seed
# enabled every dropout-capable kernel also mixes in two words read from device memory, and bump_seed() (one tiny
# in-graph kernel) advances them once per step.
_seed_buf = None


def enable_device_seed(flag: bool = True):
    global _seed_buf
    global _SEED_STEP
    _seed_buf = torch.tensor([0x1234567, 0x89ABCDE], dtype=torch.int32, device="cuda") if flag else None
    _SEED_STEP = torch.tensor([-1640531527, 2135587861], dtype=torch.int32, device="cuda") if flag else None


def bump_seed():
    if _seed_buf is not None:
        _seed_buf.add_(_SEED_STEP)          # int32 wrap-around add


_SEED_STEP = None


def _seed_ptr():
    return None if _seed_buf is None else C.c_void_p(_seed_buf.data_ptr())


def _p(t: Optional[torch.Tensor]):
    """the tensor's device address as a plain int (None stays None): ctypes converts an int to a pointer argument / struct field itself, and building a c_void_p object
    per operand was ~0.3 us x ~4000 operands per step of pure host time"""
    return None if t is None else t.data_ptr()


def _partials_ws(n: int, device):
    """workspace of the partial-row reductions (n floats produced): (ptr, bytes, keep-alive tensor)"""
    nbytes = _lib.load().mdvit_partials_ws_bytes(int(n))
    t = torch.empty((nbytes // 4,), device=device, dtype=torch.float32)
    return C.c_void_p(t.data_ptr()), nbytes, t


_raw_stream = torch._C._cuda_getCurrentRawStream          # (the Python-level torch.cuda.current_stream() costs ~5 us per call:
_cur_device = torch._C._cuda_getDevice                    #  with ~2000 launches per step that alone was a fifth of the host's enqueue time)


def _stream():
    """hipStream_t of torch's current stream on the current device (a plain int: see _p)"""
    return _raw_stream(_cur_device())


# ---- cheap stream plumbing.  torch.cuda.stream() / Stream.wait_stream() are Python-level wrappers that look the current stream
# up through several layers and create a fresh Event per call (~20-30 us per fork); the weight-gradient side stream forks a few
# hundred times per step, so these do the same thing on the C-level entry points with a ring of reusable events.
_set_stream_raw = torch._C._cuda_setStream                # (stream_id, device_index, device_type)
_get_stream_raw = torch._C._cuda_getCurrentStream         # device_index -> (stream_id, device_index, device_type)
_stream_objs = {}
_ev_ring, _ev_next = [], 0


def current_stream_obj():
    """torch.cuda.current_stream() without the Python-level device bookkeeping (Stream objects cached by id)"""
    sid, di, dt = _get_stream_raw(_cur_device())
    st = _stream_objs.get((sid, di))
    if st is None:
        st = _stream_objs[(sid, di)] = torch.cuda.Stream(stream_id=sid, device_index=di, device_type=dt)
    return st


def stream_wait(dst, src):
    """dst.wait_stream(src): everything enqueued on dst from now on runs after what src holds now.  Reuses events: a wait
    refers to the record that precedes it at ENQUEUE time, so re-recording the event later does not disturb it."""
    global _ev_next
    if torch.cuda.is_current_stream_capturing():
        dst.wait_stream(src)                               # captures want a fresh event node per dependency
        return
    if not _ev_ring:
        _ev_ring.extend(torch.cuda.Event() for _ in range(16))
    ev = _ev_ring[_ev_next]
    _ev_next = (_ev_next + 1) & 15
    ev.record(src)
    dst.wait_event(ev)


class use_stream:
    """with use_stream(s): torch's current stream is s (what torch.cuda.stream(s) does, minus ~15 us of Python per use)"""
    __slots__ = ("s", "prev")

    def __init__(self, s):
        self.s, self.prev = s, None

    def __enter__(self):
        if self.s is not None:
            self.prev = _get_stream_raw(self.s.device_index)
            _set_stream_raw(stream_id=self.s.stream_id, device_index=self.s.device_index, device_type=self.s.device_type)
        return self

    def __exit__(self, *exc):
        if self.prev is not None:
            sid, di, dt = self.prev
            _set_stream_raw(stream_id=sid, device_index=di, device_type=dt)
        return False


def _chk(*ts):
    for t in ts:
        if t is None:
            continue
        if not t.is_cuda:
            raise _lib.MdvitHipError("mdvit_amd ops need CUDA(HIP) tensors; there is no CPU path")
        if t.dtype != torch.float32:
            raise _lib.MdvitHipError(f"mdvit_amd ops are fp32 (got {t.dtype})")
        if not t.is_contiguous():
            raise _lib.MdvitHipError("mdvit_amd ops need contiguous tensors")


# ---- backward mode ----------------------------------------------------------------------------------
# 0: every gradient.   1: "aux sweep" of the merged two-sweep step (mdvit_amd/train.py): data gradients only;
#    parameters get no gradient EXCEPT the domain-adapter weights, which receive MINUS their gradient, so that a
#    following full backward of (aux + uni) leaves them with the uni-only gradient the reference computes by
#    freezing them during the aux sweep (multi_train_MDViT.py:198-207).
_dgrad_only = False


def set_dgrad_only(flag: bool):
    global _dgrad_only
    _dgrad_only = bool(flag)


# ---- second stream for weight gradients ---------------------------------------------------------------
# wgrad GEMMs / reductions are leaves of the backward graph: nothing in the sweep consumes them.  At small batch
# they use a fraction of the 256 CUs, so they are issued on a side HIP stream and overlap with the dgrad chain on
# the main stream.  Whoever consumes the gradients calls join_side_stream() first (mdvit_amd.train does).
_side_stream = None
_side_stream_obj = None


def enable_side_stream(flag: bool = True):
    """the weight-gradient stream; enabling twice keeps the stream (reserve_streams may have bound its hardware queue already)"""
    global _side_stream, _side_stream_obj
    if not flag:
        if _side_stream is not None:
            join_side_stream()          # nothing it still reads may be released behind its back (and the keep-alive lists do not outlive it)
        _side_stream = None
        return
    if _side_stream_obj is None:
        _side_stream_obj = torch.cuda.Stream()          # ONE stream object for the life of the process: every new stream wants a hardware queue
    _side_stream = _side_stream_obj


# ---- gradient sinks: leaf parameter -> persistent accumulation buffer ---------------------------------------
# When a sink is registered for a weight (mdvit_amd.parallel.GradAccumulator does it), its gradient is ADDED into
# the sink by the wgrad kernel itself -- on the side stream when enabled -- and autograd receives None for it:
# no gradient tensor ever crosses streams, and there is no per-parameter accumulation kernel.
_sinks = {}


def set_grad_sinks(sinks):
    """sinks: {parameter: tensor of the same shape to accumulate into}, or None/{} to disable."""
    global _sinks
    _sinks = {} if not sinks else {(p.data_ptr(), p.numel()): v for p, v in sinks.items()}


def _sink_of(t):
    if not _sinks or t is None or t.grad_fn is not None or not t.is_contiguous():
        return None
    return _sinks.get((t.data_ptr(), t.numel()))


# ---- peer streams: the G peer heads of a domain-batched forward are independent chains of small kernels --------------------
# Each head (its forward, and -- because autograd runs a node's backward on the stream its forward ran on -- its backward too) goes
# to a stream of its own, so the heads overlap each other instead of queueing behind each other on the main stream.
# Round 2 switched them off (host-bound step: their event traffic cost more than they overlapped).  Round 3 first re-enabled two of them
# ("+2.5 %", measured against a step whose aux sweep was ALREADY serialised by a fifth stream) and then found the real cost: see peer_stream.
# MDVIT_PEER_STREAMS=1 forces dedicated streams (A/B), 0 / auto: none.
_peer_streams = []
_peer_mode = os.environ.get("MDVIT_PEER_STREAMS", "auto")
_use_peer_streams = _peer_mode != "0"
_peer_auto_max_images = 32
_graph_peers = os.environ.get("MDVIT_GRAPH_PEERS", "0") != "0"      # fork the peer streams inside a HIP-graph capture too
_peer_stream_count = max(1, int(os.environ.get("MDVIT_PEER_STREAM_COUNT", "2")))   # heads i, i + count, ... share a stream


def peer_stream(i: int, n_images: int = 0):
    """the i-th peer stream, or None when disabled / capturing (a captured graph keeps the single-stream order) / in auto mode for large batches"""
    if not _use_peer_streams or (torch.cuda.is_current_stream_capturing() and not _graph_peers):
        return None
    if _peer_mode == "auto":
        # The GPU runs FOUR hardware queues.  main + weight-gradient stream + aux-sweep stream + two peer streams = five: the stream created last
        # (the aux sweep's) then SHARES the main stream's queue and the whole data-gradient-only sweep runs behind the full sweep instead of next
        # to it (tools/sweep_timeline.py: aux sweep runnable at 28.5 ms instead of 10.4; step 40.4 ms with two peer streams, 38.0 with one, 37.3
        # without -- 396 -> 429 images/s).  Letting the peer heads borrow the two streams that idle during the forward keeps four queues but puts the
        # full sweep's peer-head backward in front of the aux sweep on its stream: 40.3 ms (all four heads on the aux stream: 41.7 ms; all four on the weight-gradient stream: 37.3 against 36.6 without) -- whenever the aux
        # stream carries anything of the full sweep, the aux sweep ends up behind it.  So: no peer streams next to the two-stream sweeps.
        return None
    i %= _peer_stream_count
    while len(_peer_streams) <= i:
        _peer_streams.append(torch.cuda.Stream())
    return _peer_streams[i]


_branch_stream_obj = None
_use_branch_stream = os.environ.get("MDVIT_BRANCH_STREAM", "1") != "0"


def branch_stream():
    """ONE extra stream for a model with two independent trunks (TransFuse: the DeiT branch next to the ResNet branch, forward and -- autograd runs a
    node's backward on its forward's stream -- backward).  With the main and the weight-gradient stream that is three of the four hardware queues.
    None while a HIP graph is captured or when switched off (MDVIT_BRANCH_STREAM=0)."""
    global _branch_stream_obj
    if not _use_branch_stream or (torch.cuda.is_current_stream_capturing() and not _graph_streams):
        return None
    if _branch_stream_obj is None:
        _branch_stream_obj = torch.cuda.Stream()
    return _branch_stream_obj


_side_keepalive = []     # tensors the side stream still reads; holding a reference also stops autograd from
                         # accumulating into them IN PLACE on the main stream (it only does so when it is the sole owner)
_side_blocks = []        # (event recorded on the side stream after a block's launches, number of keep-alive entries up to there)


def _release_finished_side_blocks():
    """Drop the keep-alive references of side-stream blocks whose kernels have finished (the event recorded behind them has
    completed), so an un-joined sweep does not hold a whole backward's worth of gradients until the next join."""
    done = 0
    while _side_blocks and _side_blocks[0][0].query():
        done = _side_blocks.pop(0)[1]
    if done:
        del _side_keepalive[:done]
        for i, (ev, n) in enumerate(_side_blocks):
            _side_blocks[i] = (ev, n - done)


def join_side_stream():
    if _side_stream is not None:
        cur = current_stream_obj()
        stream_wait(cur, _side_stream)
        _side_keepalive.clear()
        _side_blocks.clear()
        for ev, owner, _ts, _n in _side_groups:           # tensors of ANOTHER stream (a branch stream's blocks): that stream waits too before they go back to its pool
            if owner != cur:
                owner.wait_event(ev)
        _side_groups.clear()
        _side_held[0] = 0
        for owner, _ts, _n in _side_open.values():
            if owner != cur:
                owner.wait_stream(_side_stream)           # (enqueued after the join above: the side stream holds nothing newer)
        _side_open.clear()


_side_open = {}      # owner stream handle -> [owner stream, tensors, bytes]: protected tensors that have no event yet (a group closes every 16 tensors / 1 GiB)


def _close_side_group(key):
    """record the side stream's event behind an open group (every launch that reads its tensors is enqueued by now) and apply the hold bound"""
    owner, ts, n = _side_open.pop(key)
    ev = torch.cuda.Event()
    ev.record(_side_stream)
    _side_groups.append((ev, owner, ts, n))
    _side_held[0] += n
    _trim_side_groups(_side_hold_limit)


class _on_side:
    """with _on_side(t1, t2, ...): launches go to the side stream, ordered after everything already enqueued on the
    main stream; the listed main-stream tensors are protected from reuse until the side work is done (bounded: _side_protect)."""

    def __init__(self, *tensors, foreign=True):
        """foreign: True -- the first tensor is the upstream gradient (the one protected tensor a node of another stream may have allocated); "all" -- every listed tensor
        is an upstream gradient (a node with several of them: _ComposeHeads.backward); False -- none"""
        self.tensors = [t for t in tensors if t is not None]
        self.foreign = foreign if (foreign and bool(tensors) and tensors[0] is not None) else False
        self.ctx = None

    def __enter__(self):
        if _side_stream is None:
            return self
        owner = current_stream_obj()
        stream_wait(_side_stream, owner)
        self.bounded = bool(_side_hold_limit) and not torch.cuda.is_current_stream_capturing()
        if self.bounded:
            # one open group per owning stream (TransFuse's two trunks alternate in the backward: their tensors go back to different pools)
            self.key = key = owner.cuda_stream
            grp = _side_open.get(key)
            if grp is None:
                grp = _side_open[key] = [owner, [], 0]
            grp[1].extend(self.tensors)
            for t in self.tensors:
                grp[2] += t.numel() * t.element_size()
            if self.tensors and self.foreign:
                # the FIRST tensor is the upstream gradient: the one protected tensor a node of ANOTHER stream may have allocated (a peer-head / branch-stream
                # join).  Its block goes back to THAT stream's pool, which never waits for the side stream -- so the allocator is told.
                for t in (self.tensors if self.foreign == "all" else self.tensors[:1]):
                    t.record_stream(_side_stream)
        else:
            for t in self.tensors:
                t.record_stream(_side_stream)
            _side_keepalive.extend(self.tensors)
        self.ctx = use_stream(_side_stream)
        self.ctx.__enter__()
        return self

    def __exit__(self, *exc):
        if self.ctx is not None:
            if self.bounded:
                grp = _side_open.get(self.key)
                if grp is not None and (len(grp[1]) >= 16 or grp[2] >= (1 << 30)):
                    _close_side_group(self.key)
            else:
                marked = _side_blocks[-1][1] if _side_blocks else 0
                if len(_side_keepalive) - marked >= 32 and not torch.cuda.is_current_stream_capturing():
                    ev = torch.cuda.Event()                 # one marker per ~32 protected tensors: host cost stays negligible
                    ev.record(_side_stream)
                    _side_blocks.append((ev, len(_side_keepalive)))
                    _release_finished_side_blocks()
            self.ctx.__exit__(*exc)
        return False


def _flat_like(*protos):
    """Allocate the (small) outputs of one op as slices of ONE buffer, in order, so the library can clear them
    with a single zero-fill launch.  protos: tensors (shape donors) or shapes; None entries pass through."""
    shapes = [None if t is None else (tuple(t.shape) if isinstance(t, torch.Tensor) else tuple(t)) for t in protos]
    dev = next(t.device for t in protos if isinstance(t, torch.Tensor))
    sizes = [0 if sh is None else int(torch.Size(sh).numel()) for sh in shapes]
    flat = _empty((max(sum(sizes), 1),), device=dev, dtype=torch.float32)
    out, off = [], 0
    for sh, n in zip(shapes, sizes):
        out.append(None if sh is None else flat[off:off + n].view(sh))
        off += n
    return out


def _c(t: torch.Tensor) -> torch.Tensor:
    return t if t.is_contiguous() else t.contiguous()


# ------------------------------------------------------------------------------------------------
# raw GEMM helper
# ------------------------------------------------------------------------------------------------
def gemm(A, B, out, M, N, K, *, lda, ldb, ldc, trans_a=False, trans_b=True, bias=None, out2=None, epi=_lib.EPI_NONE,
         e_drop=0.0, e_key=(0, 0), e_rowscale=None, e_rows_per_scale=1,
         residual=None, ldr=0, gelu_u=None, ldu=0, allow_split=False, accumulate=False, precision=None, colsum_a=None, rc=None, conv=None, conv_wgrad_nchw=False):
    """rc = (a, lda, b, ldb, bias, k): DGELU with the pre-activation recomputed in the kernel (include/mdvit_hip.h).
    conv = (C, H, W, Ho, Wo, stride, dilation): A is the NHWC image, gathered as the implicit im2col operand of a 3x3 convolution."""
    if precision is None:
        precision = min(_gemm_precision, 1) if (bool(trans_b) != bool(trans_a)) else 0      # built for NT and TN
    d = GemmDesc()
    d.A, d.B, d.C, d.C2 = A, B, out, out2
    d.lda, d.ldb, d.ldc = lda, ldb, ldc
    d.M, d.N, d.K = M, N, K
    d.trans_a, d.trans_b = int(trans_a), int(trans_b)
    d.bias = bias
    d.epi = epi
    d.e_drop_p, d.e_key0, d.e_key1 = e_drop, e_key[0], e_key[1]
    d.e_rowscale, d.e_rows_per_scale = e_rowscale, e_rows_per_scale
    d.residual, d.ldr = residual, ldr
    d.gelu_u, d.ldu = gelu_u, ldu
    d.allow_split = int(allow_split)
    d.accumulate = int(accumulate)
    d.precision = int(precision)
    d.colsum_a = colsum_a
    d.drop_seed = _seed_ptr() if e_drop > 0 else None
    if rc is not None:
        d.rc_a, d.rc_lda, d.rc_b, d.rc_ldb, d.rc_bias, d.rc_k = rc
    if conv is not None:
        d.conv_c, d.conv_h, d.conv_w, d.conv_ho, d.conv_wo, d.conv_stride, d.conv_dilation = conv[:7]
        d.conv_up = conv[7] if len(conv) > 7 else 0
        d.conv_wgrad_nchw = int(bool(conv_wgrad_nchw))
    ws = None
    if allow_split:
        need = _lib.load().mdvit_gemm_ws_bytes(C.byref(d))
        if need:
            ws = _empty((need // 4,), device=torch.device("cuda", torch.cuda.current_device()), dtype=torch.float32)
            d.ws, d.ws_bytes = _p(ws), need
    if _events is None:
        call("mdvit_gemm_f32", C.byref(d), _stream())
        return
    kepi = 1 if epi == _lib.EPI_GELU_DUAL else (4 if rc is not None else 2) if epi == _lib.EPI_DGELU else \
        3 if (e_drop > 0 or e_rowscale is not None or residual is not None) else 0
    pkey = (M, N, K, bool(trans_a), bool(trans_b), kepi, int(precision), bool(allow_split), colsum_a is not None, conv is not None)
    name = _plan_cache.get(pkey)
    if name is None:
        buf = C.create_string_buffer(160)
        call("mdvit_gemm_kernel_name", C.byref(d), buf, 160)       # the symbol as rocprofv3 prints it
        tm, tn, sp = C.c_int32(), C.c_int32(), C.c_int32()
        call("mdvit_gemm_plan", C.byref(d), C.byref(tm), C.byref(tn), C.byref(sp))
        name = _plan_cache[pkey] = (buf.value.decode().replace("+splitk_reduce", ""), sp.value)   # the main kernel is what is timed
    name, spv = name
    if _events_by_shape:
        name += " M=%d N=%d K=%d sp=%d" % (M, N, K, spv)
    if not _event_wanted(name):
        call("mdvit_gemm_f32", C.byref(d), _stream())
        return
    # the GEMM kernel's own begin / end timestamps (hipExtLaunchKernelGGL through mdvit_timing_arm) -- what rocprofv3 reports for it;
    # with a split K range that is the main kernel alone (its slab reduction is a second, separately named launch)
    e0, e1 = _lib_event(), _lib_event()
    call("mdvit_timing_arm", e0, e1)
    try:
        call("mdvit_gemm_f32", C.byref(d), _stream())
    finally:
        call("mdvit_timing_arm", None, None)
    # algorithmic HBM bytes of the launch: A, B read once, C (and C2 / residual / gelu_u) once
    nbytes = 4.0 * (M * K + N * K + M * N * (1 + (out2 is not None) + (residual is not None) + (gelu_u is not None)))
    if rc is not None:
        nbytes += 4.0 * (M + N) * rc[5]
    _events.append((name, 2.0 * M * N * K, nbytes, e0, e1))


# ---- GEMM arithmetic ---------------------------------------------------------------------------------------
# 0 "fp32": fp32-input MFMA, bit-for-bit an fmaf chain.  1 "bf16x3": operands split hi+lo into bf16 while staged,
# hi*hi + hi*lo + lo*hi on the bf16 matrix cores, fp32 accumulate (~1e-5 relative).  The bf16x3 kernel wants both
# operands k-contiguous, so the data-gradient GEMMs read a transposed copy of the weight (wt()).
#   "bf16" : the speed mode -- operands rounded to ONE bf16 plane, one MFMA per product, fp32 accumulate (~3e-3 relative per GEMM);
#            never the parity mode.  (Layouts the plane kernels do not cover fall back to the bf16x3 kernels.)
_PRECISIONS = {"fp32": 0, "bf16x3": 1, "bf16": 2}
_gemm_precision = _PRECISIONS[os.environ.get("MDVIT_GEMM_PRECISION", "bf16x3")]
if _gemm_precision == 2 and os.environ.get("MDVIT_MLP_RC_ONE_PLANE", "1") != "0":
    _lib.on_load(lambda lib: lib.mdvit_mlp_rc_planes(1))
_use_plane_gemm = os.environ.get("MDVIT_PLANE_GEMM", "1") != "0"      # 0: the split-while-staging kernels of gemm.hip everywhere
# bf16x3: plane kernels for K >= this.  Measured in the step (profiles/r02g_ab.txt): the plane NT kernel is a wash-to-slower against the
# split-while-staging kernel at equal arithmetic (both sit at the same ~30 % of the MFMA roof: the limit is not the split VALU), so
# the parity mode keeps the calibrated gemm.hip planner and the plane kernels serve the bf16 speed mode (half the operand bytes).
_plane_min_k = int(os.environ.get("MDVIT_PLANE_MIN_K", "1000000"))
_tn_kernel = os.environ.get("MDVIT_TN_KERNEL", "1") != "0"
if not _tn_kernel:                                                      # A/B: weight-gradient GEMMs on the general template instead of gemm_tn.hip
    _lib.on_load(lambda lib: lib.mdvit_gemm_tn_config(0, -1, 0))                      # (the implicit-convolution weight gradient lives in gemm_tn.hip: conv3x3_dense then takes im2col + GEMM)
_ph_gemm = os.environ.get("MDVIT_PH_GEMM", "1") != "0"                # the 256-wide phase-split kernel for the products it prefers (0: A/B switch)
_pm_gemm = os.environ.get("MDVIT_PM_GEMM", "1") != "0"                # the 128-row phase-split kernel for the mid-size products (0: A/B switch; the C-level block entry follows)
if not _pm_gemm:
    _lib.on_load(lambda lib: lib.mdvit_gemm_pm_config(-1))
if not _ph_gemm:
    _lib.on_load(lambda lib: lib.mdvit_gemm_ph_config(-1))
_plane_rc = os.environ.get("MDVIT_PLANE_RC", "0") != "0"              # the recomputing fc2 data gradient of the C = 128 MLPs on the plane kernel


_mlp_rc_one_plane = os.environ.get("MDVIT_MLP_RC_ONE_PLANE", "1") != "0"      # bf16 mode: the register-chained MLP kernels on one plane per operand too (0: bf16x3 there, round 4's mixed mode; A/B)


def set_gemm_precision(name: str):
    global _gemm_precision
    _gemm_precision = _PRECISIONS[name]
    _lib.load().mdvit_mlp_rc_planes(1 if (_gemm_precision == 2 and _mlp_rc_one_plane) else 2)


def gemm_precision() -> str:
    return {0: "fp32", 1: "bf16x3", 2: "bf16"}[_gemm_precision]


def _nplanes() -> int:
    return 1 if _gemm_precision == 2 else 2


_wt_cache = {}       # id(leaf weight) -> (weakref to it, version, data_ptr, rows, cols, ld, W^T); the weakref guards against id reuse

# The derived-weight caches (W^T, weight planes, implicit-convolution layouts) are filled lazily ON THE STREAM THAT FIRST NEEDS THEM and a
# later hit -- possibly from another stream -- returns the buffer with no stream ordering.  Every lazy fill therefore raises this flag;
# whoever runs two sweeps on two streams (train.mdvit_train_step) reads it with take_cache_fill_flag() after enqueueing the first sweep
# and, when set, orders the second stream after it (a cold step only: the per-step refresh_*() launches on the main stream precede
# both sweeps and do not raise the flag).
_cache_filled = False


def take_cache_fill_flag() -> bool:
    """True if a derived-weight cache entry was (re)built lazily since the last call; clears the flag."""
    global _cache_filled
    f, _cache_filled = _cache_filled, False
    return f


def wt(W):
    """W [N,K] (2-D view, or a contiguous conv weight read as [out, in*kh*kw]) -> contiguous W^T [K,N].  Cached per leaf
    parameter; refreshed (into the same buffer) when the parameter's version counter moves -- normally for all weights
    at once by refresh_transposes() at the start of a step."""
    pre = getattr(W, "_mdvit_wt", None)       # a per-step temporary whose W^T was produced with it (transpose_weights_batch: the peer heads' composed weights)
    if pre is not None and pre[0] == W._version:
        return pre[1]
    N, K, ld = _ld_view(W)
    leaf = W.grad_fn is None and W.requires_grad
    tag = (W._version + (_weights_epoch << 32), W.data_ptr(), N, K, ld)      # the epoch moves when an optimizer writes through raw pointers
    out = None
    if leaf:
        hit = _wt_cache.get(id(W))
        if hit is not None and hit[0]() is W:
            if hit[1:6] == tag:
                return hit[6]
            if hit[2:6] == tag[1:]:
                out = hit[6]                  # same storage and shape, new values: transpose into the existing buffer
    if out is None:
        out = torch.empty((K, N), device=W.device, dtype=torch.float32)
    call("mdvit_transpose_f32", _p(W), ld, _p(out), N, K, _stream())
    if leaf:
        # (only a CACHED entry can be hit from another stream later; the transpose of a non-leaf weight -- the peer heads' composed weights, every
        # step -- is a temporary of the sweep that made it.  Raising the flag for those serialised the aux sweep behind the full one in every step.)
        global _cache_filled
        _cache_filled = True
        key = id(W)
        _wt_cache[key] = (weakref.ref(W, lambda _r, key=key: _wt_cache.pop(key, None)),) + tag + (out,)
    return out


def transpose_weights_batch(Ws):
    """W^T of up to 24 non-leaf weights [N, K] (row-contiguous views) in ONE launch, attached to the tensors: wt(W) finds it in either sweep instead of
    transposing per use (the peer heads' composed weights: 20 per sweep)."""
    Ws = [W for W in Ws if W is not None and W.is_cuda and W.dtype == torch.float32]
    for i in range(0, len(Ws), 24):
        chunk = Ws[i:i + 24]
        views = [_ld_view(W) for W in chunk]
        outs = [torch.empty((K, N), device=W.device, dtype=torch.float32) for W, (N, K, _) in zip(chunk, views)]
        n = len(chunk)
        call("mdvit_transpose_batch", n, _vp_array([_p(W) for W in chunk]), (C.c_int64 * n)(*[v[2] for v in views]), _vp_array([_p(o) for o in outs]),
             (C.c_int32 * n)(*[v[0] for v in views]), (C.c_int32 * n)(*[v[1] for v in views]), _stream())
        for W, o in zip(chunk, outs):
            W._mdvit_wt = (W._version, o)


_wt_table = None     # (signature, device int64 table [n,5], blocks per item)


def refresh_transposes():
    """Re-transpose EVERY cached weight in one launch (call once per step, after the optimizer update): ~100 weights would
    otherwise each pay their own launch the first time a data-gradient GEMM touches them."""
    global _wt_table
    refresh_conv_weights()
    entries = []
    for key, hit in list(_wt_cache.items()):
        W = hit[0]()
        if W is not None and hit[2:6] == (W.data_ptr(),) + tuple(hit[3:6]) and W.data_ptr() == hit[2]:
            entries.append((key, W, hit))
    if not entries:
        refresh_weight_planes()           # plane-only configurations still get their one batched split per step
        return
    sig = tuple((h[2], h[6].data_ptr(), h[5], h[3], h[4]) for _, _, h in entries)       # (in, out, ld, rows, cols)
    if _wt_table is None or _wt_table[0] != sig:
        if torch.cuda.is_current_stream_capturing():
            return                            # the table is host-built: it must exist before capture (warm-up steps build it)
        rows = [[i, o, ld, n, k] for (i, o, ld, n, k) in sig]
        dev = entries[0][1].device
        tiles = max(((n + 31) // 32) * ((k + 31) // 32) for (_, _, _, n, k) in sig)
        _wt_table = (sig, torch.tensor(rows, dtype=torch.int64, device=dev), int(min(tiles, 64)))
    call("mdvit_transpose_many", _p(_wt_table[1]), len(entries), _wt_table[2], _stream())
    for key, W, hit in entries:
        _wt_cache[key] = (hit[0], W._version + (_weights_epoch << 32)) + tuple(hit[2:])
    refresh_weight_planes()


# ---- weight planes: every GEMM weight pre-split into bf16 planes ONCE per optimizer step -------------------------------------
# The plane GEMMs (csrc/gemm_bp.hip) take their B operand as bf16 planes: W [N,K] for the forward layers, W^T [K,N] for the data
# gradients.  Leaf parameters are cached (both orientations, refreshed together by one launch at the start of a step); weights
# that are computed on the tape (composed / sliced) are split per call.  Staleness: the tag holds the parameter's version counter
# AND a global epoch that optimizers writing through raw pointers (optim.FusedAdamW) bump after every update.
_weights_epoch = 0
_wp_cache = {}       # id(leaf weight) -> {"ref": weakref, "planes": {transposed: tensor}, "tags": {transposed: tag}}
_wp_table = {}       # plane count -> (signature, device table, tiles)


def mark_weights_updated():
    """Call after parameters were modified in a way autograd's version counters do not see (a kernel writing through data_ptr())."""
    global _weights_epoch
    _weights_epoch += 1


def _wplanes(W, transposed: bool, planes: Optional[int] = None):
    """bf16 planes [P, rows, cols] of W (rows, cols = N, K) or of W^T (K, N).  P = the GEMM mode's plane count (bf16x3: hi + lo, bf16: hi), or
    `planes` = 2 for the register-chained kernels of mlp_rc.hip, which compute in bf16x3 in either mode."""
    N, K, ld = _ld_view(W)
    P = _nplanes() if planes is None else planes
    rows, cols = (K, N) if transposed else (N, K)
    leaf = W.grad_fn is None and W.requires_grad
    buf = None
    slot = (transposed, P)
    if leaf:
        tag = (W._version, _weights_epoch, W.data_ptr(), N, K, ld, P)
        ent = _wp_cache.get(id(W))
        if ent is None or ent["ref"]() is not W:
            key = id(W)
            ent = _wp_cache[key] = {"ref": weakref.ref(W, lambda _r, key=key: _wp_cache.pop(key, None)), "planes": {}, "tags": {}}
        buf = ent["planes"].get(slot)
        if buf is not None and ent["tags"].get(slot) == tag:
            return buf
        if buf is not None and tuple(buf.shape) != (P, rows, cols):
            buf = None
    if buf is None:
        buf = torch.empty((P, rows, cols), device=W.device, dtype=torch.bfloat16)
    call("mdvit_split_planes_t", _p(W), ld, _p(buf), cols, rows * cols, N, K, int(transposed), P, _stream())
    if leaf:
        global _cache_filled
        _cache_filled = True
        ent["planes"][slot] = buf
        ent["tags"][slot] = tag
    return buf


def refresh_weight_planes():
    """Re-split EVERY cached weight orientation in one launch per plane count (start of a step, after the optimizer update)."""
    for P in (1, 2):
        items, live = [], []
        for key, ent in list(_wp_cache.items()):
            W = ent["ref"]()
            if W is None:
                continue
            N, K, ld = _ld_view(W)
            for slot, buf in ent["planes"].items():
                tr, Ps = slot
                rows, cols = (K, N) if tr else (N, K)
                if Ps != P or tuple(buf.shape) != (P, rows, cols):
                    continue
                items.append((W.data_ptr(), buf.data_ptr(), ld, N, K, int(tr), cols, rows * cols))
                live.append((ent, slot, (W._version, _weights_epoch, W.data_ptr(), N, K, ld, P)))
        if not items:
            continue
        sig = tuple(items)
        tab = _wp_table.get(P)
        if tab is None or tab[0] != sig:
            if torch.cuda.is_current_stream_capturing():
                continue                          # host-built table: must exist before capture (warm-up steps build it)
            dev = torch.device("cuda", torch.cuda.current_device())
            tiles = max(((it[3] + 31) // 32) * ((it[4] + 31) // 32) for it in items)
            tab = _wp_table[P] = (sig, torch.tensor([list(it) for it in items], dtype=torch.int64, device=dev), int(min(tiles, 64)))
        call("mdvit_split_planes_many", _p(tab[1]), len(items), tab[2], P, _stream())
        for ent, slot, tag in live:
            ent["tags"][slot] = tag


class Planes:
    """An activation stored as bf16 planes [P, M, K] (hi, lo) -- the A-operand format of the plane GEMMs.  `h` is the autograd
    handle: an fp32 tensor of the LOGICAL shape whose values are never read (stride-0 phantom), so producers and consumers stay
    ordinary autograd nodes while the data travels in `p`."""
    __slots__ = ("p", "h")

    def __init__(self, p, h=None):
        self.p, self.h = p, h

    @property
    def shape(self):
        return self.p.shape[1:] if self.h is None else self.h.shape


def to_planes(x2d):
    """fp32 [M, K] -> bf16 planes [P, M, K] (one pass; for producers that do not write planes themselves yet)"""
    M, K = x2d.shape
    P = _nplanes()
    out = _empty((P, M, K), device=x2d.device, dtype=torch.bfloat16)
    call("mdvit_split_planes", _p(x2d), K, _p(out), K, M * K, M, K, P, _stream())
    return out


_ph_cache = {}


def _ph_prefers(M, N, K, planes=2, reads=False, plain=False) -> bool:
    """a phase-split plane kernel takes [M, K] x [N, K]^T: the 256-wide one (csrc/gemm_ph.hip, mdvit_gemm_ph_prefers_epi; reads: the epilogue reads an [M, N] operand)
    or, for fp32 activations against two weight planes, the 128-row one of the mid-size products (csrc/gemm_pm.hip, mdvit_gemm_pm_prefers) -- and, plain: a product
    with no epilogue (a data gradient) that may be split along K, that tile over 2-4 K ranges (mdvit_gemm_pm_splits: long K, few tiles)"""
    k = (M, N, K, planes, bool(reads), bool(plain))
    r = _ph_cache.get(k)
    if r is None:
        lib = _lib.load()
        r = _ph_cache[k] = bool(lib.mdvit_gemm_ph_prefers_epi(M, N, K, planes, int(bool(reads)))) or bool(_pm_gemm and lib.mdvit_gemm_pm_prefers(M, N, K, planes, 1)) \
            or bool(_pm_gemm and plain and not reads and lib.mdvit_gemm_pm_splits(M, N, K, planes, 1) > 1)
    return r


def _plane_ok(M, N, K, reads=False, plain=False) -> bool:
    """the plane kernels cover this product (otherwise: the split-while-staging kernels of gemm.hip).  bf16x3: the products the 256-wide kernel
    takes (the 64 / 128 plane tiles lose to gemm.hip in the step: profiles/r02_gemm_step_ab.txt); bf16: every legal one."""
    if not (_use_plane_gemm and _gemm_precision >= 1 and K % 32 == 0 and N % 4 == 0):
        return False
    return _gemm_precision == 2 or K >= _plane_min_k or (_ph_gemm and _ph_prefers(M, N, K, 2, reads, plain))


def _pp(v):
    """tensor | ctypes pointer | None -> ctypes pointer | None"""
    return _p(v) if isinstance(v, torch.Tensor) else v


def gemm_nt(x, W, out, M, N, K, *, w_transposed=False, bias=None, epi=_lib.EPI_NONE, e_drop=0.0, e_key=(0, 0), e_rowscale=None,
            e_rows_per_scale=1, residual=None, gelu_u=None, U=None, out_planes=None, allow_split=False, accumulate=False, rc=None,
            ldr=None, ldu=None):
    """out[M,N] = x[M,K] @ B^T with B = W [N,K] (w_transposed=False: a forward layer) or B = W^T where W is [K,N] (the data
    gradient dx = g W).  x: fp32 tensor [M,K] (split while staged) or bf16 planes [P,M,K]; out: fp32 tensor or None;
    out_planes: bf16 planes [P,M,N] or None.  rc = (x_planes, W1, b1, rc_k): DGELU with the recomputed pre-activation."""
    B = _wplanes(W, w_transposed)                     # [P, N, K]
    P = B.shape[0]
    d = _lib.PlaneGemmDesc()
    a_f32 = x.dtype == torch.float32
    assert (ldr is None or ldr == N or residual is None) and (ldu is None or ldu == N or gelu_u is None)
    d.A = _p(x); d.lda = K; d.a_plane = 0 if a_f32 else M * K; d.a_f32 = int(a_f32)
    d.B = _p(B); d.ldb = K; d.b_plane = N * K
    d.planes = P; d.trans = 0
    d.M, d.N, d.K = M, N, K
    if out is not None:
        d.C = _pp(out); d.ldc = N
    if out_planes is not None:
        d.Cp = _p(out_planes); d.ldcp = N; d.c_plane = M * N
    if U is not None:
        d.U = _pp(U); d.ldu_out = N
    d.bias = _pp(bias)
    d.epi = epi
    d.e_drop_p, d.e_key0, d.e_key1 = e_drop, e_key[0], e_key[1]
    d.e_rowscale, d.e_rows_per_scale = _pp(e_rowscale), e_rows_per_scale
    d.residual, d.ldr = _pp(residual), N
    d.gelu_u, d.ldu = _pp(gelu_u), N
    keep = None
    if rc is not None:
        ra, W1, b1, rk = rc
        rb = _wplanes(W1, False)
        keep = rb
        d.rc_a = _p(ra); d.rc_lda = rk; d.rc_a_plane = M * rk
        d.rc_b = _p(rb); d.rc_ldb = rk; d.rc_b_plane = N * rk
        d.rc_bias = _pp(b1); d.rc_k = rk
    d.allow_split = int(allow_split)
    d.accumulate = int(accumulate)
    d.drop_seed = _seed_ptr() if e_drop > 0 else None
    ws = None
    if allow_split:
        need = _lib.load().mdvit_gemm_planes_ws_bytes(C.byref(d))
        if need:
            ws = _empty((need // 4,), device=B.device, dtype=torch.float32)
            d.ws, d.ws_bytes = _p(ws), need
    if _events is None:
        call("mdvit_gemm_planes", C.byref(d), _stream())
        return
    kepi = 1 if epi == _lib.EPI_GELU_DUAL else (4 if rc is not None else 2) if epi == _lib.EPI_DGELU else \
        3 if (e_drop > 0 or e_rowscale is not None or residual is not None) else 0
    pkey = ("bp", M, N, K, kepi, P, a_f32, bool(allow_split))
    plan = _plan_cache.get(pkey)
    if plan is None:
        tm, tn, sp = C.c_int32(), C.c_int32(), C.c_int32()
        call("mdvit_gemm_planes_plan", C.byref(d), C.byref(tm), C.byref(tn), C.byref(sp))
        plan = _plan_cache[pkey] = (tm.value, tn.value, sp.value)
    if plan[0] == 256:       # csrc/gemm_ph.hip, as rocprofv3 prints it
        name = "gemm_ph_kernel<%d, %s, %d>%s" % (P, "true" if a_f32 else "false", kepi, "+splitk_reduce" if plan[2] > 1 else "")
    elif plan[1] == 160 or (plan[0] == 128 and plan[1] == 128 and a_f32 and P == 2 and (_lib.load().mdvit_gemm_pm_prefers(M, N, K, P, 1) == 7 or
                                                                                          (plan[2] > 1 and _lib.load().mdvit_gemm_pm_splits(M, N, K, P, 1) == plan[2]))):
        name = "gemm_pm_kernel<%d, 2, %d>%s" % (3 if plan[1] == 160 else 2, kepi, "+splitk_reduce" if plan[2] > 1 else "")          # csrc/gemm_pm.hip
    else:
        name = "gemm_bp_nt_kernel<%d, %d, %d, %s, %d>%s" % (plan[0], plan[1], P, "true" if a_f32 else "false", kepi, "+splitk_reduce" if plan[2] > 1 else "")
    if _events_by_shape:
        name += " M=%d N=%d K=%d sp=%d" % (M, N, K, plan[2])
    if not _event_wanted(name):
        call("mdvit_gemm_planes", C.byref(d), _stream())
        return
    st = current_stream_obj()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(st)
    call("mdvit_gemm_planes", C.byref(d), _stream())
    e1.record(st)
    abytes = 4.0 if a_f32 else 2.0 * P
    nbytes = abytes * M * K + 2.0 * P * N * K + M * N * (4.0 * (out is not None) + 2.0 * P * (out_planes is not None) + 4.0 * (U is not None)
                                                        + 4.0 * (residual is not None) + 4.0 * (gelu_u is not None))
    if rc is not None:
        nbytes += 2.0 * P * (M + N) * rc[3]
    _events.append((name, 2.0 * M * N * K, nbytes, e0, e1))


_lin_rc = os.environ.get("MDVIT_LINEAR_RC", "1") != "0"     # short-K Linear layers on the streaming kernel (mlp_rc.hip: mdvit_linear_rc); 0: the tiled GEMM (A/B)


def _lin_rc_ok(M, N, K) -> bool:
    """y[M,N] = x[M,K] W^T runs on mdvit_linear_rc: bf16x3 mode, K = 64 / 128, N % 32 == 0 (bit-identical to the tiled GEMM: same products, same order)"""
    return _lin_rc and _gemm_precision >= 1 and K in (64, 128) and N % 32 == 0 and 32 <= N <= 4096 and M >= 1024


def _linear_rc(x, W, transposed, bias, y, M, N, K, drop_p=0.0, key=(0, 0), rowscale=None, rows_per_scale=1, residual=None):
    Wp = _wplanes(W, transposed, 2)         # [2, N, K]
    call("mdvit_linear_rc", _p(x), K, _p(Wp), N * K, _p(bias), _p(y), N, M, N, K, drop_p, key[0], key[1], _p(rowscale), rows_per_scale, _p(residual), N,
         _seed_ptr() if drop_p > 0 else None, _stream())
    del Wp


def _dgrad(g, W, dx, M, K, N, ldb, **kw):
    """dx[M,K] = g[M,N] @ W[N,K]: NN on the fp32 path; NT against the cached W^T (planes) on the bf16x3 / bf16 paths."""
    if _lin_rc_ok(M, K, N) and ldb == K and not (set(kw) - {"allow_split"}):
        _linear_rc(g, W, True, None, dx, M, K, N)
        return
    if _plane_ok(M, K, N, kw.get("gelu_u") is not None, plain=set(kw) == {"allow_split"} and bool(kw["allow_split"])) and "rc" not in kw and "precision" not in kw:
        gemm_nt(g, W, dx, M, K, N, w_transposed=True, **kw)
        return
    if _gemm_precision:
        Wt = wt(W)          # held until the launch is enqueued: for a non-leaf W the transpose is a temporary, and gemm() allocates
        #                     its split-K workspace before launching -- a freed W^T block could be handed out as that workspace
        gemm(_p(g), _p(Wt), _p(dx), M, K, N, lda=N, ldb=N, ldc=K, trans_b=True, **kw)
        del Wt
    else:
        gemm(_p(g), _p(W), _p(dx), M, K, N, lda=N, ldb=ldb, ldc=K, trans_b=False, **kw)


# ---- optional per-kernel timing (bench.py): HIP events on the launch stream around each GEMM ------
_events = None
_events_by_shape = False
_events_only = None
_plan_cache = {}


_events_stride, _events_seen = 1, {}


def _event_wanted(name: str) -> bool:
    if _events_only is not None and name != _events_only:
        return False
    n = _events_seen.get(name, 0)
    _events_seen[name] = n + 1
    # a hashed 1-in-stride sample: the launches of a kernel cycle through the network's shapes with a short period, and a plain
    # every-stride-th pick locks onto one phase of that cycle (it timed the largest shapes only: 43 us against rocprofv3's 25)
    return _events_stride <= 1 or ((n * 2654435761) >> 7) % _events_stride == 0


def kernel_events_begin(by_shape: bool = False, only: Optional[str] = None, stride: int = 1):
    """only: time the launches of ONE kernel name (as a previous full pass reported it); stride: time every stride-th of them.
    Two events per GEMM on every launch cost ~4 % of a step (and on a few hundred side-stream launches per step they made the HOST
    the limit of the step); two events on ~30 sampled launches of the dominant kernel cost nothing."""
    global _events, _events_by_shape, _events_only, _events_stride
    _events = []
    _events_by_shape = bool(by_shape)
    _events_only = only
    _events_stride = max(1, int(stride))
    _events_seen.clear()


def kernel_events_end():
    """-> {kernel name: {"n", "ms", "flop"}} for the launches since kernel_events_begin()."""
    global _events, _events_only
    ev, _events = _events, None
    _events_only = None
    if not ev:
        return {}
    torch.cuda.synchronize()
    table = {}
    ms = C.c_float()
    for name, flop, nbytes, e0, e1 in ev:
        r = table.setdefault(name, {"n": 0, "ms": 0.0, "flop": 0.0, "bytes": 0.0, "timer": "event pair around the launch"})
        r["n"] += 1
        if isinstance(e0, int):
            call("mdvit_event_elapsed_ms", e0, e1, C.byref(ms))
            r["ms"] += ms.value
            r["timer"] = "kernel begin/end timestamps (hipExtLaunchKernelGGL start/stop events)"
        else:
            r["ms"] += e0.elapsed_time(e1)
        r["flop"] += flop
        r["bytes"] += nbytes
    _lib_events_release()
    ovh = event_pair_overhead_ms()
    for name, r in table.items():
        r["launches"] = _events_seen.get(name, r["n"])          # all launches of that kernel since begin (n of them were timed)
        r["event_pair_overhead_ms"] = ovh if r["timer"].startswith("event pair") else 0.0
    return table


_lib_event_pool = []


def _lib_event() -> int:
    h = C.c_void_p()
    call("mdvit_event_create", C.byref(h))
    _lib_event_pool.append(h.value)
    return h.value


def _lib_events_release():
    while _lib_event_pool:
        call("mdvit_event_destroy", _lib_event_pool.pop())


def event_pair_overhead_ms(n: int = 32) -> float:
    """what an EMPTY start/end event pair measures on the idle current stream (median of n): the part of every timed launch that is
    the events themselves (~4.6 us on gfx950 / ROCm 7.2, tools/probe/event_overhead.py)"""
    torch.cuda.synchronize()
    pairs = []
    for _ in range(n):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); e1.record()
        pairs.append((e0, e1))
    torch.cuda.synchronize()
    v = sorted(a.elapsed_time(b) for a, b in pairs)
    return v[len(v) // 2]


def _ld_view(t: torch.Tensor) -> Tuple[int, int, int]:
    """(rows, cols, leading dim) of a 2-D tensor whose rows are contiguous (column slices allowed); a contiguous
    conv weight [out, in, kh, kw] is read as [out, in*kh*kw] (so leaf conv weights reach the op un-viewed)."""
    if t.dim() == 4 and t.is_contiguous():
        return t.shape[0], t.numel() // t.shape[0], t.numel() // t.shape[0]
    if t.dim() != 2 or t.stride(1) != 1:
        raise _lib.MdvitHipError("expected a 2-D row-contiguous tensor")
    return t.shape[0], t.shape[1], t.stride(0)


# ------------------------------------------------------------------------------------------------
# Linear / 1x1 conv:  y = x W^T + b  [+ dropout, DropPath row scale, residual]
# ------------------------------------------------------------------------------------------------
class _Linear(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, W, b, residual, rowscale, drop_p, rows_per_scale):
        ctx.set_materialize_grads(False)
        # x [M,K] ; W [N,K] (may be a column slice: stride(0) >= K) ; residual [M,N]
        _chk(x, b, residual, rowscale)
        M, K = x.shape
        N, K2, ldb = _ld_view(W)
        assert K == K2
        y = _empty((M, N), device=x.device, dtype=torch.float32)
        key = _next_key() if drop_p > 0 else (0, 0)
        if _lin_rc_ok(M, N, K) and ldb == K and (residual is not None or (drop_p == 0 and rowscale is None)):
            _linear_rc(x, W, False, b, y, M, N, K, drop_p, key, rowscale, rows_per_scale, residual)
        elif _plane_ok(M, N, K, residual is not None):
            gemm_nt(x, W, y, M, N, K, bias=b, e_drop=drop_p, e_key=key, e_rowscale=rowscale, e_rows_per_scale=rows_per_scale,
                    residual=residual, allow_split=True)
        else:
            gemm(_p(x), _p(W), _p(y), M, N, K, lda=K, ldb=ldb, ldc=N, bias=_p(b),
                 e_drop=drop_p, e_key=key, e_rowscale=_p(rowscale), e_rows_per_scale=rows_per_scale,
                 residual=_p(residual), ldr=N, allow_split=True)
        ctx.save_for_backward(x, W, rowscale)
        ctx.meta = (drop_p, key, rows_per_scale, b is not None, residual is not None)
        ctx.bias_ref = b if (b is not None and b.grad_fn is None) else None      # leaf bias: only to look up its gradient sink
        return y

    @staticmethod
    def backward(ctx, g):
        if g is None:
            return (None,) * 7
        x, W, rowscale = ctx.saved_tensors
        drop_p, key, rps, has_b, has_res = ctx.meta
        g = _c(g)
        M, K = x.shape
        N, _, ldb = _ld_view(W)
        dx = dW = db = None
        want_w = not _dgrad_only and (ctx.needs_input_grad[1] or (has_b and ctx.needs_input_grad[2]))
        sW = sb = None
        if want_w:
            sW = _sink_of(W)
            sb = _sink_of(ctx.bias_ref) if has_b else None
        sunk = want_w and sW is not None and (not has_b or sb is not None)
        want_b = want_w and has_b and (sunk or ctx.needs_input_grad[2])
        ride = want_b and (sunk or ctx.needs_input_grad[1])
        if want_b and not sunk:
            # (the bias gradient of a NON-leaf bias -- the peer heads' composed biases, 20 per step -- rides on the weight-gradient GEMM's column sums, which
            #  accumulate: it starts from a slice of a pre-zeroed slab instead of paying a fill launch of its own)
            db = _zeros_once(N, x.device) if ride else _empty((N,), device=x.device, dtype=torch.float32)
        # ONE pass over the upstream gradient: gm = g * dropmask * droppath scale (only if there is a mask) and the bias
        # gradient (column sums of gm); the dgrad / wgrad GEMMs below read gm with no prologue of their own
        masked = drop_p > 0 or rowscale is not None
        gm = _empty_like(g) if masked else g
        # the bias gradient (column sums of gm) rides on the wgrad GEMM's own pass over gm (colsum_a) whenever that GEMM runs: the
        # mask pass then only masks -- no partial sums, no second-stage reduction launch on the main stream
        if masked or (want_b and not ride):
            sums = want_b and not ride
            wsp, wsb, _keep = _partials_ws(N, g.device) if sums else (None, 0, None)
            call("mdvit_colsum_f32", _p(g), N, _p(sb if sunk else db) if sums else None, _p(gm) if masked else None, wsp, wsb, M, N,
                 drop_p, key[0], key[1], _p(rowscale), rps, int(sunk), _seed_ptr() if drop_p > 0 else None, _stream())
        if ctx.needs_input_grad[0]:
            dx = _empty_like(x)
            _dgrad(gm, W, dx, M, K, N, ldb, allow_split=True)
        if want_w:
            if sunk:
                # accumulate straight into the gradient buckets (side stream if enabled); autograd gets None
                with _on_side(gm, x):
                    gemm(_p(gm), _p(x), _p(sW), N, K, M, lda=N, ldb=K, ldc=K, trans_a=True, trans_b=False, allow_split=True, accumulate=True,
                         colsum_a=_p(sb) if ride else None)
                db = None
            elif ctx.needs_input_grad[1]:
                dW = _empty(tuple(W.shape) if W.dim() == 4 else (N, K), device=x.device, dtype=torch.float32)
                gemm(_p(gm), _p(x), _p(dW), N, K, M, lda=N, ldb=K, ldc=K, trans_a=True, trans_b=False, allow_split=True,
                     colsum_a=_p(db) if ride else None)
        return dx, dW, db, (g if has_res else None), None, None, None


def linear(x, W, b=None, residual=None, rowscale=None, drop_p: float = 0.0, rows_per_scale: int = 1):
    """x: [..., K] -> [..., N].  W [N,K] may be a view with W.stride(1)==1 (column slice of a wider matrix)."""
    shp = x.shape
    x2 = _c(x).view(-1, shp[-1])
    if W.dim() == 4 and not W.is_contiguous():
        W = W.reshape(W.shape[0], -1)
    r2 = None if residual is None else _c(residual).view(-1, W.shape[0])
    y = _Linear.apply(x2, W, b, r2, rowscale, float(drop_p), int(rows_per_scale))
    return y.view(*shp[:-1], W.shape[0])


class _MatMul(torch.autograd.Function):
    """C[M,N] = A[M,K] @ B[K,N] for weight composition (both operands row-contiguous 2-D views)."""

    @staticmethod
    def forward(ctx, A, B):
        ctx.set_materialize_grads(False)
        M, K, lda = _ld_view(A)
        K2, N, ldb = _ld_view(B)
        assert K == K2
        out = _empty((M, N), device=A.device, dtype=torch.float32)
        gemm(_p(A), _p(B), _p(out), M, N, K, lda=lda, ldb=ldb, ldc=N, trans_b=False, allow_split=True)
        ctx.save_for_backward(A, B)
        return out

    @staticmethod
    def backward(ctx, g):
        if g is None:
            return (None,) * 2
        if _dgrad_only:          # both operands are weights
            return None, None
        A, B = ctx.saved_tensors
        g = _c(g)
        M, K, lda = _ld_view(A)
        _, N, ldb = _ld_view(B)
        dA = dB = None
        if ctx.needs_input_grad[0]:      # dA = g @ B^T : B stored [K,N] row-major == "weight [N'=K, K'=N]"
            dA = _empty((M, K), device=A.device, dtype=torch.float32)
            gemm(_p(g), _p(B), _p(dA), M, K, N, lda=N, ldb=ldb, ldc=K, trans_b=True, allow_split=True)
        if ctx.needs_input_grad[1]:      # dB = A^T @ g
            dB = _empty((K, N), device=A.device, dtype=torch.float32)
            gemm(_p(A), _p(g), _p(dB), K, N, M, lda=lda, ldb=N, ldc=N, trans_a=True, trans_b=False, allow_split=True)
        return dA, dB


def matmul(A, B):
    return _MatMul.apply(A, B)


_zero_slab = [None, 0, None, None]        # a zero-filled slab, the offset of its first unused element, the event behind its fill, the stream that filled it


def _zeros_once(n: int, device):
    """n fresh zeros without a fill launch per call: slices of a 64K-float slab that is zero-filled once (one fill per ~100 calls); every slice is handed out
    ONCE.  A consumer on ANOTHER stream than the one that filled the slab waits for the fill's event (until it is seen complete).  Under HIP-graph capture a
    replay would see the slices dirty: the plain fill is captured instead."""
    if torch.cuda.is_current_stream_capturing() or n > 16384:
        return torch.zeros((n,), device=device, dtype=torch.float32)
    slab, off, ev, fs = _zero_slab
    n4 = (n + 3) & ~3
    cur = current_stream_obj()
    if slab is None or slab.device != device or off + n4 > slab.numel():
        slab, off = torch.zeros((65536,), device=device, dtype=torch.float32), 0
        ev = torch.cuda.Event()
        ev.record(cur)
        fs = cur.cuda_stream
    elif ev is not None:
        if ev.query():
            ev = None
        elif cur.cuda_stream != fs:
            cur.wait_event(ev)
    if slab is not None and cur.cuda_stream != fs:
        # a consumer on another stream than the filler's: the slab's block goes back to the FILLER's pool when its last slice dies, and nothing else orders that with this
        # stream's kernel that still writes the slice (peer / branch streams only)
        slab.record_stream(cur)
    _zero_slab[:] = [slab, off + n4, ev, fs]
    return slab[off:off + n]


def _vp_array(ptrs):
    return (C.c_void_p * len(ptrs))(*ptrs)


def gemm_grouped(As, Bs, outs, M, N, K, *, lda, ldb, ldc, trans_a=False, trans_b=True, accumulate=False, precision=None):
    """the same plain product on len(As) operand triples in ONE launch (mdvit_gemm_f32_grouped: no K split, no bias)"""
    if precision is None:
        precision = min(_gemm_precision, 1) if (bool(trans_b) != bool(trans_a)) else 0
    d = GemmDesc()
    d.lda, d.ldb, d.ldc = lda, ldb, ldc
    d.M, d.N, d.K = M, N, K
    d.trans_a, d.trans_b = int(trans_a), int(trans_b)
    d.accumulate = int(accumulate)
    d.precision = int(precision)
    call("mdvit_gemm_f32_grouped", C.byref(d), len(As), _vp_array([_p(t) for t in As]), _vp_array([_p(t) for t in Bs]), _vp_array([_p(t) for t in outs]), _stream())


def gemm_grouped_bias(As, Bs, outs, biases, M, N, K, *, lda, ldb, ldc, trans_b=True, precision=None):
    """gemm_grouped with a bias vector per group (NN / NT)"""
    if precision is None:
        precision = min(_gemm_precision, 1) if bool(trans_b) else 0
    d = GemmDesc()
    d.lda, d.ldb, d.ldc = lda, ldb, ldc
    d.M, d.N, d.K = M, N, K
    d.trans_a, d.trans_b = 0, int(trans_b)
    d.precision = int(precision)
    call("mdvit_gemm_f32_grouped_bias", C.byref(d), len(As), _vp_array([_p(t) for t in As]), _vp_array([_p(t) for t in Bs]), _vp_array([_p(t) for t in outs]),
         _vp_array([_p(t) for t in biases]), _stream())


class _LinearGrouped(torch.autograd.Function):
    """y_g = x_g W_g^T + b_g for G groups (same shapes, own weights: the peer heads' low-resolution 1x1 convolutions) -- forward ONE launch, data gradients ONE launch
    (blockIdx.z = group, per group the arithmetic of the single launch without a K split); weight / bias gradients per group as in _Linear."""

    @staticmethod
    def forward(ctx, G, *ts):
        ctx.set_materialize_grads(False)
        xs, Ws, bs = ts[:G], ts[G:2 * G], ts[2 * G:]
        _chk(*xs, *bs)
        M, K = xs[0].shape
        N, K2, ldb = _ld_view(Ws[0])
        assert K == K2 and all(tuple(x.shape) == (M, K) for x in xs) and all(_ld_view(W) == (N, K, ldb) for W in Ws)
        ys = [_empty((M, N), device=xs[0].device, dtype=torch.float32) for _ in range(G)]
        gemm_grouped_bias(xs, Ws, ys, bs, M, N, K, lda=K, ldb=ldb, ldc=N, trans_b=True)
        ctx.save_for_backward(*xs, *Ws)
        ctx.meta = (G, M, N, K, ldb)
        return tuple(ys)

    @staticmethod
    def backward(ctx, *gs):
        G, M, N, K, ldb = ctx.meta
        if all(g is None for g in gs):
            return (None,) * (1 + 3 * G)
        sv = ctx.saved_tensors
        xs, Ws = sv[:G], sv[G:]
        dev = xs[0].device
        gs = [_c(g) if g is not None else torch.zeros((M, N), device=dev, dtype=torch.float32) for g in gs]
        dxs = [None] * G
        if any(ctx.needs_input_grad[1:1 + G]):
            dxs = [_empty((M, K), device=dev, dtype=torch.float32) for _ in range(G)]
            if _gemm_precision:
                Wts = [wt(W) for W in Ws]           # (held until the launch is enqueued)
                gemm_grouped(gs, Wts, dxs, M, K, N, lda=N, ldb=N, ldc=K, trans_b=True)
                del Wts
            else:
                gemm_grouped(gs, list(Ws), dxs, M, K, N, lda=N, ldb=ldb, ldc=K, trans_b=False)
        dWs, dbs = [None] * G, [None] * G
        if not _dgrad_only:
            for i in range(G):
                if ctx.needs_input_grad[1 + G + i] or ctx.needs_input_grad[1 + 2 * G + i]:
                    dWs[i] = _empty((N, K), device=dev, dtype=torch.float32)
                    dbs[i] = _zeros_once(N, dev)
                    gemm(_p(gs[i]), _p(xs[i]), _p(dWs[i]), N, K, M, lda=N, ldb=K, ldc=K, trans_a=True, trans_b=False, allow_split=True, colsum_a=_p(dbs[i]))
        return (None, *dxs, *dWs, *dbs)


def linear_grouped(xs, Ws, bs):
    """xs[g]: [..., K] (equal shapes), Ws[g]: [N, K], bs[g]: [N]  ->  list of [..., N]"""
    shp = xs[0].shape
    x2 = [_c(x).view(-1, shp[-1]) for x in xs]
    ys = _LinearGrouped.apply(len(xs), *x2, *Ws, *bs)
    return [y.view(*shp[:-1], Ws[0].shape[0]) for y in ys]


class _ComposeHeads(torch.autograd.Function):
    """The weight composition of G peer heads x Q scales as grouped launches (decode.MLPDecoderFM: linear_fuse o cat o resize o linear_q evaluated as
    resize((Wf_q W_q) x_q + Wf_q b_q)):  Wc[g][q] = Wf[g][:, q*hid:(q+1)*hid] @ W[g][q],  bc[g][q] = Wf[g][:, q-block] . b[g][q].
    Inputs: G matrices Wf_g [hid, Q*hid] (row-contiguous 2-D views, e.g. the leading columns of the fuse weight), then G*Q weights W[g][q] [hid, C_q], then G*Q
    biases [hid].  Outputs: G*Q composed weights, then G*Q composed biases.  Q products forward and 2 Q backward, each ONE launch over the G heads
    (instead of G each, most with a split-K second stage), the bias parts ONE launch each way."""

    @staticmethod
    def forward(ctx, G, Q, *ts):
        ctx.set_materialize_grads(False)
        Wf, W, b = ts[:G], ts[G:G + G * Q], ts[G + G * Q:]
        _chk(*W, *b)                                            # (Wf: row-contiguous column-slice views; W: the 1x1 conv weights [hid, C_q, 1, 1] or [hid, C_q])
        hid, _, ldf = _ld_view(Wf[0])
        assert all(w.is_cuda and w.dtype == torch.float32 and _ld_view(w)[0] == hid and _ld_view(w)[2] == ldf and w.shape[1] == Q * hid for w in Wf)
        dev = Wf[0].device
        Wc, bc = [None] * (G * Q), [None] * (G * Q)
        for q in range(Q):
            Cq = W[q].numel() // hid
            for g in range(G):
                assert W[g * Q + q].shape[0] == hid and W[g * Q + q].numel() == hid * Cq
                Wc[g * Q + q] = _empty((hid, Cq), device=dev, dtype=torch.float32)
                bc[g * Q + q] = _empty((hid,), device=dev, dtype=torch.float32)
            gemm_grouped([Wf[g][:, q * hid:] for g in range(G)], [W[g * Q + q] for g in range(G)], [Wc[g * Q + q] for g in range(G)],
                         hid, Cq, hid, lda=ldf, ldb=Cq, ldc=Cq, trans_b=False)
        items = [(g, q) for g in range(G) for q in range(Q)]
        call("mdvit_compose_bias", len(items), _vp_array([_p(Wf[g][:, q * hid:]) for g, q in items]), ldf, _vp_array([_p(b[g * Q + q]) for g, q in items]),
             _vp_array([_p(bc[g * Q + q]) for g, q in items]), hid, hid, _stream())
        ctx.save_for_backward(*ts)
        ctx.meta = (G, Q, hid, ldf)
        return tuple(Wc) + tuple(bc)

    @staticmethod
    def backward(ctx, *gs):
        G, Q, hid, ldf = ctx.meta
        none = (None,) * (2 + G + 2 * G * Q)
        if _dgrad_only or all(g is None for g in gs):          # (all operands are weights: nothing to do in the data-gradient-only sweep)
            return none
        ts = ctx.saved_tensors
        Wf, W, b = ts[:G], ts[G:G + G * Q], ts[G + G * Q:]
        dev = Wf[0].device
        dWc = [None if g is None else _c(g) for g in gs[:G * Q]]
        dbc = [None if g is None else _c(g) for g in gs[G * Q:]]
        for i in range(G * Q):                                 # (a head whose output was not used hands None: treat as zeros)
            if dWc[i] is None:
                dWc[i] = torch.zeros((hid, W[i].numel() // hid), device=dev, dtype=torch.float32)
            if dbc[i] is None:
                dbc[i] = torch.zeros_like(b[i])
        dWf = [_empty((hid, Q * hid), device=dev, dtype=torch.float32) for _ in range(G)]
        sW = [_sink_of(w) for w in W]
        sb = [_sink_of(t) for t in b]
        sunk = all(t is not None for t in sW) and all(t is not None for t in sb)
        dW = list(sW) if sunk else [_empty_like(w) for w in W]
        db = list(sb) if sunk else [_empty_like(t) for t in b]
        items = [(g, q) for g in range(G) for q in range(Q)]
        A_blocks = _vp_array([_p(Wf[g][:, q * hid:]) for g, q in items])
        b_ptrs, dbc_ptrs = _vp_array([_p(b[g * Q + q]) for g, q in items]), _vp_array([_p(dbc[g * Q + q]) for g, q in items])
        for q in range(Q):
            Cq = W[q].numel() // hid
            # dWf[g][:, q-block] = dWc[g][q] @ W[g][q]^T          (NT: W[g][q] is the "weight [N = hid, K = C_q]")
            gemm_grouped([dWc[g * Q + q] for g in range(G)], [W[g * Q + q] for g in range(G)], [dWf[g][:, q * hid:] for g in range(G)],
                         hid, hid, Cq, lda=Cq, ldb=Cq, ldc=Q * hid, trans_b=True)
        # dWf[g][:, q-block] += dbc[g][q] (x) b[g][q]
        call("mdvit_compose_bias_bwd", len(items), A_blocks, ldf, b_ptrs, dbc_ptrs, _vp_array([_p(dWf[g][:, q * hid:]) for g, q in items]), Q * hid,
             None, 0, hid, hid, _stream())

        def param_grads():
            for q in range(Q):
                Cq = W[q].numel() // hid
                # dW[g][q] (+)= Wf[g][:, q-block]^T @ dWc[g][q]
                gemm_grouped([Wf[g][:, q * hid:] for g in range(G)], [dWc[g * Q + q] for g in range(G)], [dW[g * Q + q] for g in range(G)],
                             hid, Cq, hid, lda=ldf, ldb=Cq, ldc=Cq, trans_a=True, trans_b=False, accumulate=sunk)
            # db[g][q] (+)= Wf[g][:, q-block]^T dbc[g][q]
            call("mdvit_compose_bias_bwd", len(items), A_blocks, ldf, b_ptrs, dbc_ptrs, None, 0, _vp_array([_p(db[g * Q + q]) for g, q in items]), int(sunk),
                 hid, hid, _stream())

        if sunk:
            # into the gradient buckets, like every other sunk weight gradient: on the weight-gradient stream (joined before the buckets are read);
            # a sink written on whatever stream this node runs on -- a peer stream, the sweep stream -- is ordered with nothing
            with _on_side(*dWc, *dbc, foreign="all"):      # all 2 G Q of them are upstream gradients of other nodes (with peer streams: other streams' pools)
                param_grads()
        else:
            param_grads()
        if sunk:
            return (None, None) + tuple(dWf) + (None,) * (2 * G * Q)
        return (None, None) + tuple(dWf) + tuple(dW) + tuple(db)


def compose_heads(Wf_list, W_lists, b_lists):
    """Wf_list[g]: [hid, Q*hid]; W_lists[g][q]: [hid, C_q] (or the 1x1 conv weight [hid, C_q, 1, 1] itself: a leaf finds its gradient sink); b_lists[g][q]: [hid]
    ->  (Wc[g][q], bc[g][q]) nested lists"""
    G, Q = len(Wf_list), len(W_lists[0])
    flatW = [W_lists[g][q] for g in range(G) for q in range(Q)]
    flatb = [b_lists[g][q] for g in range(G) for q in range(Q)]
    out = _ComposeHeads.apply(G, Q, *Wf_list, *flatW, *flatb)
    Wc, bc = out[:G * Q], out[G * Q:]
    return [[(Wc[g * Q + q], bc[g * Q + q]) for q in range(Q)] for g in range(G)]


# ------------------------------------------------------------------------------------------------
# MLP with residual:  out = res + DropPath(Dropout(fc2(Dropout(GELU(fc1(x))))))      mpvit.py:71-78
# ------------------------------------------------------------------------------------------------
_mlp_recompute = os.environ.get("MDVIT_MLP_RECOMPUTE", "1") != "0"
_mlp_fused = os.environ.get("MDVIT_MLP_FUSED", "1") != "0"
_mlp_recompute_maxc = int(os.environ.get("MDVIT_MLP_RECOMPUTE_MAXC", "128"))
_mlp_rc = os.environ.get("MDVIT_MLP_RC", "1") != "0"      # csrc/mlp_rc.hip: no [tokens, hidden] tensor in HBM in either pass (0: round 2's kernels, A/B)
_mlp_rc16 = os.environ.get("MDVIT_MLP_RC16", "1") != "0"  # C = 128: the backward data path in one kernel on 16-token waves (0: the two data-gradient GEMMs, A/B)
# C = 64, full sweep: data AND weight gradients from one evaluation of u, d and the activation (mdvit_mlp_rc_bwd) instead of mlp_rc_dgrad + mlp_rc_wgrad, which
# recompute them twice.  "1" (default): always -- measured in the three-stream step against the two kernels (the weight-gradient kernel on the side stream): parity mode
# 474 = 474 images/s at bs=4, 557 against 553 at bs=32; bf16 mode 520 against 517 at bs=4, 614 against 601 at bs=16.  "auto": only where the weight gradients would not
# run on a side stream; "0": never (A/B)
_mlp_rc_bwd = os.environ.get("MDVIT_MLP_RC_BWD", "1")
_lib.on_load(lambda lib: lib.mdvit_block_config({"0": 0, "1": 1}.get(_mlp_rc_bwd, 2)))          # the C-level block entry follows the same switch (applied when the library loads)


def _mlp_rc_ok(Cin, Hd, b1, b2, res, W1, W2, M) -> bool:
    return (_mlp_rc and _mlp_recompute and _gemm_precision >= 1 and Cin == 64 and Hd % 256 == 0 and Hd <= 4096 and b1 is not None and b2 is not None and res is not None
            and W1.is_contiguous() and W2.is_contiguous() and M * Hd < (1 << 32))


def _mlp_rc16_ok(Cin, Hd, b1, W1, W2, M) -> bool:
    return (_mlp_rc16 and _gemm_precision >= 1 and Cin == 128 and Hd % 32 == 0 and Hd <= 4096 and b1 is not None and W1.is_contiguous() and W2.is_contiguous()
            and M * Hd < (1 << 32))


class _MlpResidual(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, res, W1, b1, W2, b2, rowscale, drop_p, rows_per_scale):
        ctx.set_materialize_grads(False)
        _chk(x, res, W1, b1, W2, b2, rowscale)
        M, Cin = x.shape
        Hd = W1.shape[0]
        k1 = _next_key() if drop_p > 0 else (0, 0)
        k2 = _next_key() if drop_p > 0 else (0, 0)
        if _mlp_rc_ok(Cin, Hd, b1, b2, res, W1, W2, M):
            # C = 64: the hidden activation stays in registers; NOTHING of size [tokens, hidden] is kept for the backward (it recomputes)
            out = _empty((M, Cin), device=x.device, dtype=torch.float32)
            W1p, W2p = _wplanes(W1, False, 2), _wplanes(W2, False, 2)
            call("mdvit_mlp_rc_fwd", _p(x), _p(W1p), _p(b1), _p(W2p), _p(b2), _p(res), _p(rowscale), rows_per_scale, _p(out), M, Cin, Hd,
                 drop_p, k1[0], k1[1], k2[0], k2[1], _seed_ptr() if drop_p > 0 else None, _stream())
            del W1p, W2p
            ctx.save_for_backward(x, None, None, W1, W2, rowscale)
            ctx.meta = (drop_p, k1, k2, rows_per_scale)
            ctx.b1_ref, ctx.b2_ref = b1, b2
            ctx.rc = True
            return out
        ctx.rc = False
        h = _empty((M, Hd), device=x.device, dtype=torch.float32)
        # HBM-bound MLPs (fc1's K = C <= 128): keep gelu(u) only; the backward recomputes the pre-activation u inside the fc2
        # data-gradient GEMM (one more K = C product per tile) instead of moving [tokens, hidden] u through HBM twice
        if _mlp_recompute and _mlp_fused and _gemm_precision >= 1 and Cin == 64 and Hd % 64 == 0 and b1 is not None and b2 is not None \
                and res is not None and W1.is_contiguous() and W2.is_contiguous() and M * Hd < (1 << 32):
            # C = 64: both GEMMs in ONE kernel -- h goes to HBM once (for the backward) and, through LDS, straight into fc2
            out = _empty((M, Cin), device=x.device, dtype=torch.float32)
            call("mdvit_mlp_fwd_f32", _p(x), _p(W1), _p(b1), _p(W2), _p(b2), _p(res), _p(rowscale), rows_per_scale, _p(h), _p(out), M, Cin, Hd,
                 drop_p, k1[0], k1[1], k2[0], k2[1], _seed_ptr() if drop_p > 0 else None, _stream())
            ctx.save_for_backward(x, None, h, W1, W2, rowscale)
            ctx.meta = (drop_p, k1, k2, rows_per_scale)
            ctx.b1_ref, ctx.b2_ref = b1, b2
            return out
        if _mlp_recompute and _mlp_recompute_maxc >= 128 and b2 is not None and res is not None and _mlp_rc16_ok(Cin, Hd, b1, W1, W2, M):
            # C = 128: both GEMMs in ONE kernel on 16-token waves, the hidden chunk chained in registers; h goes to HBM once, for the fc2 weight gradient
            out = _empty((M, Cin), device=x.device, dtype=torch.float32)
            W1p, W2p = _wplanes(W1, False, 2), _wplanes(W2, False, 2)
            call("mdvit_mlp_rc16_fwd", _p(x), _p(W1p), _p(b1), _p(W2p), _p(b2), _p(res), _p(rowscale), rows_per_scale, _p(h), _p(out), M, Cin, Hd,
                 drop_p, k1[0], k1[1], k2[0], k2[1], _seed_ptr() if drop_p > 0 else None, _stream())
            del W1p, W2p
            ctx.save_for_backward(x, None, h, W1, W2, rowscale)
            ctx.meta = (drop_p, k1, k2, rows_per_scale)
            ctx.b1_ref, ctx.b2_ref = b1, b2
            return out
        plane = _plane_ok(M, Hd, Cin) and W1.is_contiguous()
        plane2 = _plane_ok(M, Cin, Hd, res is not None) and W2.is_contiguous()
        if _mlp_recompute and _gemm_precision >= 1 and Cin <= _mlp_recompute_maxc and Cin % 32 == 0 and Hd % 4 == 0:
            u = None
            if plane:
                gemm_nt(x, W1, h, M, Hd, Cin, bias=b1, epi=_lib.EPI_GELU_DUAL, e_drop=drop_p, e_key=k1)
            else:
                gemm(_p(x), _p(W1), _p(h), M, Hd, Cin, lda=Cin, ldb=Cin, ldc=Hd, bias=_p(b1), epi=_lib.EPI_GELU_DUAL, e_drop=drop_p, e_key=k1)
        else:
            u = _empty_like(h)
            if plane:
                gemm_nt(x, W1, h, M, Hd, Cin, bias=b1, U=u, epi=_lib.EPI_GELU_DUAL, e_drop=drop_p, e_key=k1)
            else:
                gemm(_p(x), _p(W1), _p(u), M, Hd, Cin, lda=Cin, ldb=Cin, ldc=Hd, bias=_p(b1), out2=_p(h),
                     epi=_lib.EPI_GELU_DUAL, e_drop=drop_p, e_key=k1)
        out = _empty((M, Cin), device=x.device, dtype=torch.float32)
        if plane2:
            gemm_nt(h, W2, out, M, Cin, Hd, bias=b2, e_drop=drop_p, e_key=k2, e_rowscale=rowscale, e_rows_per_scale=rows_per_scale, residual=res)
        else:
            gemm(_p(h), _p(W2), _p(out), M, Cin, Hd, lda=Hd, ldb=Hd, ldc=Cin, bias=_p(b2),
                 e_drop=drop_p, e_key=k2, e_rowscale=_p(rowscale), e_rows_per_scale=rows_per_scale, residual=_p(res), ldr=Cin)
        ctx.save_for_backward(x, u, h, W1, W2, rowscale)
        ctx.meta = (drop_p, k1, k2, rows_per_scale)
        ctx.b1_ref, ctx.b2_ref = b1, b2           # leaf biases: only to look up their gradient sinks
        return out

    @staticmethod
    def backward(ctx, g):
        if g is None:
            return (None,) * 9
        x, u, h, W1, W2, rowscale = ctx.saved_tensors
        drop_p, k1, k2, rps = ctx.meta
        g = _c(g)
        M, Cin = x.shape
        Hd = W1.shape[0]
        dev = x.device
        dW1 = db1 = dW2 = db2 = None
        sinks = [_sink_of(t) for t in (W1, ctx.b1_ref, W2, ctx.b2_ref)] if not _dgrad_only else [None] * 4
        sunk = not _dgrad_only and all(t is not None for t in sinks)
        if ctx.rc:
            # gm = g * dropmask2 * droppath scale (+ db2 = its column sums, when parameter gradients are wanted) in one pass
            masked = drop_p > 0 or rowscale is not None
            want_w = not _dgrad_only
            gm = _empty_like(g) if masked else g
            if want_w and not sunk:
                db2 = _empty((Cin,), device=dev, dtype=torch.float32)
            if masked or want_w:
                wsp, wsb, _keep = _partials_ws(Cin, dev) if want_w else (None, 0, None)
                call("mdvit_colsum_f32", _p(g), Cin, _p(sinks[3] if sunk else db2) if want_w else None, _p(gm) if masked else None, wsp, wsb, M, Cin,
                     drop_p, k2[0], k2[1], _p(rowscale), rps, int(sunk), _seed_ptr() if drop_p > 0 else None, _stream())
            W1p, W2tp, W1tp = _wplanes(W1, False, 2), _wplanes(W2, True, 2), _wplanes(W1, True, 2)
            fused = want_w and Hd in (256, 512) and (_mlp_rc_bwd == "1" or (_mlp_rc_bwd == "auto" and not (sunk and _side_stream is not None)))
            if fused:
                roles = Hd // 256
                parts = _empty((roles, M, Cin), device=dev, dtype=torch.float32)
                wsb = _lib.load().mdvit_mlp_rc_wgrad_ws_bytes(M, Cin, Hd)
                ws = _empty((wsb // 4,), device=dev, dtype=torch.float32)
                if sunk:
                    tgt, accf = (sinks[0], sinks[1], sinks[2]), 1
                    db2 = None
                else:
                    dW1, db1, dW2 = _empty_like(W1), _empty((Hd,), device=dev, dtype=torch.float32), _empty_like(W2)
                    tgt, accf = (dW1, db1, dW2), 0
                call("mdvit_mlp_rc_bwd", _p(gm), _p(x), _p(W1p), _p(ctx.b1_ref), _p(W2tp), _p(W1tp), _p(parts), _p(tgt[0]), _p(tgt[1]), _p(tgt[2]), _p(ws), wsb,
                     M, Cin, Hd, drop_p, k1[0], k1[1], _seed_ptr() if drop_p > 0 else None, accf, _stream())
                if roles == 1:
                    dx = parts[0]
                else:
                    dx = _empty_like(x)
                    call("mdvit_sum_batch", _p(parts), _p(dx), roles, M * Cin, _stream())
                del W1p, W2tp, W1tp
                return dx, g, dW1, db1, dW2, db2, None, None, None
            dx = _empty_like(x)
            call("mdvit_mlp_rc_dgrad", _p(gm), _p(x), _p(W1p), _p(ctx.b1_ref), _p(W2tp), _p(W1tp), _p(dx), M, Cin, Hd,
                 drop_p, k1[0], k1[1], _seed_ptr() if drop_p > 0 else None, _stream())
            if want_w:
                wsb = _lib.load().mdvit_mlp_rc_wgrad_ws_bytes(M, Cin, Hd)
                if sunk:
                    with _on_side(gm, x, W1p, W2tp):
                        ws = _empty((wsb // 4,), device=dev, dtype=torch.float32)     # allocated on the side stream: freed in its order, nothing to protect
                        call("mdvit_mlp_rc_wgrad", _p(gm), _p(x), _p(W1p), _p(ctx.b1_ref), _p(W2tp), _p(sinks[0]), _p(sinks[1]), _p(sinks[2]), _p(ws), wsb,
                             M, Cin, Hd, drop_p, k1[0], k1[1], _seed_ptr() if drop_p > 0 else None, 1, _stream())
                    db2 = None
                else:
                    dW1, db1, dW2 = _empty_like(W1), _empty((Hd,), device=dev, dtype=torch.float32), _empty_like(W2)
                    ws = _empty((wsb // 4,), device=dev, dtype=torch.float32)
                    call("mdvit_mlp_rc_wgrad", _p(gm), _p(x), _p(W1p), _p(ctx.b1_ref), _p(W2tp), _p(dW1), _p(db1), _p(dW2), _p(ws), wsb,
                         M, Cin, Hd, drop_p, k1[0], k1[1], _seed_ptr() if drop_p > 0 else None, 0, _stream())
            del W1p, W2tp, W1tp
            return dx, g, dW1, db1, dW2, db2, None, None, None
        if not _dgrad_only and not sunk:
            db2 = torch.zeros((Cin,), device=dev, dtype=torch.float32)
        # gm = g * dropmask2 * droppath scale in one pass; db2 = column sums of gm rides on the fc2 wgrad GEMM's pass over gm (colsum_a)
        masked = drop_p > 0 or rowscale is not None
        gm = _empty_like(g) if masked else g
        if masked:
            call("mdvit_colsum_f32", _p(g), Cin, None, _p(gm), None, 0, M, Cin,
                 drop_p, k2[0], k2[1], _p(rowscale), rps, 0, _seed_ptr() if drop_p > 0 else None, _stream())
        # du = (gm W2) * gelu'(u) * mask1
        dx = _empty_like(x)
        if u is None and _mlp_fused and _gemm_precision >= 1 and Cin == 64 and Hd % 64 == 0 and W1.is_contiguous() and W2.is_contiguous():
            # C = 64: the whole data path in one kernel; the hidden-layer gradient is materialised only for the weight gradients
            du = None if _dgrad_only else _empty_like(h)
            W2t, W1t = wt(W2), wt(W1)
            call("mdvit_mlp_bwd_dgrad_f32", _p(gm), _p(x), _p(W1), _p(ctx.b1_ref), _p(W2t), _p(W1t), _p(du), _p(dx), M, Cin, Hd,
                 drop_p, k1[0], k1[1], _seed_ptr() if drop_p > 0 else None, _stream())
            del W2t, W1t
        elif u is None and _mlp_rc16_ok(Cin, Hd, ctx.b1_ref, W1, W2, M):
            # C = 128: u recomputed, du = (gm W2) * gelu'(u) * mask1 and dx = du W1 in ONE kernel; du reaches HBM only for the weight gradients
            du = None if _dgrad_only else _empty_like(h)
            W1p, W2tp, W1tp = _wplanes(W1, False, 2), _wplanes(W2, True, 2), _wplanes(W1, True, 2)
            call("mdvit_mlp_rc16_dgrad", _p(gm), _p(x), _p(W1p), _p(ctx.b1_ref), _p(W2tp), _p(W1tp), _p(du), _p(dx), M, Cin, Hd,
                 drop_p, k1[0], k1[1], _seed_ptr() if drop_p > 0 else None, _stream())
            del W1p, W2tp, W1tp
        else:
            du = _empty_like(h)
            if u is None and (_plane_rc or _gemm_precision == 2) and _use_plane_gemm and Cin % 32 == 0 and W1.is_contiguous() and W2.is_contiguous():
                # the plane kernel's recompute epilogue takes every operand as planes: x and gm are split here (two small passes)
                gemm_nt(to_planes(gm), W2, du, M, Hd, Cin, w_transposed=True, epi=_lib.EPI_DGELU, rc=(to_planes(x), W1, ctx.b1_ref, Cin),
                        e_drop=drop_p, e_key=k1)
            elif u is None:
                _dgrad(gm, W2, du, M, Hd, Cin, Hd, epi=_lib.EPI_DGELU, rc=(_p(x), Cin, _p(W1), Cin, _p(ctx.b1_ref), Cin), e_drop=drop_p, e_key=k1)
            else:
                _dgrad(gm, W2, du, M, Hd, Cin, Hd, epi=_lib.EPI_DGELU, gelu_u=_p(u), ldu=Hd, e_drop=drop_p, e_key=k1)
            _dgrad(du, W1, dx, M, Cin, Hd, Cin, allow_split=True)
        if not _dgrad_only:
            if sunk:
                dW1_, db1_, dW2_, _ = sinks
                with _on_side(gm, h, du, x):
                    gemm(_p(gm), _p(h), _p(dW2_), Cin, Hd, M, lda=Cin, ldb=Hd, ldc=Hd, trans_a=True, trans_b=False, allow_split=True, accumulate=True,
                         colsum_a=_p(sinks[3]))      # db2 = column sums of gm
                    gemm(_p(du), _p(x), _p(dW1_), Hd, Cin, M, lda=Hd, ldb=Cin, ldc=Cin, trans_a=True, trans_b=False, allow_split=True, accumulate=True,
                         colsum_a=_p(db1_))          # db1 = column sums of du, taken from the wgrad's A stream
                db2 = None
            else:
                dW2 = _empty_like(W2)
                gemm(_p(gm), _p(h), _p(dW2), Cin, Hd, M, lda=Cin, ldb=Hd, ldc=Hd, trans_a=True, trans_b=False, allow_split=True, colsum_a=_p(db2))
                dW1 = _empty_like(W1)
                db1 = torch.zeros((Hd,), device=dev, dtype=torch.float32)
                gemm(_p(du), _p(x), _p(dW1), Hd, Cin, M, lda=Hd, ldb=Cin, ldc=Cin, trans_a=True, trans_b=False, allow_split=True, colsum_a=_p(db1))
        return dx, g, dW1, db1, dW2, db2, None, None, None


def mlp_residual(x, res, W1, b1, W2, b2, rowscale=None, drop_p=0.0, rows_per_scale=1):
    shp = res.shape
    out = _MlpResidual.apply(_c(x).view(-1, shp[-1]), _c(res).view(-1, shp[-1]), W1, b1, W2, b2, rowscale,
                             float(drop_p), int(rows_per_scale))
    return out.view(shp)


# ------------------------------------------------------------------------------------------------
# LayerNorm
# ------------------------------------------------------------------------------------------------
class _LayerNorm(torch.autograd.Function):
    """fork=True returns (LN(x), x): the second output is x itself, handed to the residual consumer, so that the
    gradient of the residual branch comes back HERE and is added inside the LayerNorm backward kernel (otherwise
    autograd sums the two branches of  x + f(LN(x))  with an extra full-tensor add per block and sweep)."""

    @staticmethod
    def forward(ctx, x, gamma, beta, eps, fork):
        ctx.set_materialize_grads(False)
        _chk(x, gamma, beta)
        M, Cn = x.shape
        groups = gamma.shape[0] if gamma.dim() == 2 else 1       # [G, C]: parameter row g for the g-th of G equal row groups
        if M % groups:
            raise ValueError(f"layer_norm: {M} rows are not {groups} equal groups")
        y = _empty_like(x)
        mean = _empty((M,), device=x.device, dtype=torch.float32)
        rstd = _empty_like(mean)
        call("mdvit_layernorm_fwd", _p(x), _p(gamma), _p(beta), _p(y), _p(mean), _p(rstd), M, Cn, groups, eps, _stream())
        ctx.save_for_backward(x, gamma, mean, rstd)
        if fork:
            return y, x.view_as(x)
        return y

    @staticmethod
    def backward(ctx, g, g_pass=None):
        if g is None:
            return g_pass, None, None, None, None
        x, gamma, mean, rstd = ctx.saved_tensors
        g = _c(g)
        M, Cn = x.shape
        dx = _empty_like(x)
        gp = None if g_pass is None else _c(g_pass)
        fast = Cn in (64, 128, 320, 512)
        if _dgrad_only and fast:          # the aux sweep: no parameter gradients -> no partial sums, no reduction launch
            call("mdvit_layernorm_bwd", _p(g), _p(x), _p(gamma), _p(mean), _p(rstd), _p(gp), _p(dx), None, None, None, 0, M, Cn,
                 gamma.shape[0] if gamma.dim() == 2 else 1, _stream())
            return dx, None, None, None, None
        dg, db = _flat_like(gamma, gamma)
        wsp, wsb, _keep = _partials_ws(2 * Cn, x.device)
        call("mdvit_layernorm_bwd", _p(g), _p(x), _p(gamma), _p(mean), _p(rstd), _p(gp), _p(dx), _p(dg), _p(db), wsp, wsb, M, Cn,
             gamma.shape[0] if gamma.dim() == 2 else 1, _stream())
        if _dgrad_only:
            return dx, None, None, None, None
        return dx, dg, db, None, None


def layer_norm(x, gamma, beta, eps=1e-6):
    """gamma/beta [C], or [G, C] for G equal consecutive row groups with their own parameters (domain-specific norms)."""
    shp = x.shape
    return _LayerNorm.apply(_c(x).view(-1, shp[-1]), _c(gamma), _c(beta), float(eps), False).view(shp)


def layer_norm_fork(x, gamma, beta, eps=1e-6):
    """-> (LN(x), x_res): use x_res (same values as x) for the residual branch; see _LayerNorm."""
    shp = x.shape
    y, xr = _LayerNorm.apply(_c(x).view(-1, shp[-1]), _c(gamma), _c(beta), float(eps), True)
    return y.view(shp), xr.view(shp)


# ------------------------------------------------------------------------------------------------
# depthwise / grouped 3x3 convolutions (NHWC)
# ------------------------------------------------------------------------------------------------
class _DwConv3x3(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, w, bias, stride, add_input):
        ctx.set_materialize_grads(False)
        _chk(x, w, bias)
        B, H, W_, Cn = x.shape
        Ho, Wo = (H - 1) // stride + 1, (W_ - 1) // stride + 1
        y = _empty((B, Ho, Wo, Cn), device=x.device, dtype=torch.float32)
        call("mdvit_dwconv3x3_fwd", _p(x), _p(w), _p(bias), _p(y), B, H, W_, Cn, stride, int(add_input), _stream())
        ctx.save_for_backward(x, w)
        ctx.meta = (stride, add_input, bias is not None)
        ctx.bias_ref = bias if (bias is not None and bias.grad_fn is None) else None
        return y

    @staticmethod
    def backward(ctx, g):
        if g is None:
            return (None,) * 5
        x, w = ctx.saved_tensors
        stride, add_input, has_b = ctx.meta
        g = _c(g)
        B, H, W_, Cn = x.shape
        dx = _empty_like(x) if ctx.needs_input_grad[0] else None
        dw = db = None
        if dx is not None:
            call("mdvit_dwconv3x3_bwd", _p(g), _p(x), _p(w), _p(dx), None, None, None, 0, B, H, W_, Cn, stride, int(add_input), 0, _stream())
        if not _dgrad_only:
            sw = _sink_of(w) if _side_stream is not None else None
            sb = _sink_of(ctx.bias_ref) if (has_b and sw is not None) else None
            if sw is not None and (not has_b or sb is not None):
                # weight gradient on the side stream, accumulated straight into the gradient buckets
                with _on_side(g, x) as _:
                    wsp, wsb, keep = _partials_ws(10 * Cn, x.device)
                    call("mdvit_dwconv3x3_bwd", _p(g), _p(x), _p(w), None, _p(sw), _p(sb), wsp, wsb, B, H, W_, Cn, stride, int(add_input), 1, _stream())
            else:
                dw, db = _flat_like(w, (Cn,) if has_b else None)
                wsp, wsb, _keep = _partials_ws(10 * Cn, x.device)
                call("mdvit_dwconv3x3_bwd", _p(g), _p(x), _p(w), None, _p(dw), _p(db), wsp, wsb, B, H, W_, Cn, stride, int(add_input), 0, _stream())
        return dx, dw, db, None, None


def dwconv3x3(x, w, bias=None, stride=1, add_input=False):
    return _DwConv3x3.apply(_c(x), w, bias, int(stride), bool(add_input))


class _GConv2(torch.autograd.Function):
    @staticmethod
    def forward(ctx, skip, up, w):
        ctx.set_materialize_grads(False)
        _chk(skip, up, w)
        B, H, W_, Cn = skip.shape
        y = _empty_like(skip)
        call("mdvit_gconv2_3x3_fwd", _p(skip), _p(up), _p(w), _p(y), B, H, W_, Cn, _stream())
        ctx.save_for_backward(skip, up, w)
        return y

    @staticmethod
    def backward(ctx, g):
        if g is None:
            return (None,) * 3
        skip, up, w = ctx.saved_tensors
        g = _c(g)
        B, H, W_, Cn = skip.shape
        dskip, dup = _empty_like(skip), _empty_like(up)
        sw = _sink_of(w) if (_side_stream is not None and not _dgrad_only) else None
        dw = None if (_dgrad_only or sw is not None) else _empty_like(w)
        wsp, wsb, _keep = _partials_ws(18 * Cn, g.device) if dw is not None else (None, 0, None)
        call("mdvit_gconv2_3x3_bwd", _p(g), _p(skip), _p(up), _p(w), _p(dskip), _p(dup), _p(dw), wsp, wsb, B, H, W_, Cn, 0, _stream())
        if sw is not None:      # weight gradient on the side stream, into the gradient bucket
            with _on_side(g, skip, up):
                wsp, wsb, keep = _partials_ws(18 * Cn, g.device)
                call("mdvit_gconv2_3x3_bwd", _p(g), _p(skip), _p(up), _p(w), None, None, _p(sw), wsp, wsb, B, H, W_, Cn, 1, _stream())
        return dskip, dup, dw


def gconv2_3x3(skip, up, w):
    return _GConv2.apply(_c(skip), _c(up), w)


# ------------------------------------------------------------------------------------------------
# dense 3x3 conv = im2col + GEMM; stem.0 direct
# ------------------------------------------------------------------------------------------------
class _Im2col(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, stride, dilation=1):
        ctx.set_materialize_grads(False)
        _chk(x)
        B, H, W_, Cn = x.shape
        Ho, Wo = (H - 1) // stride + 1, (W_ - 1) // stride + 1
        col = _empty((B * Ho * Wo, Cn * 9), device=x.device, dtype=torch.float32)
        call("mdvit_im2col3x3", _p(x), _p(col), B, H, W_, Cn, stride, dilation, _stream())
        ctx.meta = (B, H, W_, Cn, stride, dilation)
        return col

    @staticmethod
    def backward(ctx, g):
        if g is None:
            return (None,) * 3
        B, H, W_, Cn, stride, dilation = ctx.meta
        g = _c(g)
        dx = _empty((B, H, W_, Cn), device=g.device, dtype=torch.float32)
        call("mdvit_col2im3x3", _p(g), _p(dx), B, H, W_, Cn, stride, dilation, _stream())
        return dx, None, None


_conv_w_cache = {}      # (id(weight), mode) -> (weakref, version tag, relaid-out copy)
_use_implicit_conv = os.environ.get("MDVIT_IMPLICIT_CONV", "1") != "0"      # A/B switch: 0 = im2col + GEMM everywhere


_conv_w_table = None     # (signature, device int64 table [n,5], blocks per item)


def refresh_conv_weights():
    """Rebuild EVERY cached implicit-convolution weight layout in one launch (called with refresh_transposes at the start of a step):
    TransFuse holds ~65 dense 3x3 convolutions x two layouts, each of which would otherwise pay its own launch on first use."""
    global _conv_w_table
    entries = []
    for key, hit in list(_conv_w_cache.items()):
        w = hit[0]()
        if w is not None and hit[1][2] == w.data_ptr() and key[1] in (0, 1):
            entries.append((key, w, hit))
    if not entries:
        return
    # unconditional, like refresh_transposes(): a replayed HIP graph updates the weights without moving any host-side tag
    sig = tuple((w.data_ptr(), h[2].data_ptr(), w.shape[0], w.shape[1], key[1]) for key, w, h in entries)
    if _conv_w_table is None or _conv_w_table[0] != sig:
        if torch.cuda.is_current_stream_capturing():
            return                            # host-built table: the warm-up steps before a capture build it
        dev = entries[0][1].device
        blocks = max((w.shape[0] * w.shape[1] * 9 + 2303) // 2304 for _, w, _ in entries)          # tiles of the largest weight (2304 floats each in both modes)
        _conv_w_table = (sig, torch.tensor([list(r) for r in sig], dtype=torch.int64, device=dev), int(min(blocks, 256)))
    call("mdvit_conv_weight_relayout_many", _p(_conv_w_table[1]), len(entries), _conv_w_table[2], _stream())
    for key, w, hit in entries:
        _conv_w_cache[key] = (hit[0], (w._version, _weights_epoch, w.data_ptr()), hit[2])


def _conv_weight(w, mode: int):
    """w [Cout, Cin, 3, 3] in the layout the implicit convolution reads (mdvit_conv_weight_relayout), cached per leaf weight and
    rebuilt when the weight changed (version counter / optimizer epoch)."""
    Cout, Cin = w.shape[0], w.shape[1]
    tag = (w._version, _weights_epoch, w.data_ptr())
    key = (id(w), mode)
    hit = _conv_w_cache.get(key)
    if hit is not None and hit[0]() is w and hit[1] == tag:
        return hit[2]
    out = hit[2] if (hit is not None and hit[0]() is w) else torch.empty(((Cout, 9 * Cin) if mode == 0 else (Cin, 9 * Cout)), device=w.device, dtype=torch.float32)
    call("mdvit_conv_weight_relayout", _p(w), _p(out), Cout, Cin, mode, _stream())
    if w.grad_fn is None:
        global _cache_filled
        _cache_filled = True
        _conv_w_cache[key] = (weakref.ref(w, lambda _r, key=key: _conv_w_cache.pop(key, None)), tag, out)
    return out


class _Conv3x3(torch.autograd.Function):
    """Dense 3x3 convolution (padding = dilation) as an IMPLICIT GEMM: the kernel gathers the tap-shifted pixel rows while it stages
    its A operand, so the [B Ho Wo, 9 Cin] im2col matrix is never written or read (forward, and the data gradient at stride 1 = the
    same call on dy with the flipped / transposed weight; the weight gradient = the TN kernel with the image as its gathered B
    operand, on the side stream).  Only the strided data gradient still goes through dcol + col2im."""

    @staticmethod
    def forward(ctx, x, w, bias, stride, dilation):
        ctx.set_materialize_grads(False)
        _chk(x, w, bias)
        B, H, W_, Cin = x.shape
        Cout = w.shape[0]
        Ho, Wo = (H - 1) // stride + 1, (W_ - 1) // stride + 1
        y = _empty((B, Ho, Wo, Cout), device=x.device, dtype=torch.float32)
        wl = _conv_weight(w, 0)         # held until the launch is enqueued (an uncached temporary must not be handed out as the split-K workspace)
        gemm(_p(x), _p(wl), _p(y), B * Ho * Wo, Cout, 9 * Cin, lda=9 * Cin, ldb=9 * Cin, ldc=Cout, bias=_p(bias), allow_split=True,
             conv=(Cin, H, W_, Ho, Wo, stride, dilation))
        del wl
        ctx.save_for_backward(x, w)
        ctx.meta = (stride, dilation, bias is not None)
        ctx.bias_ref = bias if (bias is not None and bias.grad_fn is None) else None
        return y

    @staticmethod
    def backward(ctx, g):
        if g is None:
            return (None,) * 5
        x, w = ctx.saved_tensors
        stride, dilation, has_b = ctx.meta
        g = _c(g)
        B, H, W_, Cin = x.shape
        Cout = w.shape[0]
        Ho, Wo = g.shape[1], g.shape[2]
        M = B * Ho * Wo
        dx = dW = db = None
        if ctx.needs_input_grad[0]:
            dx = _empty_like(x)
            if Cout % 32 == 0 and (stride == 1 or dilation == 1):
                # the same implicit GEMM on dy against the flipped / transposed weight; stride 2 = a transposed convolution: dy read as
                # if zero-upsampled (conv_up), rows = the INPUT pixels
                wl = _conv_weight(w, 1)
                gemm(_p(g), _p(wl), _p(dx), B * H * W_, Cin, 9 * Cout, lda=9 * Cout, ldb=9 * Cout, ldc=Cin, allow_split=True,
                     conv=(Cout, Ho, Wo, H, W_, 1, dilation, stride))
                del wl
            else:                                   # strided: dcol = g W, folded back by col2im
                dcol = _empty((M, 9 * Cin), device=g.device, dtype=torch.float32)
                _dgrad(g.view(M, Cout), w, dcol, M, 9 * Cin, Cout, 9 * Cin, allow_split=True)
                call("mdvit_col2im3x3", _p(dcol), _p(dx), B, H, W_, Cin, stride, dilation, _stream())
        want_w = not _dgrad_only and (ctx.needs_input_grad[1] or (has_b and ctx.needs_input_grad[2]))
        if want_w:
            # dW'[co][tap][ci] = sum_m dy[m][co] x[pixel(m) + tap][ci]: the TN kernel gathers the image rows as its B operand (no im2col);
            # a small pass folds the tap-major result into the PyTorch [Cout, Cin, 3, 3] gradient (or its bucket view)
            sW = _sink_of(w)
            sb = _sink_of(ctx.bias_ref) if has_b else None
            sunk = sW is not None and (not has_b or sb is not None)
            g2 = g.view(M, Cout)
            cv = (Cin, H, W_, Ho, Wo, stride, dilation)
            # (round 4: the product's / the slab reduction's stores write the PyTorch layout themselves -- conv_wgrad_nchw -- so the tap-major temporary and
            #  its relayout launch per convolution are gone)
            if sunk:
                with _on_side(g, x):
                    gemm(_p(g2), _p(x), _p(sW), Cout, 9 * Cin, M, lda=Cout, ldb=9 * Cin, ldc=9 * Cin, trans_a=True, trans_b=False, allow_split=True, accumulate=True,
                         colsum_a=_p(sb) if has_b else None, conv=cv, conv_wgrad_nchw=True)
            else:
                if has_b and ctx.needs_input_grad[2]:
                    db = torch.zeros((Cout,), device=g.device, dtype=torch.float32)
                dW = _empty_like(w)
                gemm(_p(g2), _p(x), _p(dW), Cout, 9 * Cin, M, lda=Cout, ldb=9 * Cin, ldc=9 * Cin, trans_a=True, trans_b=False, allow_split=True,
                     colsum_a=_p(db) if db is not None else None, conv=cv, conv_wgrad_nchw=True)
        return dx, dW, db, None, None


def conv3x3_dense(x, w, bias=None, stride=1, dilation=1):
    """x NHWC [B,H,W,Cin], w [Cout,Cin,3,3] -> [B,Ho,Wo,Cout]; padding = dilation (dilation > 1 at stride 1 only)."""
    B, H, W_, Cn = x.shape
    Ho, Wo = (H - 1) // stride + 1, (W_ - 1) // stride + 1
    if _use_implicit_conv and _tn_kernel and _gemm_precision >= 1 and Cn % 32 == 0 and w.dim() == 4 and w.is_contiguous() and w.shape[0] % 4 == 0:
        return _Conv3x3.apply(_c(x), w, bias, int(stride), int(dilation))
    col = _Im2col.apply(_c(x), int(stride), int(dilation))
    y = _Linear.apply(col, w if w.is_contiguous() else w.reshape(w.shape[0], -1), bias, None, None, 0.0, 1)
    return y.view(B, Ho, Wo, w.shape[0])


class _Dropout(torch.autograd.Function):
    """element-wise nn.Dropout with the counter-hash mask of the GEMM epilogues (same call, same keys, on the gradient)"""

    @staticmethod
    def forward(ctx, x, p, key):
        ctx.set_materialize_grads(False)
        _chk(x)
        y = _empty_like(x)
        call("mdvit_dropout_f32", _p(x), _p(y), x.numel(), p, key[0], key[1], _seed_ptr(), _stream())
        ctx.meta = (p, key)
        return y

    @staticmethod
    def backward(ctx, g):
        if g is None:
            return None, None, None
        p, key = ctx.meta
        g = _c(g)
        dx = _empty_like(g)
        call("mdvit_dropout_f32", _p(g), _p(dx), g.numel(), p, key[0], key[1], _seed_ptr(), _stream())
        return dx, None, None


def droppath_scales(shape, keep: float, device):
    """Per-sample DropPath scales (Bernoulli(keep) / keep, timm's drop_path as used by mdvit.py:353-360) for a whole forward in ONE launch of the
    counter-hash dropout kernel on a cached tensor of ones -- no torch.rand / compare / divide on the path; under HIP-graph replay the device
    seed re-keys them like every other mask."""
    n = 1
    for v in shape:
        n *= int(v)
    n4 = (n + 3) // 4 * 4                              # the kernel works on quads
    ones = _ones_flat(n4, device)
    y = torch.empty((n4,), device=device, dtype=torch.float32)
    key = _next_key()
    call("mdvit_dropout_f32", _p(ones), _p(y), n4, 1.0 - float(keep), key[0], key[1], _seed_ptr(), _stream())
    return y[:n].view(*shape)


_ones_cache = {}


def _ones_flat(n, device):
    k = (str(device), n)
    t = _ones_cache.get(k)
    if t is None:
        if torch.cuda.is_current_stream_capturing():
            # a fill captured into a graph only runs at replay and lives in the graph's private pool: never cache it (an eager caller before the first
            # replay would read uninitialised "ones") -- a per-call tensor inside the capture instead
            return torch.ones((n,), device=device, dtype=torch.float32)
        t = _ones_cache[k] = torch.ones((n,), device=device, dtype=torch.float32)
    return t


def dropout(x, p: float, training: bool = True):
    if not training or p <= 0.0:
        return x
    return _Dropout.apply(_c(x), float(p), _next_key())


class _GlobalAvgPool(torch.autograd.Function):
    """NHWC [B,H,W,C] -> [B,C] token mean (nn.AdaptiveAvgPool2d(1)): column sums per sample; the backward broadcasts g / (H W)
    back over the pixels with the bilinear kernel (a 1x1 source is a constant)."""

    @staticmethod
    def forward(ctx, x):
        ctx.set_materialize_grads(False)
        _chk(x)
        B, H, W_, Cn = x.shape
        out = _empty((B, Cn), device=x.device, dtype=torch.float32)
        wsp, wsb, _keep = _partials_ws(Cn, x.device)
        for b in range(B):
            call("mdvit_colsum_f32", _p(x[b]), Cn, _p(out[b]), None, wsp, wsb, H * W_, Cn, 0.0, 0, 0, None, 1, 0, None, _stream())
        ctx.meta = (B, H, W_, Cn)
        return out.mul_(1.0 / (H * W_))

    @staticmethod
    def backward(ctx, g):
        if g is None:
            return None
        B, H, W_, Cn = ctx.meta
        gs = _c(g).mul(1.0 / (H * W_))
        dx = _empty((B, H, W_, Cn), device=g.device, dtype=torch.float32)
        call("mdvit_upsample_fwd", _p(gs), None, _p(dx), B, 1, 1, H, W_, Cn, _stream())
        return dx


def global_avg_pool(x):
    return _GlobalAvgPool.apply(_c(x))


class _StemConv(torch.autograd.Function):
    @staticmethod
    def forward(ctx, img, w):
        ctx.set_materialize_grads(False)
        _chk(img, w)
        B, Cin, H, W_ = img.shape
        Cout = w.shape[0]
        y = _empty((B, (H - 1) // 2 + 1, (W_ - 1) // 2 + 1, Cout), device=img.device, dtype=torch.float32)
        call("mdvit_stemconv_fwd", _p(img), _p(w), _p(y), B, H, W_, Cin, Cout, _stream())
        ctx.save_for_backward(img, w)
        return y

    @staticmethod
    def backward(ctx, g):
        if g is None:
            return (None,) * 2
        img, w = ctx.saved_tensors
        if ctx.needs_input_grad[0]:
            raise _lib.MdvitHipError("gradient w.r.t. the input image is not built (the train path never needs it)")
        if _dgrad_only:
            return None, None
        g = _c(g)
        B, Cin, H, W_ = img.shape
        sw = _sink_of(w) if _side_stream is not None else None
        if sw is not None:
            with _on_side(g, img):
                wsp, wsb, keep = _partials_ws(27 * w.shape[0], g.device)
                call("mdvit_stemconv_wgrad", _p(img), _p(g), _p(sw), wsp, wsb, B, H, W_, Cin, w.shape[0], 1, _stream())
            return None, None
        dw = _empty_like(w)
        wsp, wsb, _keep = _partials_ws(27 * w.shape[0], g.device)
        call("mdvit_stemconv_wgrad", _p(img), _p(g), _p(dw), wsp, wsb, B, H, W_, Cin, w.shape[0], 0, _stream())
        return None, dw


def stem_conv(img, w):
    return _StemConv.apply(_c(img), w)


# ------------------------------------------------------------------------------------------------
# BatchNorm2d (+activation, +Dropout2d) on NHWC
# ------------------------------------------------------------------------------------------------
class _BNAct(torch.autograd.Function):
    @staticmethod
    def forward(ctx, y, gamma, beta, running_mean, running_var, nbt, training, eps, momentum, act, drop2d_p, rows_per_sample, groups):
        ctx.set_materialize_grads(False)
        _chk(y, gamma, beta, running_mean, running_var)
        Cn = y.shape[-1]
        M = y.numel() // Cn
        dev = y.device
        pga = int(gamma.dim() == 2)    # [G, C] parameters / running statistics: row g belongs to group g (domain-specific norms)
        if pga:
            if gamma.shape[0] != groups:
                raise ValueError(f"bn_act: {gamma.shape[0]} parameter rows for {groups} groups")
        elif not training:
            groups = 1                 # eval normalises every sample with the shared running statistics
        mean = _empty((groups, Cn), device=dev, dtype=torch.float32)
        rstd = _empty_like(mean)
        if training:
            wsb = _lib.load().mdvit_bn_ws_bytes(M, Cn, groups)
            ws = _empty((wsb // 8 + 1,), device=dev, dtype=torch.float64)
            call("mdvit_bn_stats", _p(y), _p(ws), wsb, _p(mean), _p(rstd), _p(running_mean), _p(running_var),
                 C.c_void_p(nbt.data_ptr()) if nbt is not None else None, M, Cn, groups, pga, eps, momentum, _stream())
        else:
            call("mdvit_bn_eval_prep", _p(running_mean), _p(running_var), _p(mean), _p(rstd), Cn * groups, eps, _stream())
        key = _next_key() if drop2d_p > 0 else (0, 0)
        z = _empty_like(y)
        call("mdvit_bn_apply", _p(y), _p(mean), _p(rstd), _p(gamma), _p(beta), _p(z), M, Cn, groups, pga, act, drop2d_p, key[0], key[1],
             _seed_ptr() if drop2d_p > 0 else None, rows_per_sample, _stream())
        ctx.save_for_backward(y, gamma, beta, mean, rstd)
        ctx.meta = (training, act, drop2d_p, key, rows_per_sample, groups, pga)
        return z

    @staticmethod
    def backward(ctx, g):
        if g is None:
            return (None,) * 13
        y, gamma, beta, mean, rstd = ctx.saved_tensors
        training, act, drop2d_p, key, rps, groups, pga = ctx.meta
        g = _c(g)
        Cn = y.shape[-1]
        M = y.numel() // Cn
        dy = _empty_like(y)
        dg, db = _empty_like(gamma), _empty_like(gamma)
        wsb = _lib.load().mdvit_bn_ws_bytes(M, Cn, groups)
        ws = _empty((wsb // 8 + 1,), device=y.device, dtype=torch.float64)
        call("mdvit_bn_bwd", _p(g), _p(y), _p(mean), _p(rstd), _p(gamma), _p(beta), _p(dy), _p(dg), _p(db), _p(ws), wsb,
             M, Cn, groups, pga, act, int(training), drop2d_p, key[0], key[1], _seed_ptr() if drop2d_p > 0 else None, rps, _stream())
        if _dgrad_only:
            dg = db = None
        return dy, dg, db, None, None, None, None, None, None, None, None, None, None


class _BNActRowDot(torch.autograd.Function):
    """BatchNorm2d -> activation -> Dropout2d -> 1-output 1x1 conv as ONE op (csrc/norm.hip: bn_rowdot_*): the tail of the peer heads
    (Decoders.py:304-311,333-336).  The normalised tensor between the norm and the conv is never written; the backward needs y and the
    row gradient only."""

    @staticmethod
    def forward(ctx, y, gamma, beta, running_mean, running_var, nbt, w, b, training, eps, momentum, act, drop2d_p, rows_per_sample):
        ctx.set_materialize_grads(False)
        _chk(y, gamma, beta, running_mean, running_var, w, b)
        Cn = y.shape[-1]
        M = y.numel() // Cn
        dev = y.device
        mean = _empty((Cn,), device=dev, dtype=torch.float32)
        rstd = _empty_like(mean)
        if training:
            wsb = _lib.load().mdvit_bn_ws_bytes(M, Cn, 1)
            ws = _empty((wsb // 8 + 1,), device=dev, dtype=torch.float64)
            call("mdvit_bn_stats", _p(y), _p(ws), wsb, _p(mean), _p(rstd), _p(running_mean), _p(running_var),
                 C.c_void_p(nbt.data_ptr()) if nbt is not None else None, M, Cn, 1, 0, eps, momentum, _stream())
        else:
            call("mdvit_bn_eval_prep", _p(running_mean), _p(running_var), _p(mean), _p(rstd), Cn, eps, _stream())
        key = _next_key() if drop2d_p > 0 else (0, 0)
        low = _empty((M,), device=dev, dtype=torch.float32)
        call("mdvit_bn_rowdot_fwd", _p(y), _p(mean), _p(rstd), _p(gamma), _p(beta), _p(w), _p(b), _p(low), M, Cn, act, drop2d_p, key[0], key[1],
             _seed_ptr() if drop2d_p > 0 else None, rows_per_sample, _stream())
        ctx.save_for_backward(y, gamma, beta, mean, rstd, w)
        ctx.meta = (training, act, drop2d_p, key, rows_per_sample, b is not None)
        return low

    @staticmethod
    def backward(ctx, g):
        if g is None:
            return (None,) * 14
        y, gamma, beta, mean, rstd, w = ctx.saved_tensors
        training, act, drop2d_p, key, rps, has_b = ctx.meta
        g = _c(g)
        Cn = y.shape[-1]
        M = y.numel() // Cn
        dy = _empty_like(y)
        dg = db_ = dw = dbias = None
        if not _dgrad_only:
            dg, db_, dw, dbias = _flat_like(gamma, gamma, w.reshape(-1), (1,) if has_b else None)
        wsb = _lib.load().mdvit_bn_rowdot_ws_bytes(M, Cn)
        ws = _empty((wsb // 8 + 1,), device=y.device, dtype=torch.float64)
        call("mdvit_bn_rowdot_bwd", _p(g), _p(y), _p(mean), _p(rstd), _p(gamma), _p(beta), _p(w), _p(dy), _p(dg), _p(db_), _p(dw), _p(dbias), _p(ws), wsb,
             M, Cn, act, int(training), drop2d_p, key[0], key[1], _seed_ptr() if drop2d_p > 0 else None, rps, _stream())
        return (dy, dg, db_, None, None, None, None if dw is None else dw.view_as(w), dbias, None, None, None, None, None, None)


def bn_act_rowdot(y, gamma, beta, running_mean, running_var, nbt, training, act, w, b, eps=1e-5, momentum=0.1, drop2d_p=0.0):
    """y NHWC [B,h,w,C] -> [B,h,w]: the 1-channel 1x1 conv (w: C elements, b: [1] or None) of Dropout2d(act(BatchNorm(y))).  One statistics
    group with shared parameters and C in {256, 512, 1024}: one fused op; anything else: bn_act + rowdot."""
    Cn = y.shape[-1]
    groups = _bn_groups if (training or gamma.dim() == 2) else 1
    if groups == 1 and gamma.dim() == 1 and Cn in (256, 512, 1024) and y.is_cuda:
        rows_per_sample = y.numel() // (y.shape[0] * Cn)
        low = _BNActRowDot.apply(_c(y), _c(gamma), _c(beta), running_mean, running_var, nbt, w, b, bool(training), float(eps), float(momentum), int(act),
                                 float(drop2d_p), int(rows_per_sample))
        return low.view(y.shape[:-1])
    return rowdot(bn_act(y, gamma, beta, running_mean, running_var, nbt, training, act, eps, momentum, drop2d_p), w, b)


# ---- domain-batched forward: the batch is `groups` equal consecutive domain batches -----------------------------
# The reference runs one forward per domain (multi_train_MDViT.py:137-153); the only op on the trunk that couples the
# samples of a forward is BatchNorm (batch statistics).  With bn_groups(G) active every train-mode BatchNorm keeps
# per-group statistics, so ONE forward over the concatenated G domain batches is the same function as G forwards --
# with 1/G of the kernel launches and G-times larger (better filled) kernels.
_bn_groups = 1


class bn_groups:
    def __init__(self, groups: int):
        self.groups = int(groups)

    def __enter__(self):
        global _bn_groups
        self.prev, _bn_groups = _bn_groups, self.groups

    def __exit__(self, *exc):
        global _bn_groups
        _bn_groups = self.prev


class _Fork(torch.autograd.Function):
    """x -> n aliases of x for n consumers; the backward adds the n gradients HERE (one launch per extra gradient, on the stream
    this backward runs on).  Without it autograd sums the gradients of a tensor with several consumers itself -- at::add launches
    on the stream it attributes to the node, which is not where a sweep that was moved to another stream (set_sweep_stream)
    produced them."""

    @staticmethod
    def forward(ctx, x, n):
        ctx.set_materialize_grads(False)
        return tuple(x.view_as(x) for _ in range(n))

    @staticmethod
    def backward(ctx, *gs):
        gs = [g for g in gs if g is not None]
        if not gs:
            return None, None
        acc = _c(gs[0])
        rest = [_c(g) for g in gs[1:]]
        while rest:
            out = _empty_like(acc)
            if len(rest) >= 2:          # three at a time: (a + b) + c in one pass (the same additions in the same order as two passes)
                call("mdvit_add3", _p(acc), _p(rest[0]), _p(rest[1]), _p(out), acc.numel(), _stream())
                rest = rest[2:]
            else:
                call("mdvit_ew", _p(acc), _p(rest[0]), _p(out), acc.numel(), 3, _stream())          # mode 3: y = a + b
                rest = rest[1:]
            acc = out
        return acc, None


class _ForkGroups(torch.autograd.Function):
    """x -> n full aliases + G equal batch-group views (the peer heads' inputs); the backward adds the n full gradients and the G part gradients in ONE pass
    (_Fork + _SplitGroups paid a concatenation of the parts and one or two additions)."""

    @staticmethod
    def forward(ctx, x, n, G):
        ctx.set_materialize_grads(False)
        ctx.meta = (n, G, tuple(x.shape))
        return tuple(x.view_as(x) for _ in range(n)) + tuple(x.view((G, x.shape[0] // G) + tuple(x.shape[1:])).unbind(0))

    @staticmethod
    def backward(ctx, *gs):
        n, G, shape = ctx.meta
        full = [_c(g) for g in gs[:n] if g is not None]
        parts = gs[n:]
        if all(p is None for p in parts):
            parts = None
        elif any(p is None for p in parts) or len(full) == 0 or len(full) > 2 or parts[0].numel() % 4:
            ref = next(p for p in parts if p is not None)          # unusual: fall back to a materialised concatenation
            full.append(torch.cat([p if p is not None else torch.zeros_like(ref) for p in parts], 0).view(shape))
            parts = None
        if parts is None:
            if not full:
                return None, None, None
            acc = full[0]
            for g in full[1:]:
                out = _empty_like(acc)
                call("mdvit_ew", _p(acc), _p(g), _p(out), acc.numel(), 3, _stream())
                acc = out
            return acc, None, None
        parts = [_c(p) for p in parts]
        out = _empty_like(full[0])
        call("mdvit_add_parts", _p(full[0]), _p(full[1]) if len(full) > 1 else None, _vp_array([_p(p) for p in parts]), G, parts[0].numel(), _p(out), _stream())
        return out, None, None


def fork_groups(x, n: int, G: int):
    """n aliases of x + its G batch-group views (see _ForkGroups); -> (aliases..., parts tuple)"""
    if not x.requires_grad or G <= 1:
        al = fork(x, n)
        return tuple(al) + (split_groups(x, G),)
    out = _ForkGroups.apply(_c(x), int(n), int(G))
    return tuple(out[:n]) + (tuple(out[n:]),)


# autograd nodes of torch's own whose backward launches nothing (they hand views / the same tensor on): harmless inside a sweep that runs on another stream
_KERNEL_FREE_NATIVE = ("AddBackward0", "CatBackward0", "ViewBackward0", "UnsafeViewBackward0", "ReshapeAliasBackward0", "PermuteBackward0", "TBackward0", "TransposeBackward0",
                       "AliasBackward0", "SqueezeBackward0", "SqueezeBackward1", "UnsqueezeBackward0", "DetachBackward0", "CloneBackward0", "AccumulateGrad")


def audit_sweep_graph(root):
    """Walk the autograd graph below `root` and report what would be launched by the ENGINE rather than by this package's Functions in a backward from it:
    -> (native, fanin) -- `native`: names of torch-native nodes whose backward runs a kernel; `fanin`: names of nodes that receive more than one gradient
    (autograd adds those itself).  Both run on the stream the FORWARD ran on: a sweep moved to another stream (set_sweep_stream) is unordered with them, so a
    model whose aux graph reports any may not list itself in model.AUX_SWEEP_STREAM_SAFE.  (AccumulateGrad fan-in is reported by parameter name only when the
    node can receive a gradient in the data-gradient-only sweep, i.e. never for sunk weights -- the caller filters.)"""
    import collections
    seen, stack, consumers = set(), [root.grad_fn], collections.Counter()
    native = []
    while stack:
        fn = stack.pop()
        if fn is None or fn in seen:
            continue
        seen.add(fn)
        name = type(fn).__name__
        ours = name.endswith("Backward") and name[:-len("Backward")] in globals()
        if not ours and name not in _KERNEL_FREE_NATIVE:
            native.append(name)
        for nf, idx in fn.next_functions:
            if nf is not None:
                consumers[(nf, idx)] += 1
                stack.append(nf)
    fanin = {type(nf).__name__ + (":" + str(idx) if idx else "") for (nf, idx), c in consumers.items() if c > 1 and type(nf).__name__ != "AccumulateGrad"}
    if _sinks:
        # with gradient sinks active every parameter the sweep reaches is expected to have one (this package's Functions then write the gradient themselves and hand
        # autograd None); a parameter WITHOUT a sink gets its gradient from the engine's AccumulateGrad -- an add on the stream of the forward
        for (nf, idx), c in consumers.items():
            if type(nf).__name__ == "AccumulateGrad":
                var = getattr(nf, "variable", None)
                if var is not None and var.requires_grad and _sink_of(var) is None:
                    fanin.add("AccumulateGrad without a gradient sink: parameter of shape " + "x".join(str(d) for d in var.shape))
    return sorted(set(native)), sorted(fanin)


def fork(x, n: int):
    """n aliases of x, one per consumer (see _Fork)"""
    return _Fork.apply(x, int(n)) if (n > 1 and x.requires_grad) else (x,) * n


class _SplitGroups(torch.autograd.Function):
    """x [G*B, ...] -> G views [B, ...] (no copy); backward concatenates the G gradients once."""

    @staticmethod
    def forward(ctx, x, groups):
        ctx.set_materialize_grads(False)
        ctx.shape = (groups, x.shape[0] // groups) + tuple(x.shape[1:])
        return tuple(x.view(ctx.shape).unbind(0))

    @staticmethod
    def backward(ctx, *gs):
        if all(g is None for g in gs):
            return None, None
        if all(g is not None for g in gs):          # the usual case: ONE concatenation launch instead of a copy per group
            return torch.cat(gs, 0), None
        ref = next(g for g in gs if g is not None)
        out = torch.empty(ctx.shape, device=ref.device, dtype=ref.dtype)
        for i, g in enumerate(gs):
            if g is None:
                out[i].zero_()
            else:
                out[i].copy_(g)
        return out.view((ctx.shape[0] * ctx.shape[1],) + ctx.shape[2:]), None


class _SplitCols(torch.autograd.Function):
    """W [R, sum(sizes)] -> column-block views (no copy); backward concatenates the block gradients once.  (Plain slicing makes
    autograd allocate a zero tensor of W's size, copy the block gradient in and add it, per block.)"""

    @staticmethod
    def forward(ctx, W, sizes):
        ctx.set_materialize_grads(False)
        ctx.sizes, ctx.rows = tuple(sizes), W.shape[0]
        out, off = [], 0
        for n in sizes:
            out.append(W[:, off:off + n])
            off += n
        return tuple(out)

    @staticmethod
    def backward(ctx, *gs):
        if all(g is None for g in gs):
            return None, None
        ref = next(g for g in gs if g is not None)
        parts = [g if g is not None else torch.zeros((ctx.rows, n), device=ref.device, dtype=ref.dtype) for g, n in zip(gs, ctx.sizes)]
        return torch.cat(parts, 1), None


class _CatChannels(torch.autograd.Function):
    """[.., C0] | [.., C1] -> [.., C0 + C1] (one copy launch); the backward hands each source its own CONTIGUOUS gradient block (two strided copies on the stream the
    sweep runs on -- a native CatBackward0 would hand column-slice views that every consumer then copies for itself).  Used where two features of one resolution feed
    the same 1x1 convolution sum: ONE product over the concatenated K axis writes the [tokens, hidden] tensor once (decode.MLPDecoderFM)."""

    @staticmethod
    def forward(ctx, a, b):
        ctx.set_materialize_grads(False)
        ctx.sizes = (a.shape[-1], b.shape[-1])
        return torch.cat([a, b], dim=-1)

    @staticmethod
    def backward(ctx, g):
        if g is None:
            return None, None
        c0, _c1 = ctx.sizes
        return (g[..., :c0].contiguous() if ctx.needs_input_grad[0] else None), (g[..., c0:].contiguous() if ctx.needs_input_grad[1] else None)


def cat_channels(a, b):
    return _CatChannels.apply(_c(a), _c(b))


def split_cols(W, sizes):
    assert W.dim() == 2 and sum(sizes) == W.shape[1]
    return _SplitCols.apply(W, tuple(int(n) for n in sizes))


def split_groups(x, groups: int):
    if groups == 1:
        return (x,)
    assert x.shape[0] % groups == 0
    return _SplitGroups.apply(_c(x), int(groups))


def bn_act(y, gamma, beta, running_mean, running_var, nbt, training, act, eps=1e-5, momentum=0.1, drop2d_p=0.0):
    """gamma/beta/running_* [C]; or [G, C] (+ nbt [G] or None) with bn_groups(G) active: group g is normalised with -- and
    updates -- parameter/statistics row g (the per-domain norm banks of MDViT_DSN on a domain-batched tensor)."""
    rows_per_sample = y.numel() // (y.shape[0] * y.shape[-1])
    groups = _bn_groups if (training or gamma.dim() == 2) else 1
    if groups > 1 and y.shape[0] % groups:
        raise ValueError(f"bn_groups({groups}) needs a batch that is a multiple of it, got {y.shape[0]}")
    return _BNAct.apply(_c(y), _c(gamma), _c(beta), running_mean, running_var, nbt, bool(training), float(eps), float(momentum),
                        int(act), float(drop2d_p), int(rows_per_sample), int(groups))


# ------------------------------------------------------------------------------------------------
# bilinear resize (align_corners=False), optionally accumulated onto a base tensor
# ------------------------------------------------------------------------------------------------
class _Upsample(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, Ho, Wo, base):
        ctx.set_materialize_grads(False)
        _chk(x, base)
        B, H, W_, Cn = x.shape
        y = _empty((B, Ho, Wo, Cn), device=x.device, dtype=torch.float32)
        call("mdvit_upsample_fwd", _p(x), _p(base), _p(y), B, H, W_, Ho, Wo, Cn, _stream())          # y = (base +) resize(x): one pass, no clone
        ctx.meta = (B, H, W_, Ho, Wo, Cn, base is not None)
        return y

    @staticmethod
    def backward(ctx, g):
        if g is None:
            return (None,) * 4
        B, H, W_, Ho, Wo, Cn, has_base = ctx.meta
        g = _c(g)
        dx = _empty((B, H, W_, Cn), device=g.device, dtype=torch.float32)
        wsb = _lib.load().mdvit_upsample_bwd_ws_bytes(B, H, W_, Ho, Wo, Cn)
        ws = _empty((wsb // 4,), device=g.device, dtype=torch.float32)
        call("mdvit_upsample_bwd", _p(g), _p(dx), _p(ws), wsb, B, H, W_, Ho, Wo, Cn, _stream())
        return dx, None, None, (g if has_base else None)


def upsample_bilinear(x, Ho, Wo, base=None):
    """NHWC x -> [B,Ho,Wo,C] (+ base)."""
    if base is None and x.shape[1] == Ho and x.shape[2] == Wo:
        return x
    return _Upsample.apply(_c(x), int(Ho), int(Wo), None if base is None else _c(base))


class _UpsampleSum(torch.autograd.Function):
    """y = base + sum_i resize(x_i) for up to three NHWC sources of different sizes in ONE pass; the backward folds dy along W for all sources
    in one launch (the dy row is read from HBM once), along H per source; base's gradient is dy itself."""

    @staticmethod
    def forward(ctx, base, Ho, Wo, *xs):
        ctx.set_materialize_grads(False)
        _chk(base, *xs)
        B, _, _, Cn = xs[0].shape
        n = len(xs)
        y = _empty((B, Ho, Wo, Cn), device=xs[0].device, dtype=torch.float32)
        ptrs = (C.c_void_p * n)(*[x.data_ptr() for x in xs])
        Hi = (C.c_int32 * n)(*[x.shape[1] for x in xs])
        Wi = (C.c_int32 * n)(*[x.shape[2] for x in xs])
        call("mdvit_upsample_multi_fwd", ptrs, Hi, Wi, n, _p(base), _p(y), B, Ho, Wo, Cn, _stream())
        ctx.meta = (B, Ho, Wo, Cn, [(x.shape[1], x.shape[2]) for x in xs], base is not None)
        return y

    @staticmethod
    def backward(ctx, g):
        B, Ho, Wo, Cn, dims, has_base = ctx.meta
        if g is None:
            return (None,) * (3 + len(dims))
        g = _c(g)
        n = len(dims)
        dxs = [_empty((B, h, w, Cn), device=g.device, dtype=torch.float32) for h, w in dims]
        ptrs = (C.c_void_p * n)(*[t.data_ptr() for t in dxs])
        Hi = (C.c_int32 * n)(*[h for h, _ in dims])
        Wi = (C.c_int32 * n)(*[w for _, w in dims])
        wsb = _lib.load().mdvit_upsample_multi_bwd_ws_bytes(Wi, n, B, Ho, Cn)
        ws = _empty((wsb // 4,), device=g.device, dtype=torch.float32)
        call("mdvit_upsample_multi_bwd", _p(g), ptrs, Hi, Wi, n, _p(ws), wsb, B, Ho, Wo, Cn, _stream())
        return (g if has_base else None, None, None, *dxs)


def upsample_sum(base, xs, Ho, Wo):
    """base [B,Ho,Wo,C] (or None) + the bilinear resizes (align_corners=False) of the NHWC tensors xs to Ho x Wo, in one pass per three sources."""
    xs = [_c(x) for x in xs]
    out = base
    while xs:
        chunk, xs = xs[:3], xs[3:]
        if chunk[0].shape[-1] % 4 or not chunk[0].is_cuda:
            for x in chunk:
                out = upsample_bilinear(x, Ho, Wo, base=out)
            continue
        out = _UpsampleSum.apply(None if out is None else _c(out), int(Ho), int(Wo), *chunk)
    return out


# ------------------------------------------------------------------------------------------------
# 1-output linear (finalconv / linear_out)
# ------------------------------------------------------------------------------------------------
class _RowDot(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, w, b):
        ctx.set_materialize_grads(False)
        _chk(w, b)
        M, K, ldx = _ld_view(x)
        y = _empty((M,), device=x.device, dtype=torch.float32)
        call("mdvit_rowdot_fwd", _p(x), ldx, _p(w), _p(b), _p(y), M, K, 0, _stream())
        ctx.save_for_backward(x, w)
        ctx.has_b = b is not None
        return y

    @staticmethod
    def backward(ctx, g):
        if g is None:
            return (None,) * 3
        x, w = ctx.saved_tensors
        g = _c(g)
        M, K, ldx = _ld_view(x)
        dx = _empty((M, K), device=x.device, dtype=torch.float32) if ctx.needs_input_grad[0] else None
        if _dgrad_only:                  # the aux sweep: dx = g w only
            if dx is not None:
                call("mdvit_rowdot_bwd", _p(x), ldx, _p(w), _p(g), _p(dx), K, None, None, None, 0, M, K, _stream())
            return dx, None, None
        dw, db = _flat_like(w.reshape(-1), (1,) if ctx.has_b else None)
        wsp, wsb, _keep = _partials_ws(K + 1, g.device)
        call("mdvit_rowdot_bwd", _p(x), ldx, _p(w), _p(g), _p(dx), K, _p(dw), _p(db), wsp, wsb, M, K, _stream())
        return dx, dw.view_as(w), db


def rowdot(x, w, b=None):
    """x [..., K] . w (any shape with K elements) + b[1] -> [...].  A 2-D x may be a column-slice view."""
    if x.dim() == 2:
        return _RowDot.apply(x, w, b)
    shp = x.shape
    return _RowDot.apply(_c(x).view(-1, shp[-1]), w, b).view(shp[:-1])


# ------------------------------------------------------------------------------------------------
# Domain adapter + factorized attention core
# ------------------------------------------------------------------------------------------------
_da_pre = None          # inside da_precomputed(): {W2.data_ptr(): a [B, C]} for the label batch of the forward in progress
_da_many = os.environ.get("MDVIT_DA_MANY", "1") != "0"      # 0: every block launches its own adapter kernels, forward and backward (A/B)
_da_many_bwd = os.environ.get("MDVIT_DA_MANY_BWD", "1") != "0"      # 0: the forward in one launch, the backward inside each block (A/B)


def _da_many_desc(label, params, heads, outs):
    m = _lib.DaMany()
    m.n = len(heads)
    for i in range(m.n):
        W1, b1, W2, b2 = params[4 * i:4 * i + 4]
        m.hid[i], m.C[i], m.heads[i] = W1.shape[0], W2.shape[0], int(heads[i])
        m.W1[i], m.b1[i], m.W2[i], m.b2[i], m.a[i] = _p(W1), _p(b1), _p(W2), _p(b2), _p(outs[i])
    return m


class _DaMany(torch.autograd.Function):
    """Every domain adapter of a network as ONE node: forward = mdvit_da_fwd_many (one launch), backward = mdvit_da_bwd_many (two launches) once the attention
    backward of every block has handed its e = a * dL/da in as the 'gradient' of that block's a (the nodes below return e, not dL/da: no division by a)."""

    @staticmethod
    def forward(ctx, label, heads, *params):
        ctx.set_materialize_grads(False)
        B, D = label.shape
        outs = tuple(_empty((B, params[4 * i + 2].shape[0]), device=label.device, dtype=torch.float32) for i in range(len(heads)))
        m = _da_many_desc(label, params, heads, outs)
        call("mdvit_da_fwd_many", C.byref(m), _p(label), B, D, _stream())
        ctx.save_for_backward(label, *params, *outs)
        ctx.heads = heads
        return outs

    @staticmethod
    def backward(ctx, *es):
        heads = ctx.heads
        n = len(heads)
        label, *rest = ctx.saved_tensors
        params, outs = rest[:4 * n], rest[4 * n:]
        if all(e is None for e in es):
            return (None,) * (2 + 4 * n)
        B, D = label.shape
        m = _da_many_desc(label, params, heads, outs)
        g = _lib.DaManyGrads()
        grads = [None] * (4 * n)
        live = [i for i in range(n) if es[i] is not None]
        bufs = _flat_like(*[params[4 * i + j] for i in live for j in range(4)])
        keep = []
        for k, i in enumerate(live):
            e = _c(es[i])
            keep.append(e)
            g.e[i] = _p(e)
            grads[4 * i:4 * i + 4] = bufs[4 * k:4 * k + 4]
            g.dW1[i], g.db1[i], g.dW2[i], g.db2[i] = (_p(t) for t in bufs[4 * k:4 * k + 4])
        wsb = _lib.load().mdvit_da_many_ws_bytes(C.byref(m), B)
        ws = _empty((wsb // 4,), device=label.device, dtype=torch.float32)
        # dgrad-only (aux) sweep: MINUS the adapters' gradients, cancelled against the merged sweep's (see set_dgrad_only)
        call("mdvit_da_bwd_many", C.byref(m), C.byref(g), _p(label), -1.0 if _dgrad_only else 1.0, _p(ws), wsb, B, D, _stream())
        return (None, None, *grads)


class da_precomputed:
    """Every domain adapter of a network for ONE label batch in one launch (mdvit_da_fwd_many) at the top of the forward: an adapter's output depends on the labels
    and its own four tensors only.  Inside the context the attention nodes (_FactorAtt, the C-level block through MdvitBlockDesc.a_pre) pick their `a` up by the
    adapter's second weight instead of launching mdvit_da_fwd (16 launches of ~13 us on the single-stream forward of an MDViT), and hand their e = a * dL/da back
    to the one node that owns all adapters (_DaMany) instead of launching mdvit_da_bwd (two launches of 8-12 us per block and sweep on the data-gradient chains).
    adapters: [(W1, b1, W2, b2, heads)].  Same arithmetic as the per-block launches, bit for bit."""

    def __init__(self, label, adapters):
        self.label, self.adapters = label, adapters

    def __enter__(self):
        global _da_pre
        self.prev = _da_pre
        ad = self.adapters
        if not (_da_many and self.label is not None and ad and len(ad) <= _lib.DA_MANY_MAX and self.label.is_cuda):
            return self
        label = _c(self.label.float())
        params = [t for (W1, b1, W2, b2, _) in ad for t in (W1, b1, W2, b2)]
        if not all(t.is_contiguous() and t.dtype == torch.float32 for t in params):
            return self
        heads = tuple(int(h) for (_, _, _, _, h) in ad)
        if _da_many_bwd:
            outs = _DaMany.apply(label, heads, *params)
        else:
            with torch.no_grad():
                outs = _DaMany.apply(label, heads, *params)
        _da_pre = {ad[i][2].data_ptr(): outs[i] for i in range(len(ad))}
        return self

    def __exit__(self, *exc):
        global _da_pre
        _da_pre = self.prev
        return False


def _da_lookup(W2, B):
    if _da_pre is None or W2 is None:
        return None
    a = _da_pre.get(W2.data_ptr())
    return a if (a is not None and a.shape[0] == B) else None


class _FactorAtt(torch.autograd.Function):
    """Attention core + Domain Adapter as ONE node: the adapter's backward consumes e = a * dL/da straight
    from the attention backward (no division by a).  label is None -> no adapter (BASE / mpvit flavour)."""

    @staticmethod
    def forward(ctx, qkv, w3, b3, w5, b5, w7, b7, label, W1, b1, W2, b2, a_pre, H, W_, heads, splits, aux_first=False):
        _chk(qkv, w3, b3, w5, b5, w7, b7, label, W1, b1, W2, b2, a_pre)
        ctx.set_materialize_grads(False)
        ctx.aux_first = bool(aux_first)
        B, N, C3 = qkv.shape
        Cn = C3 // 3
        Ch = Cn // heads
        dev = qkv.device
        a = None
        ctx.e_out = False          # True: a is an output of the all-adapters node (_DaMany) -- e goes back to it as a's gradient, no adapter backward here
        if label is not None:
            a = a_pre
            ctx.e_out = a_pre is not None and ctx.needs_input_grad[12]
            if a is None:
                a = _empty((B, Cn), device=dev, dtype=torch.float32)
                call("mdvit_da_fwd", _p(label), _p(W1), _p(b1), _p(W2), _p(b2), _p(a), B, label.shape[1], W1.shape[0], Cn, heads, _stream())
        out = _empty((B, N, Cn), device=dev, dtype=torch.float32)
        U = _empty_like(out)
        kmax = _empty((B, Cn), device=dev, dtype=torch.float32)
        ksum = _empty_like(kmax)
        Mmat = _empty((B, Cn, Ch), device=dev, dtype=torch.float32)
        wsb = _lib.load().mdvit_factoratt_ws_bytes(B, N, Cn, heads)
        ws = _empty((wsb // 4,), device=dev, dtype=torch.float32)
        call("mdvit_factoratt_fwd", _p(qkv), _p(w3), _p(b3), _p(w5), _p(b5), _p(w7), _p(b7), _p(a), _p(out), _p(U), _p(kmax), _p(ksum),
             _p(Mmat), _p(ws), wsb, B, H, W_, Cn, heads, splits[0], splits[1], splits[2], _stream())
        ctx.save_for_backward(qkv, w3, b3, w5, b5, w7, b7, label, W1, b1, W2, b2, a, out, U, kmax, ksum, Mmat)
        ctx.meta = (H, W_, heads, splits)
        return out

    @staticmethod
    def backward(ctx, g):
        if g is None:
            return (None,) * 18
        qkv, w3, b3, w5, b5, w7, b7, label, W1, b1, W2, b2, a, out, U, kmax, ksum, Mmat = ctx.saved_tensors
        H, W_, heads, splits = ctx.meta
        g = _c(g)
        B, N, C3 = qkv.shape
        Cn = C3 // 3
        dev = qkv.device
        if _dgrad_only and ctx.aux_first and a is not None:
            # the first adapter of the network in the aux sweep: its (negated) gradient from e = sum_n g out, and nothing is handed on
            e = _empty((B, Cn), device=dev, dtype=torch.float32)
            wsb = _lib.load().mdvit_factoratt_ws_bytes(B, N, Cn, heads)
            ws = _empty((wsb // 4,), device=dev, dtype=torch.float32)
            call("mdvit_factoratt_bwd", _p(g), _p(qkv), _p(out), _p(U), _p(w3), _p(b3), _p(w5), _p(b5), _p(w7), _p(b7), _p(a), _p(kmax), _p(ksum),
                 _p(Mmat), None, _p(e), None, None, None, None, None, None, _p(ws), wsb, B, H, W_, Cn, heads, splits[0], splits[1], splits[2], _stream())
            if ctx.e_out:
                return (None,) * 12 + (e,) + (None,) * 5
            hid = W1.shape[0]
            dW1, db1, dW2, db2 = _empty_like(W1), _empty_like(b1), _empty_like(W2), _empty_like(b2)
            dab = _lib.load().mdvit_da_ws_bytes(B, hid, Cn)
            daws = _empty((dab // 4,), device=dev, dtype=torch.float32)
            call("mdvit_da_bwd", _p(label), _p(W1), _p(b1), _p(W2), _p(b2), _p(a), _p(e), -1.0, _p(dW1), _p(db1), _p(dW2), _p(db2), _p(daws), dab,
                 B, label.shape[1], hid, Cn, heads, _stream())
            return (None,) * 8 + (dW1, db1, dW2, db2) + (None,) * 6
        dqkv = _empty_like(qkv)
        crpe = (w3, b3, w5, b5, w7, b7)
        sinks = [_sink_of(t) for t in crpe] if (not _dgrad_only and _side_stream is not None) else [None] * 6
        deferred = all(t is not None for t in sinks)        # window-weight gradients on the side stream, straight into the buckets
        if _dgrad_only or deferred:
            e = _empty((B, Cn), device=dev, dtype=torch.float32) if a is not None else None
            dws = [None] * 6
        else:
            e, *dws = _flat_like((B, Cn) if a is not None else None, w3, b3, w5, b5, w7, b7)
        wsb = _lib.load().mdvit_factoratt_ws_bytes(B, N, Cn, heads)
        ws = _empty((wsb // 4,), device=dev, dtype=torch.float32)
        call("mdvit_factoratt_bwd", _p(g), _p(qkv), _p(out), _p(U), _p(w3), _p(b3), _p(w5), _p(b5), _p(w7), _p(b7), _p(a), _p(kmax), _p(ksum),
             _p(Mmat), _p(dqkv), _p(e), *[_p(t) for t in dws], _p(ws), wsb, B, H, W_, Cn, heads, splits[0], splits[1], splits[2],
             _stream())
        if deferred:
            with _on_side(ws, qkv, foreign=False):
                call("mdvit_factoratt_wgrad", _p(qkv), _p(ws), wsb, *[_p(t) for t in sinks], B, H, W_, Cn, heads,
                     splits[0], splits[1], splits[2], 1, _stream())
        dW1 = db1 = dW2 = db2 = None
        if a is not None and not ctx.e_out:
            hid = W1.shape[0]
            dW1, db1, dW2, db2 = _empty_like(W1), _empty_like(b1), _empty_like(W2), _empty_like(b2)
            dab = _lib.load().mdvit_da_ws_bytes(B, hid, Cn)
            daws = _empty((dab // 4,), device=dev, dtype=torch.float32)
            # dgrad-only (aux) sweep: MINUS the adapter gradient, cancelled against the merged sweep's (see set_dgrad_only)
            call("mdvit_da_bwd", _p(label), _p(W1), _p(b1), _p(W2), _p(b2), _p(a), _p(e), -1.0 if _dgrad_only else 1.0,
                 _p(dW1), _p(db1), _p(dW2), _p(db2), _p(daws), dab, B, label.shape[1], hid, Cn, heads, _stream())
        return (dqkv, *dws, None, dW1, db1, dW2, db2, e if ctx.e_out else None, None, None, None, None, None)


class _AuxStop(torch.autograd.Function):
    """identity; in the data-gradient-only (aux) sweep the gradient stops here -- placed at the input of the first block that carries
    a domain adapter: nothing upstream of it has adapter parameters, which are all that sweep is run for"""

    @staticmethod
    def forward(ctx, x):
        ctx.set_materialize_grads(False)
        return x.view_as(x)

    @staticmethod
    def backward(ctx, g):
        return None if _dgrad_only else g


def aux_stop(x):
    return _AuxStop.apply(x) if x.requires_grad else x


def factor_att(qkv, crpe_params, H, W_, heads, splits=(2, 3, 3), domain_label=None, da_params=None, aux_first=False):
    """qkv [B,N,3C]; crpe_params = (w3,b3,w5,b5,w7,b7); domain_label [B,D] + da_params (W1,b1,W2,b2) or None -> [B,N,C].
    aux_first: this is the FIRST adapter of the network in forward order -- in the data-gradient-only sweep its backward produces the
    adapter gradient only and hands no gradient on (nothing below carries an adapter)."""
    w3, b3, w5, b5, w7, b7 = crpe_params
    if domain_label is None:
        return _FactorAtt.apply(_c(qkv), w3, b3, w5, b5, w7, b7, None, None, None, None, None, None, int(H), int(W_), int(heads), tuple(splits), False)
    W1, b1, W2, b2 = da_params
    return _FactorAtt.apply(_c(qkv), w3, b3, w5, b5, w7, b7, _c(domain_label.float()), W1, b1, W2, b2, _da_lookup(W2, qkv.shape[0]), int(H), int(W_), int(heads), tuple(splits),
                            bool(aux_first))


def domain_adapter(label, W1, b1, W2, b2, heads):
    """a = softmax_heads(W2 relu(W1 label + b1) + b2); forward only (its gradient path is inside factor_att)."""
    label = _c(label.float())
    B, D = label.shape
    a = _empty((B, W2.shape[0]), device=label.device, dtype=torch.float32)
    call("mdvit_da_fwd", _p(label), _p(W1), _p(b1), _p(W2), _p(b2), _p(a), B, D, W1.shape[0], W2.shape[0], int(heads), _stream())
    return a


# ------------------------------------------------------------------------------------------------
# a whole SerialBlock_adapt pass as ONE C call (csrc/block.hip)
# ------------------------------------------------------------------------------------------------
# The same kernels in the same order as cpe -> layer_norm_fork -> linear -> factor_att -> linear -> layer_norm_fork -> mlp_residual above
# (bit-identical results), but enqueued from C: one autograd node, three torch.empty and one ctypes call per pass instead of ~25
# operator calls.  The host was the limit of the bs=4 step (30-38 ms of enqueue work against 31 ms of main-stream kernel time).
_block_entry = os.environ.get("MDVIT_BLOCK_ENTRY", "1") != "0"
_BLOCK_SINKABLE = (0, 1, 4, 5, 6, 7, 8, 9, 10, 11, 16, 17, 20, 21, 22, 23)        # indices into _lib.BLOCK_PARAMS: cpe, qkv, crpe windows, proj, fc1, fc2
_BLOCK_FRESH = (2, 3, 12, 13, 14, 15, 18, 19)                                      # LayerNorm and adapter gradients: handed to autograd ...
_BLOCK_LN = (2, 3, 18, 19)                                                         # ... unless the LayerNorm parameters have bucket sinks too (ln_accumulate)
_block_ln_sinks = os.environ.get("MDVIT_BLOCK_LN_SINKS", "1") != "0"
_blk_events = None


def _block_streams(side):
    global _blk_events
    if _blk_events is None:
        n = 32
        arr = (C.c_void_p * n)()
        for i in range(n):
            h = C.c_void_p()
            call("mdvit_event_create", C.byref(h))
            arr[i] = h.value
        _blk_events = (arr, C.c_int32(0), n)
    st = _lib.BlockStreams()
    st.main = _stream()
    st.side = C.c_void_p(side.cuda_stream) if side is not None else None
    st.events = C.cast(_blk_events[0], C.POINTER(C.c_void_p))
    st.n_events = _blk_events[2]
    st.next_event = C.pointer(_blk_events[1])
    return st


_side_groups = []       # (event on the side stream behind a block's weight-gradient launches, the stream that owns the tensors, the tensors, bytes); oldest first
_side_held = [0]
# Bound on what the block entries keep allocated for the weight-gradient stream.  Past it the OWNING stream waits (on the GPU) for the side stream to pass the
# oldest blocks and their tensors are released in stream order -- the footprint no longer depends on how far the side stream lags or the host runs ahead
# (bs=32: 52 GiB of such tensors per backward sweep).  0: no bound (release on completed events only).
_side_hold_limit = int(float(os.environ.get("MDVIT_SIDE_HOLD_GIB", "16")) * 2 ** 30)


def _trim_side_groups(limit):
    while _side_groups and _side_groups[0][0].query():
        _side_held[0] -= _side_groups.pop(0)[3]
    while _side_groups and _side_held[0] > limit:
        ev, owner, _ts, n = _side_groups.pop(0)
        owner.wait_event(ev)          # whatever the owner enqueues from here on runs after the side stream's reads: the memory may be reused in stream order
        _side_held[0] -= n


def _side_protect(*tensors, foreign=None):
    """tensors of the current stream that weight-gradient launches on the side stream read: kept allocated (and safe from autograd's in-place accumulation,
    which needs sole ownership) until the side stream has passed those launches -- known either from the completed event or because the owning stream
    was made to wait for it (the hold bound)."""
    ts = [t for t in tensors if t is not None]
    if torch.cuda.is_current_stream_capturing() or not _side_hold_limit:
        for t in ts:
            t.record_stream(_side_stream)
        _side_keepalive.extend(ts)
        marked = _side_blocks[-1][1] if _side_blocks else 0
        if len(_side_keepalive) - marked >= 32 and not torch.cuda.is_current_stream_capturing():
            ev = torch.cuda.Event()
            ev.record(_side_stream)
            _side_blocks.append((ev, len(_side_keepalive)))
            _release_finished_side_blocks()
        return
    if foreign is not None:
        foreign.record_stream(_side_stream)       # the upstream gradient: possibly allocated by another stream's node (see _on_side)
    ev = torch.cuda.Event()
    ev.record(_side_stream)
    n = sum(t.numel() * t.element_size() for t in ts)
    _side_groups.append((ev, current_stream_obj(), ts, n))
    _side_held[0] += n
    _trim_side_groups(_side_hold_limit)


_store_bf16 = os.environ.get("MDVIT_STORE_BF16", "1") != "0"      # 0: fp32 storage in the bf16 mode too (A/B)


def _block_desc(x, label, rs1, rs2, meta, keys, params, backward, a_pre=None):
    H, W_, heads, splits, eps, drop_p, ln_groups = meta[:7]
    B, N, Cn = x.shape
    d = _lib.BlockDesc()
    d.a_pre = _p(a_pre)
    d.B, d.H, d.W, d.C, d.heads, d.hidden = B, H, W_, Cn, heads, params[20].shape[0]
    d.s3, d.s5, d.s7 = splits
    d.attn_kind = int(meta[8]) if len(meta) > 8 else 0          # 1: the DeiT Block_adapt of TransFuse (include/mdvit_hip.h: MdvitBlockDesc.attn_kind)
    d.ln_groups = ln_groups
    d.precision = min(_gemm_precision, 1)
    d.store_bf16 = int(_gemm_precision == 2 and _store_bf16)      # the mixed mode: h / du of the C = 128 MLP as bf16 (include/mdvit_hip.h: MdvitBlockDesc.store_bf16)
    d.eps, d.drop_p = eps, drop_p
    d.key_proj[0], d.key_proj[1] = keys[0]
    d.key_fc1[0], d.key_fc1[1] = keys[1]
    d.key_fc2[0], d.key_fc2[1] = keys[2]
    d.drop_seed = _seed_ptr() if drop_p > 0 else None
    d.rowscale1, d.rowscale2 = _p(rs1), _p(rs2)
    if label is not None:
        d.label, d.D, d.da_hidden = _p(label), label.shape[1], params[12].shape[0]
    for name, t in zip(_lib.BLOCK_PARAMS, params):
        setattr(d, name, _p(t))
    keep = []
    W1, W2 = params[20], params[22]
    rc16 = d.precision == 1 and _mlp_recompute and _mlp_recompute_maxc >= 128 and params[23] is not None and _mlp_rc16_ok(Cn, d.hidden, params[21], W1, W2, B * N)
    if (d.precision == 1 and _mlp_rc_ok(Cn, d.hidden, params[21], params[23], x, W1, W2, B * N)) or rc16:
        p1, p2 = _wplanes(W1, False, 2), _wplanes(W2, False, 2)
        d.fc1_p, d.fc2_p = _p(p1), _p(p2)
        keep += [p1, p2]
        if backward:
            p3, p4 = _wplanes(W2, True, 2), _wplanes(W1, True, 2)
            d.fc2t_p, d.fc1t_p = _p(p3), _p(p4)
            keep += [p3, p4]
        rc = True
    else:
        rc = False
    if _lin_rc_ok(B * N, Cn, Cn) and params[4].is_contiguous() and params[16].is_contiguous():
        ps = [_wplanes(params[4], False, 2), _wplanes(params[16], False, 2)] + ([_wplanes(params[16], True, 2)] if backward else [])
        d.qkv_p, d.proj_p = _p(ps[0]), _p(ps[1])
        if backward:
            d.projt_p = _p(ps[2])
        keep += ps
    if d.precision == 1 and _ph_gemm and Cn >= 256 and not rc and W1.is_contiguous() and W2.is_contiguous() and params[4].is_contiguous() and params[16].is_contiguous():
        # wide blocks: every NT product the 256-wide plane kernel prefers runs on it (block.hip: ph_takes) -- hand in the planes of the weights it may want
        M_ = B * N
        want = [("qkv_p", params[4], False, (M_, 3 * Cn, Cn), 0), ("proj_p", params[16], False, (M_, Cn, Cn), 1), ("fc1_p", W1, False, (M_, d.hidden, Cn), 0),
                ("fc2_p", W2, False, (M_, Cn, d.hidden), 1)]
        if backward:
            want = [("fc2t_p", W2, True, (M_, d.hidden, Cn), 1), ("fc1t_p", W1, True, (M_, Cn, d.hidden), 0), ("projt_p", params[16], True, (M_, Cn, Cn), 0),
                    ("qkvt_p", params[4], True, (M_, Cn, 3 * Cn), 0)]
        for name, Wx, tr, shp, rd in want:
            if _ph_prefers(*shp, 2, rd, plain=backward and not rd):          # (plain: the data gradients block.hip may split along K, pm_split_takes)
                pl = _wplanes(Wx, tr, 2)
                setattr(d, name, _p(pl))
                keep.append(pl)
    if backward and d.precision == 1:
        ts = [wt(params[4]), wt(params[16])] + ([] if rc else [wt(W1), wt(W2)])
        d.qkv_wt, d.proj_wt = _p(ts[0]), _p(ts[1])
        if not rc:
            d.fc1_wt, d.fc2_wt = _p(ts[2]), _p(ts[3])
        keep += ts
    return d, keep


class _SerialBlock(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, label, rs1, rs2, meta, a_pre, *params):
        ctx.set_materialize_grads(False)
        _chk(x, label, rs1, rs2, a_pre, *params)
        drop_p = meta[5]
        keys = tuple(_next_key() if drop_p > 0 else (0, 0) for _ in range(3))          # proj, fc1, fc2: the order of the operator-level path
        if label is None:
            a_pre = None
        ctx.e_out = a_pre is not None and ctx.needs_input_grad[5]          # a is an output of the all-adapters node (_DaMany): e goes back to it, no adapter backward here
        d, keep = _block_desc(x, label, rs1, rs2, meta, keys, params, False, a_pre)
        lib = _lib.load()
        sb, wb = lib.mdvit_block_save_bytes(C.byref(d)), lib.mdvit_block_fwd_ws_bytes(C.byref(d))
        if not sb:
            raise _lib.MdvitHipError("mdvit_block_save_bytes: " + lib.mdvit_last_error().decode(errors="replace"))
        save = _empty((sb // 4,), device=x.device, dtype=torch.float32)
        ws = _empty((wb // 4,), device=x.device, dtype=torch.float32)
        y = _empty_like(x)
        call("mdvit_block_fwd", C.byref(d), _p(x), _p(y), _p(save), sb, _p(ws), wb, _stream())
        del keep
        ctx.save_for_backward(x, save, label, rs1, rs2, *params)
        ctx.meta, ctx.keys, ctx.a_pre = meta, keys, a_pre
        return y

    @staticmethod
    def backward(ctx, g):
        n_in = 6 + len(_lib.BLOCK_PARAMS)
        if g is None:
            return (None,) * n_in
        x, save, label, rs1, rs2, *params = ctx.saved_tensors
        meta = ctx.meta
        aux_first = meta[7]
        g = _c(g)
        dev = x.device
        d, keep = _block_desc(x, label, rs1, rs2, meta, ctx.keys, params, True, ctx.a_pre)
        G = _lib.BlockGrads()
        G.dgrad_only, G.aux_first = int(_dgrad_only), int(aux_first)
        want_w = not _dgrad_only
        out = [None] * len(params)
        acc = False
        if want_w:
            sinks = [_sink_of(params[i]) if params[i] is not None else None for i in _BLOCK_SINKABLE]
            acc = _side_stream is not None and all((s is not None) or (params[i] is None) for s, i in zip(sinks, _BLOCK_SINKABLE))
        ln_sinks = [_sink_of(params[i]) for i in _BLOCK_LN] if (acc and _block_ln_sinks and d.C in (64, 128, 320, 512)) else None
        ln_acc = ln_sinks is not None and all(s_ is not None for s_ in ln_sinks)
        e_out = None
        if ctx.e_out:
            e_out = _empty((x.shape[0], d.C), device=dev, dtype=torch.float32)
            G.e_out = _p(e_out)
        fresh_idx = [i for i in _BLOCK_FRESH if params[i] is not None and ((e_out is None) if 12 <= i <= 15 else want_w) and not (ln_acc and i in _BLOCK_LN)]
        if want_w and not acc:
            fresh_idx = sorted(fresh_idx + [i for i in _BLOCK_SINKABLE if params[i] is not None])
        bufs = _flat_like(*[params[i] for i in fresh_idx]) if fresh_idx else []
        for i, b in zip(fresh_idx, bufs):
            out[i] = b
            setattr(G, _lib.BLOCK_PARAMS[i], _p(b))
        if acc:
            for s_, i in zip(sinks, _BLOCK_SINKABLE):
                if s_ is not None:
                    setattr(G, _lib.BLOCK_PARAMS[i], _p(s_))
        G.accumulate = int(acc)
        if ln_acc:
            # the LayerNorm gradients (and the fc2 bias sums) ADD into their buckets too: their second-stage reductions leave the data-gradient chain
            # and run on the weight-gradient stream
            for s_, i in zip(ln_sinks, _BLOCK_LN):
                setattr(G, _lib.BLOCK_PARAMS[i], _p(s_))
            G.ln_accumulate = 1
        stop_here = _dgrad_only and aux_first and label is not None
        dx = _empty_like(x) if (ctx.needs_input_grad[0] and not stop_here) else None
        side = _side_stream if (acc and want_w) else None
        st = _block_streams(side)
        sbytes = C.c_size_t(0)
        wb = _lib.load().mdvit_block_bwd_ws_bytes(C.byref(d), C.byref(G), int(side is not None), C.byref(sbytes))
        ws = _empty((wb // 4,), device=dev, dtype=torch.float32)                  # main-stream temporaries: released when this function returns
        ws_side = _empty((sbytes.value // 4,), device=dev, dtype=torch.float32)   # what the weight-gradient kernels read: kept until the side stream is done
        call("mdvit_block_bwd", C.byref(d), C.byref(G), C.byref(st), _p(x), _p(save), save.numel() * 4, _p(g), _p(dx), _p(ws), wb, _p(ws_side), sbytes.value)
        if side is not None:
            _side_protect(ws_side, save, g, x, *keep, foreign=g)
        del keep
        return (dx, None, None, None, None, e_out, *out)


def serial_block(x, label, rs1, rs2, meta, params):
    """x [B, N, C] -> [B, N, C]: one SerialBlock_adapt (mdvit.py:346-361).  meta = (H, W, heads, head_splits, eps, drop_p, ln_groups, aux_first);
    params in _lib.BLOCK_PARAMS order (the adapter's four are None without a domain label)."""
    a_pre = _da_lookup(params[14], x.shape[0]) if label is not None else None          # params[14] = da_w2
    return _SerialBlock.apply(_c(x), None if label is None else _c(label.float()), rs1, rs2, meta, a_pre, *params)


_deit_block_entry = os.environ.get("MDVIT_DEIT_BLOCK_ENTRY", "1") != "0"      # TransFuse's DeiT blocks through the C-level block entry (0: the operator-level path, A/B)


def deit_block_entry_ok(x, heads, params) -> bool:
    """the C-level entry covers this Block_adapt (vision_transformer.py:191-211): 256 tokens, head dimension 64, <= 6 heads (the fp32-matrix-core attention of sdpa.hip),
    the parity arithmetic (fp32 / bf16x3), contiguous fp32 parameters"""
    B, N, Cn = x.shape
    return (_block_entry and _deit_block_entry and _use_mfma_sdpa_entry and x.is_cuda and torch.is_grad_enabled() and N == 256 and heads <= 6 and Cn == heads * 64
            and _gemm_precision <= 1 and all(p is None or (p.is_contiguous() and p.dtype == torch.float32) for p in params))


_use_mfma_sdpa_entry = os.environ.get("MDVIT_SDPA_MFMA", "1") != "0"


def block_entry_ok(Cn, hidden, params) -> bool:
    """the C-level block covers this configuration (otherwise: the operator-level path)"""
    if not _block_entry or Cn % 4:
        return False
    if _gemm_precision > 1 and Cn > 128:
        return False           # bf16 mode: the MFMA-bound C >= 320 blocks stay on the operator path (single-plane GEMMs); the HBM- / VALU-bound C <= 128
        #                        blocks run the same bf16x3 register-chained kernels as the parity mode (the "mixed" mode of BASELINE configs[3])
    if not all(p is None or (p.is_contiguous() and p.dtype == torch.float32) for p in params):
        return False
    if Cn == 64 and _gemm_precision >= 1 and not (_mlp_rc and _mlp_recompute and hidden % 256 == 0 and hidden <= 4096):
        return False                                   # round 2's fused C = 64 MLP kernels are reachable through the operator path only
    if _gemm_precision >= 1 and Cn <= 128 and not _mlp_recompute:
        return False
    return True


# ------------------------------------------------------------------------------------------------
# fused step losses on logits
# ------------------------------------------------------------------------------------------------
# Data parallelism: losses over the global batch (see mdvit_seg_losses_sums in the header).  On by default whenever a
# process group with more than one rank is initialised; set_global_batch_losses(False) gives per-rank losses instead.
_loss_global = True
_loss_group = None


def set_global_batch_losses(flag: bool = True, group=None):
    global _loss_global, _loss_group
    _loss_global, _loss_group = bool(flag), group


_force_collectives = False       # tests: take the collective code paths even in a 1-rank process group


def _loss_world() -> int:
    import torch.distributed as dist
    if not (_loss_global and dist.is_available() and dist.is_initialized()):
        return 1
    return dist.get_world_size(_loss_group)


class _SegLosses(torch.autograd.Function):
    """Three scalar outputs so that a backward() that involves only some of them (the aux sweep) hands back
    None -- not a zero tensor -- for the logits that take no part (multi_train_MDViT.py:201)."""

    @staticmethod
    def forward(ctx, out, aux, label):
        _chk(out, aux, label)
        n = out.numel()
        sums = _empty((16,), device=out.device, dtype=torch.float64)
        losses = _empty((3,), device=out.device, dtype=torch.float32)
        world = _loss_world()
        if world > 1 or (_force_collectives and torch.distributed.is_initialized()):
            # nn.DataParallel semantics: BCE / Dice over the GLOBAL batch of the domain (the replicas' outputs gathered)
            call("mdvit_seg_losses_sums", _p(out), _p(aux), _p(label), _p(sums), n, _stream())
            torch.distributed.all_reduce(sums, op=torch.distributed.ReduceOp.SUM, group=_loss_group)
            call("mdvit_seg_losses_final", _p(sums), _p(losses), n * world, int(aux is not None), _stream())
        else:
            call("mdvit_seg_losses_fwd", _p(out), _p(aux), _p(label), _p(sums), _p(losses), n, _stream())
        ctx.dice_gain = float(world)
        ctx.save_for_backward(out, aux, label, sums)
        ctx.set_materialize_grads(False)
        return losses[0], losses[1], losses[2]          # (three views of one fresh buffer: nothing writes them in place; the clones were three copy launches per domain)

    @staticmethod
    def backward(ctx, g0, g1, g2):
        out, aux, label, sums = ctx.saved_tensors
        n = out.numel()
        gs = [None if gi is None else _c(gi.reshape(()).float()) for gi in (g0, g1, g2)]      # NULL = that loss takes no part (was: zeros + stack, two launches per domain and sweep)
        need_out = ctx.needs_input_grad[0] and (g0 is not None or g2 is not None)
        need_aux = aux is not None and ctx.needs_input_grad[1] and (g1 is not None or g2 is not None)
        dout = _empty_like(out) if need_out else None
        daux = _empty_like(aux) if need_aux else None
        if need_out or need_aux:
            call("mdvit_seg_losses_bwd3", _p(out), _p(aux), _p(label), _p(sums), _p(gs[0]), _p(gs[1]), _p(gs[2]), _p(dout), _p(daux), n, ctx.dice_gain, _stream())
        return dout, daux, None


class _SegLossesGroups(torch.autograd.Function):
    """_SegLosses over the G domain batches of a domain-batched forward, summed over the batches: one sums launch, one final, one backward launch (it was one
    of each per domain plus the additions of the per-domain losses and a concatenation of the per-domain logit gradients)."""

    @staticmethod
    def forward(ctx, out, aux, label, G):
        _chk(out, aux, label)
        n = out.numel() // G
        sums = _empty((16 * G,), device=out.device, dtype=torch.float64)
        losses = _empty((3,), device=out.device, dtype=torch.float32)
        world = _loss_world()
        call("mdvit_seg_losses_groups_sums", _p(out), _p(aux), _p(label), _p(sums), n, G, _stream())
        if world > 1 or (_force_collectives and torch.distributed.is_initialized()):
            torch.distributed.all_reduce(sums, op=torch.distributed.ReduceOp.SUM, group=_loss_group)      # BCE / Dice over the GLOBAL batch of every domain
        call("mdvit_seg_losses_groups_final", _p(sums), _p(losses), None, n * world, int(aux is not None), G, _stream())
        ctx.dice_gain = float(world)
        ctx.G = G
        ctx.save_for_backward(out, aux, label, sums)
        ctx.set_materialize_grads(False)
        return losses[0], losses[1], losses[2]

    @staticmethod
    def backward(ctx, g0, g1, g2):
        out, aux, label, sums = ctx.saved_tensors
        gs = [None if gi is None else _c(gi.reshape(()).float()) for gi in (g0, g1, g2)]
        need_out = ctx.needs_input_grad[0] and (g0 is not None or g2 is not None)
        need_aux = aux is not None and ctx.needs_input_grad[1] and (g1 is not None or g2 is not None)
        dout = _empty_like(out) if need_out else None
        daux = _empty_like(aux) if need_aux else None
        if need_out or need_aux:
            call("mdvit_seg_losses_groups_bwd", _p(out), _p(aux), _p(label), _p(sums), _p(gs[0]), _p(gs[1]), _p(gs[2]), _p(dout), _p(daux), out.numel() // ctx.G, ctx.G,
                 ctx.dice_gain, _stream())
        return dout, daux, None, None


def seg_losses_groups(out, aux, label, G: int):
    """the three losses of G equal consecutive domain batches, each a mean over ITS batch (multi_train_MDViT.py:147-153), summed over the batches"""
    if G <= 1:
        return seg_losses(out, aux, label)
    if out.shape[0] % G:
        raise _lib.MdvitHipError(f"seg_losses_groups: batch {out.shape[0]} is not {G} equal domain batches")
    return _SegLossesGroups.apply(_c(out), None if aux is None else _c(aux), _c(label.float()), int(G))


def seg_losses(out, aux, label):
    """logits -> (BCE+Dice(out), BCE+Dice(aux), Dice(sigmoid(aux), sigmoid(out))) as three 0-dim tensors."""
    return _SegLosses.apply(_c(out), None if aux is None else _c(aux), _c(label.float()))


# ------------------------------------------------------------------------------------------------
# on-device metrics and input pipeline (no autograd)
# ------------------------------------------------------------------------------------------------
def seg_metrics(out, aux, label):
    """logits, logits | None, label -> (metrics [4] f32 = dice, iou, aux dice, aux iou; counts [8] int64) on the device:
    the thresholded-output Dice / IoU of multi_train_MDViT.py:172-179 without a device-to-host copy per domain."""
    out, label = _c(out.detach()), _c(label.detach().float())
    aux = None if aux is None else _c(aux.detach())
    _chk(out, aux, label)
    counts = torch.empty((8,), device=out.device, dtype=torch.int64)
    metrics = torch.empty((4,), device=out.device, dtype=torch.float32)
    call("mdvit_seg_metrics", _p(out), _p(aux), _p(label), _p(counts), _p(metrics), out.numel(), _stream())
    return metrics, counts


def image_normalize_u8(img_u8_nhwc):
    """uint8 [B,H,W,3] on the device -> ImageNet-normalised fp32 [B,3,H,W] (the loader's norm01 + permute + Normalize)."""
    if img_u8_nhwc.dtype != torch.uint8 or img_u8_nhwc.dim() != 4 or img_u8_nhwc.shape[-1] != 3 or not img_u8_nhwc.is_cuda:
        raise _lib.MdvitHipError("image_normalize_u8 expects a CUDA uint8 tensor [B,H,W,3]")
    x = _c(img_u8_nhwc)
    B, H, W_, _ = x.shape
    y = torch.empty((B, 3, H, W_), device=x.device, dtype=torch.float32)
    call("mdvit_image_normalize_u8", _p(x), _p(y), B, H, W_, _stream())
    return y


# ------------------------------------------------------------------------------------------------
# a whole backward sweep on a stream of its own
# ------------------------------------------------------------------------------------------------
# Autograd runs a node's backward on the stream its forward ran on, so two sweeps over the same graph queue behind each other.  With
# set_sweep_stream(s) every backward of THIS module's Functions switches torch's current stream to s for its duration: a sweep
# whose nodes are all ours (ops.fork at every tensor with several consumers, an explicit `gradient=` at the root) then runs on s
# from end to end, next to the sweep that stays on the main stream (train.mdvit_train_step: the data-gradient-only aux sweep).
_sweep_stream = None


def set_sweep_stream(stream):
    global _sweep_stream
    _sweep_stream = stream


_sweep_stream_obj = None
_ones = {}
# Dispatch priority of the data-gradient-only sweep's stream (torch: 0 normal, -1 high).  Measured with every stream on a hardware queue of its own
# (tools/probe/main_priority_probe.py): priorities move the step by < 1 % either way (aux high 37.1-37.2 ms, main high 36.7-36.9, both normal 37.3-38.0
# on one box).  (With FIVE streams on four queues a high-priority aux stream took the step from 40.7 to 51.8 ms.)
_aux_priority = int(os.environ.get("MDVIT_AUX_PRIORITY", "0"))


# A replayed HIP graph runs its parallel branches concurrently (tools/probe/graph_branch_probe.py: two captured streams of ten 100-us kernels replay in
# 1.05 ms, 2.01 ms on one stream), so the whole-step graph may keep the aux sweep's stream: 1 = fork it inside the capture too
_graph_streams = os.environ.get("MDVIT_GRAPH_STREAMS", "0") != "0"


def sweep_stream():
    """the stream the second backward sweep runs on (created once); None while a HIP graph is being captured"""
    global _sweep_stream_obj
    if torch.cuda.is_current_stream_capturing() and not _graph_streams:
        return None
    if _sweep_stream_obj is None:
        _sweep_stream_obj = torch.cuda.Stream(priority=_aux_priority)
    return _sweep_stream_obj


def reserve_streams(side: bool = True, sweep: bool = True, branch: bool = False):
    """Create the train step's extra streams NOW and run a first (empty-sized) kernel on each, so that they bind their hardware queues before anyone
    else's streams do.  The GPU has four hardware queues; a stream that binds later shares one and runs BEHIND its owner's work.  Call this right after
    torch.cuda.set_device(...) and BEFORE torch.distributed.init_process_group("nccl", device_id=...): RCCL's communicator brings streams of its own,
    and when they come first the aux-sweep stream shares the main stream's queue -- the data-gradient-only sweep then runs after the full sweep
    instead of next to it (tools/probe/rccl_timeline_probe.py: 40.2 ms per step instead of 36)."""
    sts = []
    if side:
        if _side_stream is None:
            enable_side_stream(True)
        sts.append(_side_stream)
    if sweep:
        sts.append(sweep_stream())
    if branch:
        sts.append(branch_stream())
    dev = torch.device("cuda", torch.cuda.current_device())
    torch.zeros(256, device=dev)                          # (the main stream's queue first)
    for st in sts:
        if st is not None:
            with torch.cuda.stream(st):
                torch.zeros(256, device=dev)
    torch.cuda.synchronize()


# One device per process: the autograd engine's hand-off of a sweep to its per-device worker thread buys nothing and costs a thread
# switch per sweep plus cross-thread stream bookkeeping; the sweeps run on the calling thread (+1-2 % on the bs=4 step).
_autograd_mt = os.environ.get("MDVIT_AUTOGRAD_MT", "0") != "0"


def backward(loss, **kw):
    """loss.backward(**kw) on the CALLING thread (MDVIT_AUTOGRAD_MT=1: on autograd's worker thread, torch's default)"""
    if _autograd_mt:
        loss.backward(**kw)
    else:
        with torch.autograd.set_multithreading_enabled(False):
            loss.backward(**kw)


def one_like(t):
    """a cached tensor of ones shaped like t: the explicit root gradient of a sweep (autograd would fill a fresh one on the main stream)"""
    key = (t.device, tuple(t.shape), t.dtype)
    o = _ones.get(key)
    if o is None:
        o = _ones[key] = torch.ones_like(t)
    return o


def _install_sweep_stream_override():
    def wrap(orig):
        def backward(ctx, *gs):
            st = _sweep_stream
            if st is None:
                return orig(ctx, *gs)
            with use_stream(st):
                return orig(ctx, *gs)
        backward.__wrapped__ = orig
        return backward
    for _name, cls in list(globals().items()):
        if isinstance(cls, type) and issubclass(cls, torch.autograd.Function) and cls is not torch.autograd.Function:
            cls.backward = staticmethod(wrap(cls.backward))


_install_sweep_stream_override()
