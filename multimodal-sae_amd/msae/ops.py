"""PyTorch-ROCm custom ops (``torch.ops.msae.*``) over the C ABI of libmsae_hip.so.

These are the operators the drop-in ``Sae`` module (msae/sae/sae.py) and the feature cache
(msae/features/cache.py) are built from.  Reference call sites they replace:

    msae::pre_acts      nn.Linear + relu                     sae_auto_interp/sae/sae.py:172-177
    msae::topk          Tensor.topk                          sae/sae.py:179-181, features/cache.py:210
    msae::encode_topk   Sae.encode, fused (+ hook edits)     sae/sae.py:183-185, steering.py:113-114
    msae::decode        decoder_impl / TritonDecoder.apply   sae/utils.py:107-129, kernels.py:403-429
    msae::sparsify      scatter_ + Cache.get_nonzeros/add    features/cache.py:214-217,42-92

All ops run on the tensor's device on the current stream, never synchronise (``sparsify`` reads one
int64 back, as ``torch.nonzero`` does), and raise on CPU tensors.
"""
from __future__ import annotations

import collections
import contextlib
import ctypes
import os
import weakref
from typing import Optional, Tuple

import torch
from torch import Tensor

from . import _hip

_WS: "collections.OrderedDict" = collections.OrderedDict()
_WS_MAX_STREAMS = 8      # scratch buffers kept alive: the most recently used (device, stream) pairs


def _workspace(dev: torch.device, nbytes: int) -> Tensor:
    """Grow-only scratch buffer per (device, stream) -- calls enqueued on different streams may run
    concurrently and must not share scratch (uint8, 256-B aligned by the caching allocator).  At most
    _WS_MAX_STREAMS buffers stay alive (least recently used first out; a dropped buffer goes back to the
    caching allocator, which keeps it valid for the kernels already enqueued on its stream)."""
    key = (dev, torch.cuda.current_stream(dev).cuda_stream)
    ws = _WS.pop(key, None)
    if ws is None or ws.numel() < nbytes:
        ws = None
        ws = torch.empty(max(nbytes, 1 << 20), dtype=torch.uint8, device=dev)
    _WS[key] = ws                      # most recently used last
    while len(_WS) > _WS_MAX_STREAMS:
        _WS.popitem(last=False)
    return ws


_WS_EPOCH = [0]


def workspace_epoch() -> int:
    """Bumped whenever cached scratch buffers are dropped: anything that captured their addresses (a HIP graph of the hooks'
    S = 1 step) re-captures."""
    return _WS_EPOCH[0]


def release_workspaces() -> None:
    """Drop every cached scratch buffer (they are re-created on demand)."""
    _WS_EPOCH[0] += 1
    _WS.clear()
    _CERT_CACHE.clear()
    _OPTS_CACHE.clear()


# ---- per-call options of the fused encoder (struct msae_options) ------------------------------------------
# The shared library keeps no state: coarse mode, band width, diagnostics and the stage profile travel with
# every call.  `Options` is the host-side mirror; `_defaults` is what the ops of THIS Python process use when a
# call does not bring its own (a convenience of the host layer, set by set_coarse_mode() & co.).
class StageProfile:
    """Handle of msae_profile_create: HIP events at the stage boundaries of every fused encode that carries it."""

    STAGES = 6

    def __init__(self, max_steps: int):
        self.max_steps = max_steps
        self._h = ctypes.c_void_p()
        _hip.check(_hip.load().msae_profile_create(max_steps, ctypes.byref(self._h)), "msae_profile_create")

    @property
    def handle(self):
        return self._h

    def read(self):
        """-> float32 array [n_steps, 6] of stage times in ms; restarts the handle."""
        import numpy as np

        buf = (ctypes.c_float * (self.max_steps * self.STAGES))()
        n = ctypes.c_int(0)
        _hip.check(_hip.load().msae_profile_read(self._h, buf, ctypes.byref(n)), "msae_profile_read")
        return np.array(buf[:], dtype=np.float32).reshape(self.max_steps, self.STAGES)[: n.value]

    def close(self):
        if self._h:
            _hip.load().msae_profile_destroy(self._h)
            self._h = ctypes.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Options:
    COARSE = {"default": -1, "bf16": 0, "int8": 1, "fp8": 2}
    DITHER = {"default": 0, "on": 1, "off": 2}

    def __init__(self, coarse: str = "default", guard_z: float = 0.0, status_detail: bool = False,
                 profile: Optional[StageProfile] = None, exact: bool = False, dither: str = "default",
                 dither_seed: int = 0):
        self.coarse, self.guard_z, self.status_detail, self.profile = coarse, guard_z, status_detail, profile
        self.exact = exact
        self.dither, self.dither_seed = dither, int(dither_seed)   # msae_options::dither / dither_seed (0: drawn per call)
        self.rows_rescored: Optional[Tensor] = None     # device int32 [T] (msae_options::rows_rescored) or None
        self.certified = False                          # msae_options::certified (+ the operand buffer of the call)
        self.certified_operands: Optional[Tensor] = None

    def struct(self) -> "_hip.MsaeOptions":
        o = _hip.MsaeOptions()
        o.size = ctypes.sizeof(_hip.MsaeOptions)
        o.coarse_mode = self.COARSE[self.coarse]
        o.guard_z = float(self.guard_z)
        o.status_detail = int(bool(self.status_detail))
        o.profile = self.profile.handle if self.profile is not None else None
        o.exact = int(bool(self.exact))
        o.dither = self.DITHER[self.dither]
        o.dither_seed = self.dither_seed & 0xFFFFFFFFFFFFFFFF
        o.rows_rescored = self.rows_rescored.data_ptr() if self.rows_rescored is not None else None
        o.certified = int(bool(self.certified))
        o.certified_operands = self.certified_operands.data_ptr() if self.certified_operands is not None else None
        return o

    def ref(self):
        self._keep = self.struct()        # keeps the struct alive for the duration of the call
        return ctypes.byref(self._keep)


_defaults = Options()


class _OptsRef:
    """A filled msae_options struct and its byref, built once per distinct content (the S = 1 latency path is
    host-bound: two ctypes structs per call showed up as +10 us per decode step)."""
    __slots__ = ("struct", "_ref", "profile", "cert_ops")

    def __init__(self, o: Options):
        self.struct, self.profile = o.struct(), o.profile          # (keeps the profile handle alive)
        self.cert_ops = o.certified_operands                       # (... and the certified operand buffer)
        self._ref = ctypes.byref(self.struct)

    def ref(self):
        return self._ref


_OPTS_CACHE: dict = {}
_WS_BYTES_CACHE: dict = {}


_DITHER_NAME = {0: "default", 1: "on", 2: "off"}


def _opts(coarse_mode: int = -1, guard_z: float = 0.0, status_detail: bool = False, exact: bool = False,
          dither: int = 0, dither_seed: int = 0, cert_ops: Optional[Tensor] = None) -> _OptsRef:
    """Options of one call: explicit arguments win, the process defaults fill the rest."""
    coarse = _defaults.coarse if coarse_mode < 0 else {0: "bf16", 1: "int8", 2: "fp8"}[coarse_mode]
    z = guard_z if guard_z > 0.0 else _defaults.guard_z
    detail = bool(status_detail or _defaults.status_detail)
    exact = bool(exact or _defaults.exact)
    dith = _DITHER_NAME[dither] if dither else _defaults.dither
    seed = dither_seed if dither_seed else _defaults.dither_seed
    prof, rows = _defaults.profile, _defaults.rows_rescored
    if cert_ops is not None:
        # a certified call brings its 2 N d-byte operand buffer: built per call and never cached -- a cache entry would pin the
        # buffer of every weight version a training / evaluation loop has been through; the call is milliseconds
        o = Options(coarse, z, detail, prof, exact, dith, seed)
        o.certified, o.certified_operands = True, cert_ops
        o.rows_rescored = rows
        return _OptsRef(o)
    key = (coarse, z, detail, id(prof) if prof is not None else 0, exact, rows.data_ptr() if rows is not None else 0,
           dith, seed)
    ref = _OPTS_CACHE.get(key)
    if ref is None or ref.profile is not prof:
        if len(_OPTS_CACHE) > 64:
            _OPTS_CACHE.clear()
        o = Options(coarse, z, detail, prof, exact, dith, seed)
        o.rows_rescored = rows
        ref = _OPTS_CACHE[key] = _OptsRef(o)
    return ref


def _encode_ws_bytes(lib, T: int, d: int, N: int, k: int, opts: _OptsRef) -> int:
    """msae_encode_topk_ws_bytes, memoised (a pure function of the shape and the coarse mode)."""
    key = (T, d, N, k, opts.struct.coarse_mode, os.environ.get("MSAE_COARSE") if opts.struct.coarse_mode < 0 else None,
           os.environ.get("MSAE_FM"), opts.struct.certified)
    n = _WS_BYTES_CACHE.get(key)
    if n is None:
        if len(_WS_BYTES_CACHE) > 4096:
            _WS_BYTES_CACHE.clear()
        n = _WS_BYTES_CACHE[key] = lib.msae_encode_topk_ws_bytes(T, d, N, k, opts.ref())
    return n


@contextlib.contextmanager
def profiling(profile: StageProfile):
    """Fused encodes issued inside the block record their stage boundaries into `profile`."""
    prev, _defaults.profile = _defaults.profile, profile
    try:
        yield profile
    finally:
        _defaults.profile = prev


@contextlib.contextmanager
def rescore_rows(buf: Tensor):
    """Fused encodes of <= buf.numel() tokens issued inside the block write their per-token re-score statistics
    (msae_options::rows_rescored: [bit 30: first round feature-major] | rounds << 24 | first-round rows << 12 | rows of W_enc
    read; 0 = not verified by the large-batch re-score) into `buf` (device int32)."""
    assert buf.is_cuda and buf.dtype == torch.int32 and buf.is_contiguous()
    prev, _defaults.rows_rescored = _defaults.rows_rescored, buf
    try:
        yield buf
    finally:
        _defaults.rows_rescored = prev


@contextlib.contextmanager
def _without_rows():
    prev, _defaults.rows_rescored = _defaults.rows_rescored, None
    try:
        yield
    finally:
        _defaults.rows_rescored = prev


def _opts_for(T: int, dev) -> "_OptsRef":
    """The process defaults as the options of a call of T tokens on `dev` (see encode_topk: the statistics buffer only
    where it is large enough)."""
    rows = _defaults.rows_rescored
    if rows is not None and (T > rows.numel() or rows.device != dev):
        with _without_rows():
            return _opts()
    return _opts()


_DEBUG_BOUNDS = os.environ.get("MSAE_DEBUG_BOUNDS", "0") not in ("", "0")


def set_debug_bounds(on: bool) -> None:
    """Debug mode (environment MSAE_DEBUG_BOUNDS=1): decode and its backward read the kernels' out-of-range flag
    back after the call (one host synchronisation per call) and raise IndexError -- the analogue of the reference's
    tl.device_assert(i < N) (sae/kernels.py:276,389).  Off: out-of-range indices are skipped silently."""
    global _DEBUG_BOUNDS
    _DEBUG_BOUNDS = bool(on)


def _bounds_flag(dev) -> Optional[Tensor]:
    return torch.zeros(1, dtype=torch.int32, device=dev) if _DEBUG_BOUNDS else None


def _check_bounds(flag: Optional[Tensor], what: str) -> None:
    if flag is not None and int(flag.item()) != 0:
        raise IndexError(f"{what}: a latent index lies outside [0, num_latents)")


def _f32c(t: Optional[Tensor]) -> Optional[Tensor]:
    if t is None:
        return None
    return t.detach().to(torch.float32).contiguous()


def _act(x: Tensor) -> Tensor:
    x = x.detach()
    if x.dtype not in _hip.DTYPE_CODE:
        x = x.to(torch.float32)
    return x.contiguous()


# ------------------------------------------------------------------------------------------------
@torch.library.custom_op("msae::pre_acts", mutates_args=())
def pre_acts(x: Tensor, W_enc: Tensor, b_enc: Optional[Tensor], b_dec: Optional[Tensor]) -> Tensor:
    """relu((x - b_dec) @ W_enc.T + b_enc) -> dense [..., N] f32 (exact f32 MFMA path)."""
    dev = _hip.require_device(x, W_enc, b_enc, b_dec)
    lib = _hip.load()
    xa, W, be, bd = _act(x), _f32c(W_enc), _f32c(b_enc), _f32c(b_dec)
    N, d = W.shape
    assert xa.shape[-1] == d, f"x last dim {xa.shape[-1]} != d_in {d}"
    T = xa.numel() // d
    out = torch.empty(*xa.shape[:-1], N, dtype=torch.float32, device=dev)
    with torch.cuda.device(dev):
        _hip.check(lib.msae_pre_acts_f32(_hip.ptr(xa), _hip.DTYPE_CODE[xa.dtype], _hip.ptr(W),
                                         _hip.ptr(be), _hip.ptr(bd), T, d, N, 1, _hip.ptr(out),
                                         _hip.stream_of(xa)), "msae_pre_acts_f32")
    return out


@pre_acts.register_fake
def _(x, W_enc, b_enc, b_dec):
    return x.new_empty(*x.shape[:-1], W_enc.shape[0], dtype=torch.float32)


def _dense_gemm_nt(a: Tensor, b: Tensor) -> Tensor:
    """a [M, K] @ b[P, K]^T -> [M, P] f32 on the exact f32 MFMA kernel (msae_pre_acts_f32 without bias / ReLU)."""
    lib = _hip.load()
    a, b = a.contiguous(), b.contiguous()
    out = torch.empty(a.shape[0], b.shape[0], dtype=torch.float32, device=a.device)
    if out.numel():
        with torch.cuda.device(a.device):
            _hip.check(lib.msae_pre_acts_f32(_hip.ptr(a), 0, _hip.ptr(b), None, None, a.shape[0], a.shape[1],
                                             b.shape[0], 0, _hip.ptr(out), _hip.stream_of(a)), "msae_pre_acts_f32")
    return out


def _pre_acts_setup(ctx, inputs, output):
    x, W_enc, b_enc, b_dec = inputs
    ctx.save_for_backward(x, W_enc, b_dec, output)
    ctx.has = (b_enc is not None, b_dec is not None)


def _pre_acts_backward(ctx, grad_out):
    """The dense graph of the reference's nn.Linear + relu (sae.py:172-177): legacy callers that differentiate
    pre_acts -> [mask] -> select_topk -> decode (features/patching/utils.py:41-49, tools/*.py) get the same
    gradients as from torch autograd -- two dense GEMMs on the exact f32 MFMA kernel.  (The fused
    Sae.encode(zero_feature=...) path has the sparse backward and never builds [T, N] gradients.)"""
    x, W_enc, b_dec, out = ctx.saved_tensors
    N, d = W_enc.shape
    g = (grad_out.reshape(-1, N).float() * (out.reshape(-1, N) > 0)).contiguous()       # relu'
    need_x, need_W, need_be, need_bd = ctx.needs_input_grad
    g_x = g_W = g_be = g_bd = None
    if need_x or (need_bd and ctx.has[1]):
        da = _dense_gemm_nt(g, W_enc.detach().float().t())                 # [T, N] @ [N, d]
        g_x = da.view(x.shape).to(x.dtype) if need_x else None
        g_bd = -da.sum(0) if (need_bd and ctx.has[1]) else None
    if need_W:
        a = x.detach().reshape(-1, d).float()
        if ctx.has[1]:
            a = a - b_dec.detach().float()
        g_W = _dense_gemm_nt(g.t(), a.t()).to(W_enc.dtype)                 # [N, T] @ [T, d]
    if need_be and ctx.has[0]:
        g_be = g.sum(0)
    return g_x, g_W, g_be, g_bd


pre_acts.register_autograd(_pre_acts_backward, setup_context=_pre_acts_setup)


@torch.library.custom_op("msae::topk", mutates_args=())
def topk(latents: Tensor, k: int) -> Tuple[Tensor, Tensor]:
    """Canonical top-k along the last dim: values descending, ties by ascending index; int64 idx."""
    dev = _hip.require_device(latents)
    lib = _hip.load()
    lat = _f32c(latents)
    N = lat.shape[-1]
    T = lat.numel() // N
    vals = torch.empty(*lat.shape[:-1], k, dtype=torch.float32, device=dev)
    idx = torch.empty(*lat.shape[:-1], k, dtype=torch.int32, device=dev)
    with torch.cuda.device(dev):
        _hip.check(lib.msae_topk_f32(_hip.ptr(lat), T, N, k, _hip.ptr(vals), _hip.ptr(idx), None, 0,
                                     _hip.stream_of(lat)), "msae_topk_f32")
    return vals, idx.to(torch.int64)


@topk.register_fake
def _(latents, k):
    return (latents.new_empty(*latents.shape[:-1], k, dtype=torch.float32),
            latents.new_empty(*latents.shape[:-1], k, dtype=torch.int64))


def _topk_setup(ctx, inputs, output):
    ctx.save_for_backward(output[1])
    ctx.in_shape, ctx.in_dtype = inputs[0].shape, inputs[0].dtype


def _topk_backward(ctx, g_vals, _g_idx):
    """Tensor.topk's backward: the values' gradient scattered to the selected positions (sae.py:179-181)."""
    idx, = ctx.saved_tensors
    g = torch.zeros(ctx.in_shape, dtype=g_vals.dtype, device=g_vals.device)
    g.scatter_(-1, idx, g_vals)
    return g.to(ctx.in_dtype), None


topk.register_autograd(_topk_backward, setup_context=_topk_setup)


def set_coarse_mode(mode: str) -> None:
    """Default operand type of the fused encoder's candidate pass for this process's ops: "int8", "bf16", "fp8" (e4m3
    operands -- BASELINE configs[4]; its prepared operands take the int8 ones' place, so prepare under the mode you encode in) or
    "default" (environment MSAE_COARSE, else int8).  Outputs do not depend on it (candidates are re-scored
    exactly); speed does.  A host-layer default: the library itself takes the mode per call (msae_options)."""
    if mode not in Options.COARSE:
        raise ValueError(f"coarse mode {mode!r}: expected one of {sorted(Options.COARSE)}")
    _defaults.coarse = mode


def set_guard_z(z: float) -> None:
    """Default width of the candidate pass's error band in standard deviations of its per-(token, feature)
    rounding noise (library default 7; 0 restores it).  Verified results do not depend on it; rows read per token do."""
    if z != 0.0 and not (0.25 <= z <= 64.0):
        raise ValueError("guard z must lie in [0.25, 64] (or 0 for the default)")
    _defaults.guard_z = float(z)


def set_exact(on: bool) -> None:
    """Default of msae_options::exact for this process's ops: every fused encode computes every token by the exact path
    (no statistical contract; ~20x slower on large batches)."""
    _defaults.exact = bool(on)


def set_dither(mode: str = "default", seed: int = 0) -> None:
    """Default of msae_options::dither / dither_seed for this process's ops: "on" rounds the int8 operands stochastically
    (activations per encode call, weights per prepare / refresh) so that the fused encoder's miss bound holds for EVERY input
    (include/msae.h); "off" is round-to-nearest with the statistical noise model; "default" = environment MSAE_DITHER, else on.
    seed != 0 pins the hash seed of every call (reproducible candidate sets); 0 lets the library draw one per call."""
    if mode not in Options.DITHER:
        raise ValueError(f"dither mode {mode!r}: expected one of {sorted(Options.DITHER)}")
    _defaults.dither, _defaults.dither_seed = mode, int(seed)


def set_certified(on: bool) -> None:
    """Default of msae_options::certified for this process's ops: every fused encode runs the certified candidate pass (two
    int8 planes per operand, deterministic band: include/msae.h) -- its operands are built on first use per (weight, bias)
    version and cached (`prepare_encoder_certified`)."""
    _defaults.certified = bool(on)


_CERT_CACHE: "collections.OrderedDict" = collections.OrderedDict()
_CERT_NOTED: set = set()          # shapes whose "no certified pass -> exact path" warning has been given


def prepare_encoder_certified(W_enc: Tensor, b_enc: Optional[Tensor]) -> Optional[Tensor]:
    """msae_encoder_prepare_certified: the certified pass's operand buffer for (W_enc, b_enc), or None when the shape has no
    certified pass (the library then runs the exact path).  Cached per (pointer, version) of both tensors, two most recent
    (weights edited through `.data` do not bump the version: call release_workspaces() after such an edit)."""
    dev = _hip.require_device(W_enc, b_enc)
    lib = _hip.load()
    key = (dev, W_enc.data_ptr(), W_enc._version, tuple(W_enc.shape), 0 if b_enc is None else b_enc.data_ptr(),
           0 if b_enc is None else b_enc._version)
    hit = _CERT_CACHE.get(key)
    # (the entry remembers WHICH tensor objects it was built from, weakly: the key is addresses and version counters, and the
    # caching allocator hands a freed parameter's address to the next model of the same shape)
    if hit is not None and hit[1]() is W_enc and (b_enc is None or hit[2]() is b_enc):
        _CERT_CACHE.move_to_end(key)
        return hit[0]
    W, b = _f32c(W_enc), _f32c(b_enc)
    N, d = W.shape
    nbytes = lib.msae_encoder_certified_bytes(N, d)
    if nbytes == 0:
        return None
    buf = torch.empty(nbytes, dtype=torch.uint8, device=dev)
    with torch.cuda.device(dev):
        _hip.check(lib.msae_encoder_prepare_certified(_hip.ptr(W), _hip.ptr(b), N, d, _hip.ptr(buf), _hip.stream_of(W)),
                   "msae_encoder_prepare_certified")
    _CERT_CACHE[key] = (buf, weakref.ref(W_enc), weakref.ref(b_enc) if b_enc is not None else None)
    while len(_CERT_CACHE) > 2:
        _CERT_CACHE.popitem(last=False)
    return buf


def invalidate_certified(W_enc: Optional[Tensor] = None) -> None:
    """Forget the certified operand buffers built from `W_enc` (all of them when None).  The cache trusts the tensors' version
    counters, which an edit through `.data` does not bump: `Sae.invalidate_prepared` calls this, so the documented
    remedy after such an edit covers `encode(certified=True)` as well -- a stale certified buffer would rank features by
    the OLD weights and a token could verify with a wrong top-k."""
    if W_enc is None:
        _CERT_CACHE.clear()
        return
    for key in [k for k in _CERT_CACHE if k[1] == W_enc.data_ptr() or _CERT_CACHE[k][1]() is W_enc]:
        _CERT_CACHE.pop(key, None)


def coarse_in_force() -> str:
    """The candidate pass's operand type an encode issued NOW would resolve to: the process default, else the environment's
    MSAE_COARSE, else int8 (the library's rule, csrc/encode_defs.h resolve_opts)."""
    if _defaults.coarse != "default":
        return _defaults.coarse
    e = os.environ.get("MSAE_COARSE", "")[:1]
    return {"b": "bf16", "f": "fp8"}.get(e, "int8")


def dither_in_force() -> bool:
    """Whether an encode / prepare issued NOW rounds the int8 operands stochastically (process default, else MSAE_DITHER)."""
    if _defaults.dither != "default":
        return _defaults.dither == "on"
    return os.environ.get("MSAE_DITHER", "1")[:1] != "0"


def set_status_detail(on: bool) -> None:
    """Diagnostics: tokens recomputed inside the call report 1 | reason << 8 instead of 1."""
    _defaults.status_detail = bool(on)


def prepare_encoder(W_enc: Tensor, out: Optional[Tensor] = None, active_mode_only: bool = False,
                    tokens_next: int = 0) -> Tensor:
    """Coarse-pass operands of the encoder weights for the fused path (bf16 copy, int8 quantisation,
    sampled rows).  Once per weight load; `out` + `active_mode_only` is the per-step refresh of a
    training loop (rebuilds only what the coarse mode in force reads, into the same buffer; `tokens_next` > 0: only what
    ONE following encode of that many tokens reads -- msae_encoder_refresh_for)."""
    dev = _hip.require_device(W_enc)
    lib = _hip.load()
    W = _f32c(W_enc)
    N, d = W.shape
    nbytes = lib.msae_encoder_prepared_bytes(N, d)
    if out is None or out.numel() != nbytes or out.device != dev:
        out, active_mode_only = torch.empty(nbytes, dtype=torch.uint8, device=dev), False
    with torch.cuda.device(dev):
        if active_mode_only and tokens_next > 0:
            _hip.check(lib.msae_encoder_refresh_for(_hip.ptr(W), N, d, _hip.ptr(out), int(tokens_next), _opts().ref(),
                                                    _hip.stream_of(W)), "msae_encoder_refresh_for")
        elif active_mode_only:
            _hip.check(lib.msae_encoder_refresh(_hip.ptr(W), N, d, _hip.ptr(out), _opts().ref(), _hip.stream_of(W)),
                       "msae_encoder_refresh")
        else:
            _hip.check(lib.msae_encoder_prepare_opts(_hip.ptr(W), N, d, _hip.ptr(out), _opts().ref(), _hip.stream_of(W)),
                       "msae_encoder_prepare_opts")
    return out


_TRAIN_PREPARED: dict = {}
# key -> (weight version, coarse mode, covers batches of <= 256 tokens, weak reference to the weight tensor).  The reference
# pins the entry to ONE tensor object: the key is an address, and the caching allocator hands a freed parameter's address to
# the next model of the same shape -- whose version counter can stand at the same value (the certified cache met exactly that).
_TRAIN_FRESH: dict = {}


def _train_key(W_enc: Tensor):
    return (W_enc.device, W_enc.data_ptr(), tuple(W_enc.shape))


def train_operand_buffer(W_enc: Tensor) -> Tensor:
    """The per-parameter operand buffer of the training loop (allocated on first use): what `adam_rows_(refresh=...)`
    rebuilds inside the optimiser pass and `_refresh_train_operands` hands to the next encode."""
    key = _train_key(W_enc)
    buf = _TRAIN_PREPARED.get(key)
    if buf is None:
        nbytes = _hip.load().msae_encoder_prepared_bytes(W_enc.shape[0], W_enc.shape[1])
        if len(_TRAIN_PREPARED) > 8:
            _TRAIN_PREPARED.clear(); _TRAIN_FRESH.clear()
        buf = _TRAIN_PREPARED[key] = torch.empty(nbytes, dtype=torch.uint8, device=W_enc.device)
    return buf


def mark_train_operands_fresh(W_enc: Tensor, tokens_next: int) -> None:
    """The optimiser pass has just rebuilt train_operand_buffer(W_enc) from the updated weight (its version as of now)."""
    _TRAIN_FRESH[_train_key(W_enc)] = (W_enc._version, coarse_in_force(), tokens_next <= 256, weakref.ref(W_enc))


def invalidate_train_operands(W_enc: Optional[Tensor] = None) -> None:
    """Forget that the training loop's operand buffer of `W_enc` (all buffers when None) is fresh: the next training encode
    rebuilds it.  `_refresh_train_operands` trusts the tensor's version counter, which `p.data.copy_()` / `.data.mul_()`, a
    custom C op or an external optimiser writing through `.data` do NOT bump: call this wherever the weight is
    edited that way (`Sae.invalidate_prepared`, which `Sae.load_from_disk` ends with, does).  MSAE_DEBUG_OPERANDS=1 rebuilds
    before every training encode regardless."""
    if W_enc is None:
        _TRAIN_FRESH.clear()
    else:
        _TRAIN_FRESH.pop(_train_key(W_enc), None)


def _refresh_train_operands(W_enc: Tensor, tokens: int) -> Tensor:
    """Per-step operands of a weight that changes every step: one buffer per parameter, rebuilt in
    place for the coarse mode in force and for the batch size of the ONE encode that follows (it runs in that mode on
    `tokens` tokens; the buffer is rebuilt before it is read again) -- unless the optimiser pass that produced this
    version of the weight has already rebuilt it (adam_rows_(refresh=...): no second sweep over W_enc)."""
    key = _train_key(W_enc)
    fresh = _TRAIN_FRESH.get(key) if os.environ.get("MSAE_DEBUG_OPERANDS", "0") in ("", "0") else None
    if fresh is not None and fresh[:2] == (W_enc._version, coarse_in_force()) and fresh[3]() is W_enc \
            and (tokens > 256 or fresh[2]) and key in _TRAIN_PREPARED:
        return _TRAIN_PREPARED[key]
    buf = prepare_encoder(W_enc, _TRAIN_PREPARED.get(key), active_mode_only=True, tokens_next=tokens)
    if len(_TRAIN_PREPARED) > 8 and key not in _TRAIN_PREPARED:
        _TRAIN_PREPARED.clear(); _TRAIN_FRESH.clear()
    _TRAIN_PREPARED[key] = buf
    _TRAIN_FRESH[key] = (W_enc._version, coarse_in_force(), tokens <= 256, weakref.ref(W_enc))
    return buf


@torch.library.custom_op("msae::encode_topk", mutates_args=())
def encode_topk(x: Tensor, W_enc: Tensor, b_enc: Optional[Tensor], b_dec: Optional[Tensor],
                prepared: Optional[Tensor], k: int, set_feature: int = -1, set_value: float = 0.0,
                zero_feature: int = -1, coarse_mode: int = -1, guard_z: float = 0.0,
                status_detail: bool = False, exact: bool = False, dither: int = 0,
                dither_seed: int = 0, certified: bool = False) -> Tuple[Tensor, Tensor, Tensor]:
    """Fused Sae.encode -> (top_acts f32 [...,k], top_indices int64 [...,k], status int32 [...]).
    coarse_mode (-1 default / 0 bf16 / 1 int8), guard_z (0 = default), status_detail and exact (every token by the
    exact path: include/msae.h, msae_options::exact), dither (0 default / 1 on / 2 off) and dither_seed (0 = drawn by the
    library) and certified (the two-plane pass with the deterministic band; its operand buffer comes from
    prepare_encoder_certified's cache) are this call's msae_options; what is left at its default comes from the process
    defaults (set_coarse_mode & co.)."""
    dev = _hip.require_device(x, W_enc, b_enc, b_dec, prepared)
    lib = _hip.load()
    xa, W, be, bd = _act(x), _f32c(W_enc), _f32c(b_enc), _f32c(b_dec)
    N, d = W.shape
    assert xa.shape[-1] == d, f"x last dim {xa.shape[-1]} != d_in {d}"
    T = xa.numel() // d
    # the kernels write every element of all three outputs (int64 indices directly: no widening pass)
    vals = torch.empty(*xa.shape[:-1], k, dtype=torch.float32, device=dev)
    idx = torch.empty(*xa.shape[:-1], k, dtype=torch.int64, device=dev)
    status = torch.empty(xa.shape[:-1], dtype=torch.int32, device=dev)
    if T == 0:
        return vals, idx, status
    cert_ops = None
    if (certified or _defaults.certified) and not (exact or _defaults.exact):
        cert_ops = prepare_encoder_certified(W_enc, b_enc)
        if cert_ops is None:
            exact = True                 # a shape without the certified pass: the exact path is the certified answer
            if (N, d) not in _CERT_NOTED:                           # ... and ~20x the time on large batches: say so, once per shape
                _CERT_NOTED.add((N, d))
                import warnings

                warnings.warn(f"certified encode: no certified pass for num_latents = {N}, d_in = {d} (it needs num_latents % 8192 "
                              "== 0 and d_in % 128 == 0, d_in <= 65536); these encodes run the exact f32 path -- the same "
                              "answer, ~20x slower on large batches (include/msae.h, msae_options::certified)", stacklevel=3)
    opts = _opts(coarse_mode, guard_z, status_detail, exact, dither, dither_seed, cert_ops)
    rows = _defaults.rows_rescored
    if rows is not None and (T > rows.numel() or rows.device != dev):
        # rescore_rows(buf) serves encodes of <= buf.numel() tokens on buf's device; a larger call inside the block runs
        # WITHOUT the statistics instead of writing past the buffer
        with _without_rows():
            opts = _opts(coarse_mode, guard_z, status_detail, exact, dither, dither_seed, cert_ops)
    ws = _workspace(dev, _encode_ws_bytes(lib, T, d, N, k, opts))
    with torch.cuda.device(dev):
        _hip.check(lib.msae_encode_topk_i64(_hip.ptr(xa), _hip.DTYPE_CODE[xa.dtype], _hip.ptr(W),
                                            _hip.ptr(be), _hip.ptr(bd), _hip.ptr(prepared), T, d, N, k,
                                            set_feature, set_value, zero_feature, _hip.ptr(vals),
                                            _hip.ptr(idx), _hip.ptr(status), _hip.ptr(ws), ws.numel(),
                                            opts.ref(), _hip.stream_of(xa)), "msae_encode_topk_i64")
    return vals, idx, status


@encode_topk.register_fake
def _(x, W_enc, b_enc, b_dec, prepared, k, set_feature=-1, set_value=0.0, zero_feature=-1, coarse_mode=-1,
      guard_z=0.0, status_detail=False, exact=False, dither=0, dither_seed=0, certified=False):
    return (x.new_empty(*x.shape[:-1], k, dtype=torch.float32),
            x.new_empty(*x.shape[:-1], k, dtype=torch.int64),
            x.new_empty(x.shape[:-1], dtype=torch.int32))


def shard_candidates(x: Tensor, b_enc_shard: Optional[Tensor], b_dec: Optional[Tensor], prepared_shard: Tensor,
                     N_shard: int, k: int, row_offset: int, C: int, set_feature: int = -1,
                     zero_feature: int = -1) -> Tensor:
    """Feature-sharded group, sender side (msae_shard_candidates): -> records uint8 [T, record_bytes(C)]."""
    dev = _hip.require_device(x, b_enc_shard, b_dec, prepared_shard)
    lib = _hip.load()
    xa, be, bd = _act(x), _f32c(b_enc_shard), _f32c(b_dec)
    d = xa.shape[-1]
    T = xa.numel() // d
    stride = lib.msae_shard_record_bytes(C)
    recs = torch.empty(T, stride, dtype=torch.uint8, device=dev)
    if T == 0:
        return recs
    opts = _opts_for(T, dev)
    ws = _workspace(dev, _encode_ws_bytes(lib, T, d, N_shard, k, opts))
    with torch.cuda.device(dev):
        _hip.check(lib.msae_shard_candidates(_hip.ptr(xa), _hip.DTYPE_CODE[xa.dtype], _hip.ptr(be), _hip.ptr(bd),
                                             _hip.ptr(prepared_shard), T, d, N_shard, k, row_offset, C, set_feature,
                                             zero_feature, _hip.ptr(recs), _hip.ptr(ws), ws.numel(), opts.ref(),
                                             _hip.stream_of(xa)), "msae_shard_candidates")
    return recs


def rescore_candidates(x: Tensor, W_enc: Tensor, b_enc: Optional[Tensor], b_dec: Optional[Tensor], k: int,
                       records: Tensor, C: int, T_valid: Optional[int] = None, set_feature: int = -1,
                       set_value: float = 0.0, zero_feature: int = -1) -> Tuple[Tensor, Tensor, Tensor]:
    """Feature-sharded group, owner side (msae_rescore_candidates): records uint8 [G, T, record_bytes(C)] of
    this rank's T tokens -> (top_acts f32 [T, k], top_indices int64 [T, k] global ids, status int32 [T])."""
    dev = _hip.require_device(x, W_enc, b_enc, b_dec, records)
    lib = _hip.load()
    xa, W, be, bd = _act(x), _f32c(W_enc), _f32c(b_enc), _f32c(b_dec)
    N, d = W.shape
    T = xa.numel() // d
    G = records.shape[0]
    assert records.dtype == torch.uint8 and records.is_contiguous() and records.shape[1] == T, records.shape
    assert records.shape[2] == lib.msae_shard_record_bytes(C)
    T_valid = T if T_valid is None else T_valid
    vals = torch.zeros(T, k, dtype=torch.float32, device=dev) if T_valid < T else torch.empty(T, k, dtype=torch.float32, device=dev)
    idx = torch.zeros(T, k, dtype=torch.int64, device=dev) if T_valid < T else torch.empty(T, k, dtype=torch.int64, device=dev)
    status = torch.zeros(T, dtype=torch.int32, device=dev) if T_valid < T else torch.empty(T, dtype=torch.int32, device=dev)
    if T_valid == 0:
        return vals, idx, status
    ws = _workspace(dev, lib.msae_rescore_candidates_ws_bytes(T, d, N, k, G, C))
    with torch.cuda.device(dev):
        _hip.check(lib.msae_rescore_candidates(_hip.ptr(xa), _hip.DTYPE_CODE[xa.dtype], _hip.ptr(W), _hip.ptr(be),
                                               _hip.ptr(bd), T, T_valid, d, N, k, G, C, _hip.ptr(records), set_feature,
                                               set_value, zero_feature, _hip.ptr(vals), _hip.ptr(idx),
                                               _hip.ptr(status), _hip.ptr(ws), ws.numel(), _opts_for(T, dev).ref(),
                                               _hip.stream_of(xa)),
                   "msae_rescore_candidates")
    return vals, idx, status


# ---- decoder (differentiable, mirrors TritonDecoder: kernels.py:403-429) ---------------------------
def _idx32(top_indices: Tensor) -> Tensor:
    return top_indices.detach().to(torch.int32).contiguous()


@torch.library.custom_op("msae::decode", mutates_args=())
def decode(top_indices: Tensor, top_acts: Tensor, W_dec: Tensor, b_dec: Optional[Tensor]) -> Tensor:
    """sum_j top_acts[:, j] * W_dec[top_indices[:, j]] (+ b_dec);  W_dec is [N, d] row-major."""
    dev = _hip.require_device(top_indices, top_acts, W_dec, b_dec)
    lib = _hip.load()
    assert top_indices.shape == top_acts.shape, "indices / acts shape mismatch"  # kernels.py:193
    acts, W, bd = _f32c(top_acts), _f32c(W_dec), _f32c(b_dec)
    if top_indices.dtype == torch.int64:      # Tensor.topk's index type: read as it is, no narrowing copy
        idx, fn, name = top_indices.detach().contiguous(), lib.msae_decode_i64_f32, "msae_decode_i64_f32"
    else:
        idx, fn, name = _idx32(top_indices), lib.msae_decode_f32, "msae_decode_f32"
    N, d = W.shape
    k = idx.shape[-1]
    A = idx.numel() // k
    out = torch.empty(*idx.shape[:-1], d, dtype=torch.float32, device=dev)
    flag = _bounds_flag(dev)
    with torch.cuda.device(dev):
        _hip.check(fn(_hip.ptr(idx), _hip.ptr(acts), _hip.ptr(W), _hip.ptr(bd), A, k, N, d, _hip.ptr(out),
                      _hip.ptr(flag), _hip.stream_of(acts)), "%s" % name)
    _check_bounds(flag, name)
    return out


@decode.register_fake
def _(top_indices, top_acts, W_dec, b_dec):
    return top_acts.new_empty(*top_acts.shape[:-1], W_dec.shape[1], dtype=torch.float32)


_WGRAD_COLLECT: Optional[dict] = None


@contextlib.contextmanager
def collect_wgrad_sumsq():
    """Weight-gradient kernels launched inside the block also write the squared norm of every gradient row (while it is in
    registers) and record it: {weight data_ptr: [calls, gradient data_ptr, row_sumsq f32 [N]]}.  The trainer's
    clip_grad_norm_ total (train/sae/sae/trainer.py:390) is then a 512-KB sum instead of another read of the 2 GiB
    gradient -- valid for a parameter whose .grad IS the one recorded tensor (exactly one call, no accumulation)."""
    global _WGRAD_COLLECT
    prev, _WGRAD_COLLECT = _WGRAD_COLLECT, {}
    try:
        yield _WGRAD_COLLECT
    finally:
        _WGRAD_COLLECT = prev


_WGRAD_ROWSUM: Optional[list] = None      # set by _SparseEncode.backward around its weight-gradient call


@torch.library.custom_op("msae::decode_bwd", mutates_args=())
def decode_bwd(top_indices: Tensor, top_acts: Tensor, W_dec: Tensor, grad_out: Tensor,
               need_acts: bool, need_w: bool) -> Tuple[Tensor, Tensor]:
    dev = _hip.require_device(top_indices, top_acts, W_dec, grad_out)
    lib = _hip.load()
    idx, acts, W, g = _idx32(top_indices), _f32c(top_acts), _f32c(W_dec), _f32c(grad_out)
    N, d = W.shape
    k = idx.shape[-1]
    A = idx.numel() // k
    g_acts = torch.empty_like(acts) if need_acts else acts.new_empty(0)
    g_w = torch.empty_like(W) if need_w else W.new_empty(0)   # the kernel writes every row
    flag = _bounds_flag(dev)
    with torch.cuda.device(dev):
        st = _hip.stream_of(g)
        if need_acts:
            _hip.check(lib.msae_decode_bwd_acts_f32(_hip.ptr(idx), _hip.ptr(g), _hip.ptr(W), A, k, N, d,
                                                    _hip.ptr(g_acts), _hip.ptr(flag), st), "msae_decode_bwd_acts_f32")
        if need_w:
            ws = _workspace(dev, lib.msae_decode_bwd_wdec_ws_bytes(A, k, N))
            rowsq = torch.empty(N, dtype=torch.float32, device=dev) if _WGRAD_COLLECT is not None else None
            rowsum = torch.empty(N, dtype=torch.float32, device=dev) if _WGRAD_ROWSUM is not None else None
            _hip.check(lib.msae_decode_bwd_wdec_f32(_hip.ptr(idx), _hip.ptr(acts), _hip.ptr(g), A, k, N,
                                                    d, _hip.ptr(g_w), _hip.ptr(rowsq), _hip.ptr(rowsum), _hip.ptr(flag),
                                                    _hip.ptr(ws), ws.numel(), st),
                       "msae_decode_bwd_wdec_f32")
            if rowsum is not None:
                _WGRAD_ROWSUM.append(rowsum)
            if rowsq is not None:
                ent = _WGRAD_COLLECT.setdefault(W_dec.data_ptr(), [0, 0, None])
                ent[0] += 1
                ent[1], ent[2] = g_w.data_ptr(), rowsq
    _check_bounds(flag, "msae_decode_bwd")
    return g_acts, g_w


@decode_bwd.register_fake
def _(top_indices, top_acts, W_dec, grad_out, need_acts, need_w):
    return (torch.empty_like(top_acts, dtype=torch.float32) if need_acts else top_acts.new_empty(0),
            torch.empty_like(W_dec, dtype=torch.float32) if need_w else W_dec.new_empty(0))


def _decode_setup(ctx, inputs, output):
    top_indices, top_acts, W_dec, b_dec = inputs
    ctx.save_for_backward(top_indices, top_acts, W_dec)
    ctx.has_bdec = b_dec is not None


def _decode_backward(ctx, grad_out):
    top_indices, top_acts, W_dec = ctx.saved_tensors
    need_acts, need_w = ctx.needs_input_grad[1], ctx.needs_input_grad[2]
    g_acts = g_w = g_b = None
    if need_acts or need_w:
        ga, gw = decode_bwd(top_indices, top_acts, W_dec, grad_out.contiguous(), need_acts, need_w)
        g_acts = ga.to(top_acts.dtype) if need_acts else None
        g_w = gw.to(W_dec.dtype) if need_w else None
    if ctx.has_bdec and ctx.needs_input_grad[3]:
        g_b = grad_out.reshape(-1, grad_out.shape[-1]).sum(0)
    return None, g_acts, g_w, g_b


decode.register_autograd(_decode_backward, setup_context=_decode_setup)


# ---- cache sparsify ---------------------------------------------------------------------------------
def sparsify(top_acts: Tensor, top_indices: Tensor, num_latents: int, row_base: int = 0,
             thresh: float = 1e-5, filter_bitmap: Optional[Tensor] = None, sync: bool = True):
    """[B, S, k] top-k -> (locations [nnz, 3] int64, activations [nnz] f32) in the reference cache's
    record order (row, pos, feature ascending); |v| > thresh and optional feature bitmap applied.

    sync=True reads nnz back (one int64, as `torch.nonzero` does) and returns exactly-sized tensors.
    sync=False never touches the host: returns (locations [B*S*k, 3], activations [B*S*k], nnz) with the
    first nnz rows valid and nnz a device int64 scalar -- for loops inside a forward hook that must stay
    stream-ordered (msae/features/cache.py collects these and reads the counts back once per flush)."""
    dev = _hip.require_device(top_acts, top_indices, filter_bitmap)
    lib = _hip.load()
    assert top_acts.dim() == 3 and top_acts.shape == top_indices.shape
    B, S, k = top_acts.shape
    vals, idx = _f32c(top_acts), _idx32(top_indices)
    counts = torch.empty(B * S + 1, dtype=torch.int64, device=dev)
    fb = None if filter_bitmap is None else filter_bitmap.to(torch.uint8).contiguous()
    with torch.cuda.device(dev):
        st = _hip.stream_of(vals)
        _hip.check(lib.msae_sparsify_count(_hip.ptr(vals), _hip.ptr(idx), B, S, k, thresh, _hip.ptr(fb),
                                           num_latents, _hip.ptr(counts), st), "msae_sparsify_count")
        nnz = int(counts[-1].item()) if sync else B * S * k   # the one host read (torch.nonzero does the same)
        loc = torch.empty(nnz, 3, dtype=torch.int64, device=dev)
        act = torch.empty(nnz, dtype=torch.float32, device=dev)
        if nnz:
            _hip.check(lib.msae_sparsify_write(_hip.ptr(vals), _hip.ptr(idx), B, S, k, thresh,
                                               _hip.ptr(fb), num_latents, row_base, _hip.ptr(counts),
                                               _hip.ptr(loc), _hip.ptr(act), st), "msae_sparsify_write")
    return (loc, act) if sync else (loc, act, counts[-1])


def merge_topk_gathered(gathered: Tensor, T: int, G: int, kl: int, k: int):
    """Canonical top-k of the all-gathered per-shard pairs (int32 [G*2, T, kl]) on the device.
    -> (vals f32 [T,k], idx int64 [T,k], flagged int32 [T])."""
    dev = _hip.require_device(gathered)
    lib = _hip.load()
    assert gathered.dtype == torch.int32 and gathered.is_contiguous() and gathered.numel() == G * 2 * T * kl
    vals = torch.empty(T, k, dtype=torch.float32, device=dev)
    idx = torch.empty(T, k, dtype=torch.int32, device=dev)
    flagged = torch.empty(T, dtype=torch.int32, device=dev)
    with torch.cuda.device(dev):
        _hip.check(lib.msae_merge_topk(_hip.ptr(gathered), T, G, kl, k, _hip.ptr(vals), _hip.ptr(idx),
                                       _hip.ptr(flagged), _hip.stream_of(gathered)), "msae_merge_topk")
    return vals, idx.to(torch.int64), flagged


def merge_topk_gathered_masked_(gathered: Tensor, T: int, G: int, kl: int, k: int, mask: Tensor, vals: Tensor,
                                idx: Tensor) -> None:
    """msae_merge_topk_masked: rows t of (vals f32 [T,k], idx int64 [T,k]) with mask[t] != 0 (int32 [T]) become the
    canonical top-k of the gathered pairs, in place; the other rows keep what they hold."""
    dev = _hip.require_device(gathered, mask, vals, idx)
    assert gathered.dtype == torch.int32 and gathered.is_contiguous() and gathered.numel() == G * 2 * T * kl
    assert mask.dtype == torch.int32 and mask.numel() == T and mask.is_contiguous()
    assert vals.dtype == torch.float32 and idx.dtype == torch.int64 and vals.is_contiguous() and idx.is_contiguous()
    assert vals.shape == (T, k) and idx.shape == (T, k)
    with torch.cuda.device(dev):
        _hip.check(_hip.load().msae_merge_topk_masked(_hip.ptr(gathered), T, G, kl, k, _hip.ptr(mask), _hip.ptr(vals),
                                                      None, _hip.ptr(idx), _hip.stream_of(gathered)),
                   "msae_merge_topk_masked")


def compact_flags(flags: Tensor) -> Tuple[Tensor, Tensor]:
    """flags int32 [T] -> (rows int32 [T]: the flagged t in ascending order in rows[:n], n int32 [1]), all on the
    device: the redo list `encode_topk_rows_` takes.  Nothing is read back."""
    dev = _hip.require_device(flags)
    assert flags.dtype == torch.int32 and flags.is_contiguous()
    T = flags.numel()
    rows = torch.empty(max(T, 1), dtype=torch.int32, device=dev)
    n = torch.empty(1, dtype=torch.int32, device=dev)
    with torch.cuda.device(dev):
        _hip.check(_hip.load().msae_compact_flags(_hip.ptr(flags), T, _hip.ptr(rows), _hip.ptr(n),
                                                  _hip.stream_of(flags)), "msae_compact_flags")
    return rows, n


def encode_topk_rows_(x: Tensor, W_enc: Tensor, b_enc: Optional[Tensor], b_dec: Optional[Tensor], rows: Tensor,
                      n_rows: Tensor, k: int, vals: Tensor, idx: Tensor, status: Optional[Tensor] = None,
                      set_feature: int = -1, set_value: float = 0.0, zero_feature: int = -1) -> None:
    """msae_encode_topk_rows: the exact Sae.encode of the tokens rows[:n_rows] (device-side list and count) of
    x [T, d], written to rows of vals f32 [T, k] / idx int64 [T, k] (/ status int32 [T] = 1) in place.  The work is
    sized on the device; nothing is read back."""
    dev = _hip.require_device(x, W_enc, b_enc, b_dec, rows, n_rows, vals, idx, status)
    lib = _hip.load()
    xa, W, be, bd = _act(x), _f32c(W_enc), _f32c(b_enc), _f32c(b_dec)
    N, d = W.shape
    T = xa.numel() // d
    assert rows.dtype == torch.int32 and n_rows.dtype == torch.int32 and rows.numel() >= T
    assert vals.dtype == torch.float32 and idx.dtype == torch.int64 and vals.is_contiguous() and idx.is_contiguous()
    assert vals.shape == (T, k) and idx.shape == (T, k)
    if T == 0:
        return
    ws = _workspace(dev, lib.msae_encode_topk_rows_ws_bytes(T, N))
    with torch.cuda.device(dev):
        _hip.check(lib.msae_encode_topk_rows(_hip.ptr(xa), _hip.DTYPE_CODE[xa.dtype], _hip.ptr(W), _hip.ptr(be),
                                             _hip.ptr(bd), _hip.ptr(rows), _hip.ptr(n_rows), T, d, N, k, set_feature,
                                             float(set_value), zero_feature, _hip.ptr(vals), _hip.ptr(idx),
                                             _hip.ptr(status), _hip.ptr(ws), ws.numel(), _hip.stream_of(xa)),
                   "msae_encode_topk_rows")


# ---- trainable encoder (training forward, sae.py:193-247) -------------------------------------------
class _SparseEncode(torch.autograd.Function):
    """pre_acts + the three TopK selections of Sae.forward as ONE autograd node with a SPARSE
    backward.  The reference back-propagates a dense [T, N] gradient through topk -> relu ->
    nn.Linear (a second full GEMM, dW = g^T a); only the selected (token, latent) pairs carry
    gradient, so here
        dW_enc[n] += g[t,j] * a[t]      -> msae_decode_bwd_wdec_f32 (sparse outer-product accumulate)
        da[t]      = sum_j g[t,j] W_enc[n_tj]  -> msae_decode_f32 over W_enc (gather matmul)
        db_enc[n] += g[t,j]  ;  dx = da ;  db_dec = -sum_t da[t]
    with g masked by relu'(pre) = (value > 0)."""

    @staticmethod
    def forward(ctx, x, W_enc, b_enc, b_dec, k, dead_mask, k_aux, k_multi, prepared=None, set_feature=-1,
                set_value=0.0, zero_feature=-1):
        vals, idxs = [], []
        edits = set_feature >= 0 or zero_feature >= 0
        assert not (edits and (k_aux > 0 or k_multi > 0)), "hook edits apply to the plain top-k only"
        if k_aux == 0 and max(k, k_multi) <= 256:
            # no AuxK term: the fused encoder gives the canonical top-max(k, 4k); the top-k is its
            # prefix (same order), and the dense [T, N] latents are never built.  `prepared`: operands
            # of a weight that does not change between calls (inference hooks); None = training step
            kk = max(k, k_multi)
            v, i, _ = encode_topk(x, W_enc, b_enc, b_dec,
                                  prepared if prepared is not None else _refresh_train_operands(W_enc, x.shape[0]), kk,
                                  set_feature, set_value, zero_feature)
            vals.append(v[..., :k].contiguous()); idxs.append(i[..., :k].contiguous())
            if k_multi > 0:
                vals.append(v); idxs.append(i)
        else:
            pre = pre_acts(x, W_enc, b_enc, b_dec)
            if set_feature >= 0:
                pre[..., set_feature] = set_value
            if zero_feature >= 0:
                pre[..., zero_feature] = 0.0
            v, i = topk(pre, k)
            vals.append(v); idxs.append(i)
            if k_aux > 0:
                dead_pre = torch.where(dead_mask[None], pre, -torch.inf)            # sae.py:217-220
                v, i = topk(dead_pre, k_aux)        # LDS-resident selection: k_aux <= 16384 (d_in <= 32768)
                vals.append(v); idxs.append(i)
            if k_multi > 0:
                v, i = topk(pre, k_multi)                                           # sae.py:233
                vals.append(v); idxs.append(i)
        ctx.save_for_backward(x, W_enc, b_dec, torch.cat(idxs, -1), torch.cat(vals, -1))
        ctx.splits = [t.shape[-1] for t in vals]
        ctx.set_feature = set_feature
        ctx.has_b_enc = b_enc is not None
        out = []
        for v, i in zip(vals, idxs):
            out += [v, i]
        for i in idxs:
            ctx.mark_non_differentiable(i)
        return tuple(out)

    @staticmethod
    def backward(ctx, *grads):
        x, W_enc, b_dec, idx_cat, val_cat = ctx.saved_tensors
        g = [grads[2 * j] if grads[2 * j] is not None else
             torch.zeros(val_cat.shape[0], n, device=val_cat.device) for j, n in enumerate(ctx.splits)]
        g_cat = torch.cat(g, -1).float() * (val_cat > 0)           # relu'
        if ctx.set_feature >= 0:                                   # a latent overwritten by a constant carries no gradient
            g_cat = g_cat * (idx_cat != ctx.set_feature)
        a = x.float() - b_dec
        need_x, need_W, need_be, need_bd = ctx.needs_input_grad[:4]
        g_x = g_W = g_be = g_bd = None
        rowsum = None
        if need_W:
            # the weight-gradient kernel also sums every feature's latent gradients (ascending pair order): the encoder-bias
            # gradient without index_add_'s atomics, and the vector s of the b_dec gradient -(s^T W_enc) below
            global _WGRAD_ROWSUM
            prev, _WGRAD_ROWSUM = _WGRAD_ROWSUM, []
            try:
                _, g_W = decode_bwd(idx_cat, g_cat, W_enc, a.contiguous(), False, True)
                rowsum = _WGRAD_ROWSUM[0] if _WGRAD_ROWSUM else None
            finally:
                _WGRAD_ROWSUM = prev
        if need_be and ctx.has_b_enc:
            g_be = rowsum if rowsum is not None else torch.zeros(W_enc.shape[0], device=g_cat.device).index_add_(
                0, idx_cat.reshape(-1), g_cat.reshape(-1))
        if need_bd and not need_x and rowsum is not None and W_enc.shape[1] % 4 == 0:
            g_bd = weighted_row_sum(W_enc, rowsum, -1.0)          # ONE streaming read of W_enc (training: x is a constant)
        elif need_x or need_bd:
            da = decode(idx_cat, g_cat, W_enc, None)
            g_x = da.to(x.dtype) if need_x else None
            g_bd = -da.sum(0) if need_bd else None
        return g_x, g_W, g_be, g_bd, None, None, None, None, None, None, None, None


def sparse_encode(x: Tensor, W_enc: Tensor, b_enc: Tensor, b_dec: Tensor, k: int,
                  dead_mask: Optional[Tensor] = None, k_aux: int = 0, k_multi: int = 0, *,
                  prepared: Optional[Tensor] = None, set_feature: int = -1, set_value: float = 0.0,
                  zero_feature: int = -1):
    """-> [(acts, idx)] for the top-k, (optional) AuxK and (optional) Multi-TopK selections.
    Differentiable w.r.t. x, W_enc, b_enc, b_dec through the selected latents (the graph of the
    reference's pre_acts -> [mask] -> topk, sae.py:172-185, patching/utils.py:43-49)."""
    lead = x.shape[:-1]
    out = _SparseEncode.apply(x.reshape(-1, x.shape[-1]), W_enc, b_enc, b_dec, k, dead_mask, k_aux, k_multi, prepared,
                              set_feature, float(set_value), zero_feature)        # the node works on [T, d]
    if len(lead) != 1:
        out = tuple(o.reshape(*lead, o.shape[-1]) for o in out)
    return [(out[2 * j], out[2 * j + 1]) for j in range(len(out) // 2)]


# ---- parameter-sized passes of one optimisation step (csrc/train.hip) --------------------------------
def unit_norm_rows_(W: Tensor, eps: float) -> Tensor:
    """W /= ||W||_row + eps, in place, one read + one write (sae.py:249-255)."""
    dev = _hip.require_device(W)
    assert W.dtype == torch.float32 and W.dim() == 2 and W.is_contiguous()
    with torch.cuda.device(dev):
        _hip.check(_hip.load().msae_unit_norm_rows_f32(_hip.ptr(W), W.shape[0], W.shape[1], float(eps),
                                                       _hip.stream_of(W)), "msae_unit_norm_rows_f32")
    return W


def grad_sumsq_(accum: Tensor, g: Tensor) -> Tensor:
    """accum (f32 device scalar) += sum(g^2): the total-norm half of clip_grad_norm_, no host sync."""
    dev = _hip.require_device(g, accum)
    assert g.dtype == torch.float32 and g.is_contiguous() and accum.dtype == torch.float32 and accum.numel() == 1
    with torch.cuda.device(dev):
        _hip.check(_hip.load().msae_grad_sumsq_f32(_hip.ptr(g), g.numel(), _hip.ptr(accum), _hip.stream_of(g)),
                   "msae_grad_sumsq_f32")
    return accum


def weighted_row_sum(W: Tensor, s: Tensor, scale: float = 1.0) -> Tensor:
    """scale * sum_n s[n] W[n, :] -> [d] f32, summed in a fixed order; rows with s[n] == 0 are not read
    (msae_weighted_row_sum_f32)."""
    dev = _hip.require_device(W, s)
    lib = _hip.load()
    Wc, sc = _f32c(W), _f32c(s)
    N, d = Wc.shape
    assert sc.numel() == N
    out = torch.empty(d, dtype=torch.float32, device=dev)
    ws = _workspace(dev, lib.msae_weighted_row_sum_ws_bytes(N, d))
    with torch.cuda.device(dev):
        _hip.check(lib.msae_weighted_row_sum_f32(_hip.ptr(Wc), _hip.ptr(sc), N, d, float(scale), _hip.ptr(out), _hip.ptr(ws),
                                                 ws.numel(), _hip.stream_of(Wc)), "msae_weighted_row_sum_f32")
    return out


def sum_into_(accum: Tensor, v: Tensor) -> Tensor:
    """accum (f32 device scalar) += sum(v), summed in a fixed order (msae_sum_f32): the total of a weight gradient's
    per-row squared norms (collect_wgrad_sumsq)."""
    dev = _hip.require_device(accum, v)
    assert v.dtype == torch.float32 and v.is_contiguous() and accum.dtype == torch.float32 and accum.numel() == 1
    with torch.cuda.device(dev):
        _hip.check(_hip.load().msae_sum_f32(_hip.ptr(v), v.numel(), _hip.ptr(accum), _hip.stream_of(v)), "msae_sum_f32")
    return accum


def adam_rows_(p: Tensor, g: Tensor, m: Tensor, v: Tensor, step: int, lr: float, *,
               total_sumsq: Optional[Tensor] = None, max_norm: float = 1.0, project: bool = False,
               betas: Tuple[float, float] = (0.9, 0.999), eps: float = 1e-8, renorm_eps: Optional[float] = None,
               refresh: Optional[Tensor] = None, tokens_next: int = 0) -> None:
    """One fused pass: clip by the global gradient norm, optionally remove the component of each
    gradient row parallel to the parameter row (sae.py:257-271), Adam update of p, m, v in place.
    `renorm_eps` (the decoder): rows divided by their norm + eps after the update -- the NEXT step's
    set_decoder_norm_to_unit_norm (sae.py:249-255), same bits, no extra sweep.  `refresh` (the encoder weight; a
    train_operand_buffer): the coarse-pass operands of the updated rows for the next encode of `tokens_next` tokens,
    as prepare_encoder(..., active_mode_only=True, tokens_next=...) would rebuild them (msae_adam_rows_fused_f32)."""
    dev = _hip.require_device(p, g, m, v, refresh)
    for t in (p, g, m, v):
        assert t.dtype == torch.float32 and t.is_contiguous() and t.shape == p.shape
    if p.dim() == 2:
        rows, d = p.shape
    else:   # vectors: any row length works without the projection
        assert not project
        d = 1024 if p.numel() % 1024 == 0 else p.numel()
        rows = p.numel() // d
    with torch.cuda.device(dev):
        if renorm_eps is not None or refresh is not None:
            assert p.dim() == 2, "renorm / refresh are row operations on a [rows, d] matrix"
            _hip.check(_hip.load().msae_adam_rows_fused_f32(
                _hip.ptr(p), _hip.ptr(g), _hip.ptr(m), _hip.ptr(v), rows, d,
                _hip.ptr(total_sumsq) if total_sumsq is not None else None, float(max_norm), int(project),
                float(lr), float(betas[0]), float(betas[1]), float(eps), int(step),
                float(renorm_eps) if renorm_eps is not None else -1.0, _hip.ptr(refresh), int(tokens_next),
                _opts().ref(), _hip.stream_of(p)), "msae_adam_rows_fused_f32")
            return
        _hip.check(_hip.load().msae_adam_rows_f32(
            _hip.ptr(p), _hip.ptr(g), _hip.ptr(m), _hip.ptr(v), rows, d,
            _hip.ptr(total_sumsq) if total_sumsq is not None else None, float(max_norm), int(project),
            float(lr), float(betas[0]), float(betas[1]), float(eps), int(step), _hip.stream_of(p)),
            "msae_adam_rows_f32")
