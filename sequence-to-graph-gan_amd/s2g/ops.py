"""Tensor-level wrappers over the C ABI: each function checks its operands on the host
(device, dtype, contiguity, sizes the kernel's grid assumes) and enqueues one kernel on the
current HIP stream.  torch is used only for device memory and the stream handle."""
from __future__ import annotations

import ctypes
from typing import Optional

import torch

from . import lib
from .lib import S2GError, S2GGemm, call

H = 50
G4 = 200
MASKW = 52


def stream_ptr() -> int:
    """Raw handle of torch's current HIP stream (the C-level getter: torch.cuda.current_stream() costs
    several microseconds of Python per call, and every launch asks for it)."""
    return torch._C._cuda_getCurrentRawStream(torch._C._cuda_getDevice())


_STREAMS = {}


def cur_stream() -> torch.cuda.Stream:
    """torch's current stream as a Stream object, through the C-level getter and a table of the objects handed out
    so far: torch.cuda.current_stream() spends ~10 us in device-index lookups, and every Event.record() without a
    stream argument calls it (48 records per training step)."""
    key = torch._C._cuda_getCurrentStream(torch._C._cuda_getDevice())
    st = _STREAMS.get(key)
    if st is None:
        st = _STREAMS[key] = torch.cuda.Stream(stream_id=key[0], device_index=key[1], device_type=key[2])
    return st


def record(ev: torch.cuda.Event) -> None:
    """ev.record() on the current stream (see cur_stream)."""
    ev.record(cur_stream())


class on_stream:
    """`with on_stream(s):` makes torch stream `s` current, like torch.cuda.stream(s), through the C-level
    getter / setter only: the torch context manager costs ~10 us of Python per use (device-index lookups), and a
    critic iteration enters nine of them.  s = None: no change."""
    __slots__ = ("s", "prev")

    def __init__(self, s):
        self.s = s

    def __enter__(self):
        s = self.s
        if s is not None:
            self.prev = torch._C._cuda_getCurrentStream(s.device_index)
            torch._C._cuda_setStream(stream_id=s.stream_id, device_index=s.device_index, device_type=s.device_type)
        return s

    def __exit__(self, *exc):
        if self.s is not None:
            sid, dev, typ = self.prev
            torch._C._cuda_setStream(stream_id=sid, device_index=dev, device_type=typ)
        return False


def _chk(t: torch.Tensor, dtype=torch.float32, name="tensor"):
    if not isinstance(t, torch.Tensor) or not t.is_cuda:
        raise S2GError(f"{name}: expected a CUDA/HIP tensor (this implementation has no CPU path)")
    if t.dtype != dtype:
        raise S2GError(f"{name}: expected dtype {dtype}, got {t.dtype}")
    if not t.is_contiguous():
        raise S2GError(f"{name}: expected a contiguous tensor")
    return t


def ptr(t: Optional[torch.Tensor], off_elems: int = 0) -> Optional[int]:
    if t is None:
        return None
    p = t.data_ptr()
    return p + off_elems * t.element_size() if off_elems else p


# ----------------------------------------------------------------------------- GEMM
def gemm_raw(A: int, sam: int, sak: int, B: int, sbk: int, sbn: int, C: int, ldc: int, M: int, N: int, K: int,
             bias: Optional[int] = None, alpha: float = 1.0, accumulate: bool = False, act: int = 0,
             nsplit: int = 1, slab: int = 0, segT: int = 0, a_seg: int = 0, b_seg: int = 0, shift: bool = False,
             kchunk: int = 0) -> None:
    g = S2GGemm(A, sam, sak, B, sbk, sbn, C, ldc, slab, bias, M, N, K, kchunk, segT, a_seg, b_seg, int(shift),
                int(accumulate), act, alpha)
    call("s2g_gemm_f32", ctypes.byref(g), nsplit, stream_ptr())


def gemm_tall(A: int, lda: int, B: int, ldb: int, C: int, ldc: int, M: int, N: int, K: int,
              bf16: bool = False) -> None:
    """C[M,N] = A[M,K] B[K,N], row-major: the tall products of the critic's input layer.  Shapes the
    tall kernel does not cover (N outside (160, 224], B rows not 16-byte) go to the generic kernel.
    bf16: operands rounded to bf16, fp32 accumulation (the reduced-precision path; K, N <= 224)."""
    if bf16 and N <= 224 and K <= 224:
        call("s2g_gemm_tall_bf16", A, lda, B, ldb, C, ldc, M, N, K, stream_ptr())
    elif 160 < N <= 224 and ldb % 4 == 0 and B % 16 == 0:
        call("s2g_gemm_tall", A, lda, B, ldb, C, ldc, M, N, K, stream_ptr())
    else:
        gemm_raw(A, lda, 1, B, ldb, 1, C, ldc, M, N, K)


def pick_split(M: int, N: int, K: int, target_blocks: int = 768, min_chunk: int = 128) -> int:
    """Split-K factor of a product with M x N outputs over K: enough blocks to fill the chip, at least
    `min_chunk` reduction rows per block."""
    tiles = ((M + 63) // 64) * ((N + 63) // 64)
    ns = max(1, min(target_blocks // max(tiles, 1), (K + min_chunk - 1) // min_chunk))
    return ns


def reduce_slabs(slabs: torch.Tensor, slab: int, nslab: int, rows: int, cols: int, out: torch.Tensor,
                 permute_rows: bool = False, accumulate: bool = False, scale: float = 1.0) -> None:
    if out.numel() < rows * cols:
        raise S2GError("reduce_slabs: output too small")
    call("s2g_reduce_slabs", ptr(slabs), slab, nslab, rows, cols, ptr(out), int(permute_rows), int(accumulate),
         scale, stream_ptr())


class Scratch:
    """Grow-only fp32 scratch buffer for split-K slabs (device memory only; no compute)."""

    def __init__(self, device):
        self.device = device
        self.buf = torch.empty(0, dtype=torch.float32, device=device)

    def get(self, n: int) -> torch.Tensor:
        if self.buf.numel() < n:
            self.buf = torch.empty(int(n * 1.25) + 1024, dtype=torch.float32, device=self.device)
        return self.buf


def linear_fwd(x: torch.Tensor, W: torch.Tensor, b: Optional[torch.Tensor], y: torch.Tensor, act: int = 0) -> None:
    """y[M,N] = act(x[M,K] @ W[N,K]^T + b)   (nn.Linear forward)"""
    M, K = x.shape
    N = W.shape[0]
    _chk(x, name="x"); _chk(W, name="W"); _chk(y, name="y")
    if W.shape[1] != K or tuple(y.shape) != (M, N):
        raise S2GError("linear_fwd: shape mismatch")
    gemm_raw(ptr(x), K, 1, ptr(W), 1, K, ptr(y), N, M, N, K, bias=ptr(b), act=act)


def linear_fwd_small_batch(x: torch.Tensor, W: torch.Tensor, b: Optional[torch.Tensor], y: torch.Tensor,
                           workgroups: int = 0) -> None:
    """y[M,N] = x[M,K] @ W[N,K]^T + b for a batch M <= 64 and a large N (the generator's output layer): the
    weight is streamed from HBM exactly once; other shapes go to linear_fwd.  workgroups: CUs the current
    stream can reach if it is CU-masked (0: the whole device)."""
    M, K = x.shape
    N = W.shape[0]
    if M > 64 or N < 4096 or N % 16 or K % 128 or K > 512:
        linear_fwd(x, W, b, y)
        return
    _chk(x, name="x"); _chk(W, name="W"); _chk(y, name="y")
    if W.shape[1] != K or tuple(y.shape) != (M, N):
        raise S2GError("linear_fwd_small_batch: shape mismatch")
    call("s2g_linear_small_batch", ptr(W), ptr(x), ptr(b), ptr(y), N, M, K, int(workgroups), stream_ptr())


def linear_wgrad_adam_ok(batch: int, K: int) -> bool:
    """Shapes the fused weight-gradient + Adam kernel takes (s2g_linear_wgrad_adam)."""
    return batch <= 64 and K in (256, 512)


def linear_wgrad_adam(dy: torch.Tensor, a: torch.Tensor, W: torch.Tensor, m: torch.Tensor, v: torch.Tensor, lr: float,
                      b1: float, b2: float, eps: float, step: int, grad: Optional[torch.Tensor] = None,
                      workgroups: int = 0) -> None:
    """Adam step of W[N,K] on the gradient dy[M,N]^T @ a[M,K] without materialising it (grad: optional
    tensor that also receives the gradient).  workgroups: CUs the current stream can reach (0: all)."""
    M, N = dy.shape
    K = a.shape[1]
    for t, n in ((dy, "dy"), (a, "a"), (W, "W"), (m, "m"), (v, "v")):
        _chk(t, name=n)
    if grad is not None:
        _chk(grad, name="grad")
    if a.shape[0] != M or W.numel() != N * K or m.numel() != N * K or v.numel() != N * K or \
            (grad is not None and grad.numel() != N * K):
        raise S2GError("linear_wgrad_adam: shape mismatch")
    if not linear_wgrad_adam_ok(M, K):
        raise S2GError("linear_wgrad_adam: batch <= 64 and K = 256 or 512 required")
    call("s2g_linear_wgrad_adam", ptr(dy), ptr(a), ptr(W), ptr(m), ptr(v), ptr(grad), N, M, K, lr, b1, b2, eps,
         int(step), int(workgroups), stream_ptr())


def linear_bwd_dx(dy: torch.Tensor, W: torch.Tensor, dx: torch.Tensor, scratch: Scratch) -> None:
    """dx[M,K] = dy[M,N] @ W[N,K]"""
    M, N = dy.shape
    K = W.shape[1]
    _chk(dy, name="dy"); _chk(W, name="W"); _chk(dx, name="dx")
    # a one-tile-row product (the generator's batch of 64) is a chain of dependent load rounds, one per 32
    # reduction rows of a block; beside an HBM-saturating kernel a round costs several microseconds, so such a
    # product is cut down to ONE round per block (32 rows) and one more for the slab sum
    ns = pick_split(M, K, N, min_chunk=32 if M <= 64 else 128)
    if ns == 1:
        gemm_raw(ptr(dy), N, 1, ptr(W), K, 1, ptr(dx), K, M, K, N)
    else:
        slabs = scratch.get(ns * M * K)
        gemm_raw(ptr(dy), N, 1, ptr(W), K, 1, ptr(slabs), K, M, K, N, nsplit=ns, slab=M * K)
        reduce_slabs(slabs, M * K, ns, M, K, dx)


def linear_bwd_dw(dy: torch.Tensor, x: torch.Tensor, dW: torch.Tensor, scratch: Scratch,
                  accumulate: bool = False) -> None:
    """dW[N,K] (+)= dy[M,N]^T @ x[M,K]"""
    M, N = dy.shape
    K = x.shape[1]
    _chk(dy, name="dy"); _chk(x, name="x"); _chk(dW, name="dW")
    ns = pick_split(N, K, M)
    if ns == 1:
        gemm_raw(ptr(dy), 1, N, ptr(x), K, 1, ptr(dW), K, N, K, M, accumulate=accumulate)
    else:
        slabs = scratch.get(ns * N * K)
        gemm_raw(ptr(dy), 1, N, ptr(x), K, 1, ptr(slabs), K, N, K, M, nsplit=ns, slab=N * K)
        reduce_slabs(slabs, N * K, ns, N, K, dW, accumulate=accumulate)


def small_linear_grads(jobs) -> None:
    """jobs: up to four (dy[M,N], x[M,K], dW[N,K], db[N] or None) of small Linear layers sharing the batch M:
    dW = dy^T x and db = column sums of dy, all in one launch (s2g_small_linear_grads)."""
    import ctypes
    n = len(jobs)
    if not 1 <= n <= 4:
        raise S2GError("small_linear_grads: 1..4 jobs")
    M = jobs[0][0].shape[0]
    for dy, x, dW, db in jobs:
        for t, nm in ((dy, "dy"), (x, "x"), (dW, "dW")):
            _chk(t, name=nm)
        if db is not None:
            _chk(db, name="db")
        if dy.shape[0] != M or x.shape[0] != M or tuple(dW.shape) != (dy.shape[1], x.shape[1]) or \
                (db is not None and db.numel() != dy.shape[1]):
            raise S2GError("small_linear_grads: shape mismatch")
    PA, IA = ctypes.c_void_p * n, ctypes.c_int * n
    call("s2g_small_linear_grads", n, PA(*[ptr(j[0]) for j in jobs]), PA(*[ptr(j[1]) for j in jobs]),
         PA(*[ptr(j[2]) for j in jobs]), PA(*[ptr(j[3]) for j in jobs]), IA(*[j[0].shape[1] for j in jobs]),
         IA(*[j[1].shape[1] for j in jobs]), M, stream_ptr())


def colsum(X: torch.Tensor, rows: int, cols: int, ld: int, out: torch.Tensor, scratch: Scratch,
           permute_rows: bool = False, accumulate: bool = False, row0: int = 0) -> None:
    """out[c] (+)= sum_r X[row0 + r][c]"""
    rpc = max(64, (rows + 255) // 256)
    nch = (rows + rpc - 1) // rpc
    if nch == 1 and not permute_rows and not accumulate:
        # a single chunk of rows: its "partial" sums are the result (saves the reduce launch)
        call("s2g_colsum", ptr(X, row0 * ld), ld, rows, cols, ptr(out), rpc, stream_ptr())
        return
    part = scratch.get(nch * cols)
    call("s2g_colsum", ptr(X, row0 * ld), ld, rows, cols, ptr(part), rpc, stream_ptr())
    if permute_rows:
        reduce_slabs(part, cols, nch, cols, 1, out, permute_rows=True, accumulate=accumulate)
    else:
        reduce_slabs(part, cols, nch, 1, cols, out, accumulate=accumulate)


# ----------------------------------------------------------------------------- random draws
def _splitmix64(x: int) -> int:
    m = 0xFFFFFFFFFFFFFFFF
    x = (x + 0x9E3779B97F4A7C15) & m
    x = ((x ^ (x >> 30)) * 0xBF58476D1CE4E5B9) & m
    x = ((x ^ (x >> 27)) * 0x94D049BB133111EB) & m
    return x ^ (x >> 31)


class DeviceRng:
    """Seed and running offset of the library's counter-based generator (Philox4x32-10): every draw is a
    pure function of (seed, offset, position) and takes the next offset, so no generator state lives on
    the device.  Stands in for the reference's CPU-generator draws (wgan_gp.py:58, 228, 291), which no
    device generator reproduces; parity runs inject their noise instead."""

    _instances = 0

    def __init__(self, seed: Optional[int] = None):
        """seed None: torch.initial_seed() mixed with the number of default-seeded generators made so far in
        this process and with the rank of the process, so that the engines of one process and the ranks of a
        data-parallel run (all seeded alike by torch.manual_seed) draw from different streams.  An explicit
        seed is taken as is (the draws are then a pure function of it: tests and replays rely on that)."""
        if seed is None:
            DeviceRng._instances += 1
            rank = 0
            try:
                import torch.distributed as dist
                if dist.is_available() and dist.is_initialized():
                    rank = dist.get_rank()
            except Exception:      # noqa: BLE001
                rank = 0
            seed = torch.initial_seed() ^ _splitmix64((DeviceRng._instances << 20) + rank + 1)
        self.seed = int(seed) & 0xFFFFFFFFFFFFFFFF
        self.offset = 0

    def take(self):
        o = self.offset
        self.offset += 1
        return self.seed, o


def rng_fill(out: torch.Tensor, kind: int, rng: DeviceRng) -> torch.Tensor:
    """out <- uniform [0,1) (kind 0) or standard normal (kind 1) draws."""
    _chk(out, name="out")
    seed, off = rng.take()
    call("s2g_rng_fill", ptr(out), out.numel(), kind, seed, off, stream_ptr())
    return out


def dropout_mask_rng(rows: int, p: float, device, rng: DeviceRng, width: int = MASKW, live: int = H) -> torch.Tensor:
    """[rows, width] multipliers (0 or 1/(1-p); columns >= live zero) of an LSTM's inter-layer dropout
    (default: the H = 50 critic's rows padded to 52)."""
    m = torch.empty(rows, width, device=device)
    seed, off = rng.take()
    call("s2g_dropout_mask_rng", ptr(m), rows, width, live, float(p), seed, off, stream_ptr())
    return m


# ----------------------------------------------------------------------------- sampler
MODES = {"gumbel_softmax": 0, "relaxed_bernoulli": 1}


def gumbel_softmax_fwd(logits: torch.Tensor, U: Optional[torch.Tensor], tau: float, mode: int, hard: bool,
                       bias: Optional[torch.Tensor] = None, bias_rows: int = 0,
                       out: Optional[torch.Tensor] = None, rng: Optional[DeviceRng] = None):
    """rng (mode 0 with U None): the uniforms are drawn inside the kernel."""
    _chk(logits, name="logits")
    V = logits.shape[-1]
    rows = logits.numel() // V
    if U is not None:
        _chk(U, name="U")
        if U.numel() != logits.numel():
            raise S2GError("gumbel_softmax: U must have the shape of logits")
    if out is not None:
        _chk(out, name="out")
        if out.numel() != logits.numel():
            raise S2GError("gumbel_softmax: out must have the size of logits")
    y = out.view_as(logits) if out is not None else torch.empty_like(logits)
    y_soft = torch.empty_like(logits) if hard else None
    if mode == 0 and U is None:
        if rng is None:
            raise S2GError("gumbel_softmax: the Gumbel mode needs the uniforms U or a DeviceRng")
        seed, off = rng.take()
        call("s2g_gumbel_softmax_fwd_rng", ptr(logits), ptr(bias), bias_rows, ptr(y), ptr(y_soft), rows, V, float(tau),
             int(hard), seed, off, stream_ptr())
        return y, (y_soft if hard else y)
    call("s2g_gumbel_softmax_fwd", ptr(logits), ptr(U), ptr(bias), bias_rows, ptr(y), ptr(y_soft), rows, V,
         float(tau), mode, int(hard), stream_ptr())
    return y, (y_soft if hard else y)


def gumbel_softmax_bwd(y_soft: torch.Tensor, dy: torch.Tensor, scale: float) -> torch.Tensor:
    _chk(y_soft, name="y"); _chk(dy, name="dy")
    V = y_soft.shape[-1]
    dl = torch.empty_like(y_soft)
    call("s2g_gumbel_softmax_bwd", ptr(y_soft), ptr(dy), ptr(dl), y_soft.numel() // V, V, float(scale), stream_ptr())
    return dl


def argmax_rows(y: torch.Tensor) -> torch.Tensor:
    _chk(y, name="y")
    V = y.shape[-1]
    tok = torch.empty(y.shape[:-1], dtype=torch.int64, device=y.device)
    call("s2g_argmax_rows", ptr(y), ptr(tok), y.numel() // V, V, stream_ptr())
    return tok


# ----------------------------------------------------------------------------- optimiser
def adam(p: torch.Tensor, g: torch.Tensor, m: torch.Tensor, v: torch.Tensor, lr: float, b1: float, b2: float,
         eps: float, step: int, gscale: float = 1.0) -> None:
    for t, n in ((p, "p"), (g, "g"), (m, "m"), (v, "v")):
        _chk(t, name=n)
        if t.numel() != p.numel():
            raise S2GError("adam: size mismatch")
    call("s2g_adam", ptr(p), ptr(g), ptr(m), ptr(v), p.numel(), lr, b1, b2, eps, step, gscale, stream_ptr())
