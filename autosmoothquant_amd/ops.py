"""Thin torch-tensor wrappers over the C-ABI (device memory + current stream plumbing).

Every function validates dtype / device / contiguity / shape and raises instead of
falling back: the reference passes raw ``data_ptr`` with no checks (bindings.cpp:73-80)."""
import threading

import os

import torch

from . import _lib as L

_DT = {torch.float32: L.ASQ_F32, torch.float16: L.ASQ_F16, torch.bfloat16: L.ASQ_BF16}
_ACT = {"per-tensor-round": L.ASQ_ACT_ROUND, "per-tensor-div": L.ASQ_ACT_DIV, "per-token": L.ASQ_ACT_PER_TOKEN}


def _dev(t, name):
    if not isinstance(t, torch.Tensor):
        raise TypeError(f"{name} must be a torch.Tensor")
    if not t.is_cuda:
        raise RuntimeError(f"{name} is on {t.device}: autosmoothquant_amd ops need a HIP (cuda:N) tensor; there is no CPU fallback")
    if not t.is_contiguous():
        raise ValueError(f"{name} must be contiguous")
    return t


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)


def _stream(t):
    """hipStream_t of torch's current stream on t's device, as an int.  The raw getter costs ~0.3 us; the public
    torch.cuda.current_stream(...).cuda_stream ~4.5 us, which is noticeable next to a 6 us decode GEMM."""
    if _raw_stream is not None:
        return _raw_stream(t.device.index)
    return torch.cuda.current_stream(t.device).cuda_stream


class _NoGuard:
    def __enter__(self):
        return None

    def __exit__(self, *exc):
        return False


_NO_GUARD = _NoGuard()


def _on(dev):
    """Make `dev` the current HIP device for the launch (the C-ABI launches on the current device): a no-op
    object when it already is -- torch.cuda.device(...) costs ~4 us per call even then."""
    return _NO_GUARD if torch.cuda.current_device() == dev.index else torch.cuda.device(dev)


def _ptr(t):
    return None if t is None else t.data_ptr()


_WS_PERSIST_MAX = 64 << 20   # beyond that (split-K slabs of large-M calls) a fresh buffer per call; decode-sized calls never get there
_ws_tls = threading.local()


def _workspace(lib, size_fn, M, N, K, dev, stream):
    """Caller-owned scratch of the GEMM entry points, kept per (thread, device, stream).

    The C-ABI's workspace contract (include/asq_hip.h): a workspace starts with a header (magic word + the arrival tickets of the weight-streaming
    kernel's in-launch reduction) that asq_workspace_init() writes ONCE; every launch leaves it clean, one launch at a time may use a buffer.  Hence:
    the buffer is persistent (not torch.empty per call: a fresh allocation has no header), initialised on the stream it is used on when it is
    created, and never shared across streams or threads.  Same-stream launches are ordered, so consecutive calls of any shape share it.  A buffer
    that is outgrown is retired, not freed: a captured hipGraph may still replay launches that point at it.  While the stream is being CAPTURED
    nothing is created in the cache (an asq_workspace_init recorded into a graph has not run: the next eager call on that stream handle would
    find a header-less buffer and trap): a capture without a warmed-up buffer gets a graph-owned one whose init is part of the graph.
    Returns (tensor | None, nbytes)."""
    cache = _ws_tls.__dict__.setdefault("c", {})
    key = (dev.index, stream, size_fn, M, N, K)
    hit = cache.get(key)
    if hit is not None:
        return hit
    n = getattr(lib, size_fn)(M, N, K)
    capturing = n != 0 and torch.cuda.is_current_stream_capturing()
    if capturing:
        slot = cache.get((dev.index, stream))
        if slot is not None and slot.numel() >= n:
            return slot, n
    if n == 0:
        res = (None, 0)
    elif n > _WS_PERSIST_MAX or capturing:
        buf = torch.empty((n,), dtype=torch.uint8, device=dev)
        L.check(lib.asq_workspace_init(buf.data_ptr(), n, stream), "asq_workspace_init")
        return buf, n          # (not cached: one buffer per call, header written ahead of the launch on the same stream)
    else:
        slot = cache.get((dev.index, stream))
        if slot is None or slot.numel() < n:
            if slot is not None:
                _ws_tls.__dict__.setdefault("retired", []).append(slot)
            slot = torch.empty((max(2 * n, 1 << 20),), dtype=torch.uint8, device=dev)
            L.check(lib.asq_workspace_init(slot.data_ptr(), slot.numel(), stream), "asq_workspace_init")
            cache[(dev.index, stream)] = slot
            for k in [k for k in cache if len(k) == 6 and k[:2] == key[:2]]:   # shape entries of this stream point at the old slot
                del cache[k]
        res = (slot, n)
    if len(cache) > 512:
        for k in [k for k in cache if len(k) == 6]:
            del cache[k]
    cache[key] = res
    return res


def release_workspaces():
    """Drop this thread's cached workspaces (per-stream GEMM slots, retired slots, the 64 MiB grouped-launch buffers).  They are otherwise kept for the life of
    the thread: a captured hipGraph may replay launches that point at them.  Call it only when no captured graph that used this thread's ops will be replayed
    again and the streams are idle; the next call re-creates (and re-initialises) what it needs.
    Graph note: a graph captured on stream A bakes A's slot into its launches -- replay it serialised with A (same stream, or ordered by events); replayed
    concurrently with eager calls or another graph on A it shares one ticket header with them, which the kernels answer with a trap."""
    _ws_tls.__dict__.pop("c", None)
    _ws_tls.__dict__.pop("retired", None)


def _grouped_ws(lib, M, N, K, G, dev, stream):
    """workspace of a grouped launch (asq_grouped_workspace_bytes: header + 64 MiB of K-piece images, one size for every shape that uses it):
    one persistent, initialised buffer per (thread, device, stream), separate from the GEMM slot; under stream capture without a warmed-up
    buffer the launch simply runs without one (same results, no tail split)."""
    n = lib.asq_grouped_workspace_bytes(M, N, K, G)
    if n == 0:
        return None, 0
    cache = _ws_tls.__dict__.setdefault("c", {})
    key = (dev.index, stream, "grouped")
    buf = cache.get(key)
    if buf is None or buf.numel() < n:
        if torch.cuda.is_current_stream_capturing():
            return None, 0
        buf = torch.empty((n,), dtype=torch.uint8, device=dev)
        L.check(lib.asq_workspace_init(buf.data_ptr(), n, stream), "asq_workspace_init")
        cache[key] = buf
    return buf, n


def _gemm_ws(M, N, K, dev, stream):
    """scratch of a GEMM-only call (None when the shape never needs any)"""
    return _workspace(L.lib(), "asq_gemm_workspace_bytes", M, N, K, dev, stream)


def _forward_ws(lib, M, N, K, dev, stream):
    """workspace of asq_linear_w8a8_forward (GEMM scratch + int8 activation + row scales)"""
    return _workspace(lib, "asq_linear_w8a8_workspace_bytes", M, N, K, dev, stream)


def _same_device(*ts):
    d = None
    for t in ts:
        if t is None:
            continue
        if d is None:
            d = t.device
        elif t.device != d:
            raise RuntimeError(f"tensors on different devices: {d} vs {t.device}")
    return d


def gemm_i8_i32(x, w, out):
    """out[M,N] i32 = x[M,K] i8 . w[N,K]^T i8  (K1, bindings.cpp:69-84)."""
    _dev(x, "input"), _dev(w, "weight"), _dev(out, "out")
    if x.dtype != torch.int8 or w.dtype != torch.int8 or out.dtype != torch.int32:
        raise RuntimeError("expected int8 input, int8 weight, int32 out")
    if x.dim() != 2 or w.dim() != 2 or x.shape[1] != w.shape[1] or tuple(out.shape) != (x.shape[0], w.shape[0]):
        raise ValueError(f"shape mismatch: input {tuple(x.shape)}, weight {tuple(w.shape)}, out {tuple(out.shape)}")
    dev = _same_device(x, w, out)
    with _on(dev):
        ws, n = _gemm_ws(x.shape[0], w.shape[0], x.shape[1], dev, _stream(x))
        L.check(L.lib().asq_gemm_i8_i32(x.data_ptr(), w.data_ptr(), out.data_ptr(), x.shape[0], w.shape[0], x.shape[1], _ptr(ws), n,
                                        _stream(x)), "asq_gemm_i8_i32")
    return out


def gemm_i8_i8(x, w, out, alpha, beta=0.0):
    """out[M,N] i8 = sat(rne(alpha*acc + beta*out))  (K3-K5, bindings.cpp:86-142)."""
    _dev(x, "input"), _dev(w, "weight"), _dev(out, "out")
    if x.dtype != torch.int8 or w.dtype != torch.int8 or out.dtype != torch.int8:
        raise RuntimeError("expected int8 input, weight and out")
    if x.dim() != 2 or w.dim() != 2 or x.shape[1] != w.shape[1] or tuple(out.shape) != (x.shape[0], w.shape[0]):
        raise ValueError(f"shape mismatch: input {tuple(x.shape)}, weight {tuple(w.shape)}, out {tuple(out.shape)}")
    dev = _same_device(x, w, out)
    with _on(dev):
        ws, n = _gemm_ws(x.shape[0], w.shape[0], x.shape[1], dev, _stream(x))
        L.check(L.lib().asq_gemm_i8_i8(x.data_ptr(), w.data_ptr(), out.data_ptr(), x.shape[0], w.shape[0], x.shape[1],
                                       float(alpha), float(beta), _ptr(ws), n, _stream(x)), "asq_gemm_i8_i8")
    return out


def quantize_act(x, mode, quant_scale=1.0):
    """x [M,K] f32/f16/bf16 -> (xq int8 [M,K], s_row f32 [M] or None).  mode in
    {"per-token", "per-tensor-round", "per-tensor-div"} (linear.py:88-96, 283-292)."""
    _dev(x, "x")
    if x.dtype not in _DT or x.dim() != 2:
        raise ValueError("x must be a 2-D float32/float16/bfloat16 tensor")
    M, K = x.shape
    xq = torch.empty((M, K), dtype=torch.int8, device=x.device)
    s_row = torch.empty((M,), dtype=torch.float32, device=x.device) if mode == "per-token" else None
    with _on(x.device):
        L.check(L.lib().asq_quantize_act(x.data_ptr(), _DT[x.dtype], _ACT[mode], float(quant_scale), xq.data_ptr(), _ptr(s_row),
                                         M, K, _stream(x)), "asq_quantize_act")
    return xq, s_row


def offsets_supported(M, N, K, out_dtype):
    """True when the C-ABI would run a [M,K] x [N,K]^T linear with `out_dtype` outputs on offset operand images (include/asq_hip.h)."""
    return out_dtype in _DT and bool(L.lib().asq_offsets_supported(M, N, K, _DT[out_dtype]))


def forward_is_fused(M, N, K, dtype, mode="per-tensor-round"):
    """True when linear_w8a8_forward runs this shape / quantiser mode as ONE launch (activation quantiser = the GEMM's prologue; asq_forward_fused_supported)."""
    return dtype in _DT and bool(L.lib().asq_forward_fused_supported(int(M), int(N), int(K), _DT[dtype], _ACT[mode]))


def linear_w8a8_forward_fused(x2d, w, act_mode, quant_scale, s_scalar, s_col=None, bias=None):
    """linear_w8a8_forward as ONE launch, on request (asq_linear_w8a8_forward_fused): any shape the fused kernel can run (<= 16 rows, K % 128 == 0, the int8 activation
    image fits 64 KiB of LDS), also where linear_w8a8_forward itself would take two launches.  Raises on shapes outside the kernel's limits."""
    _dev(x2d, "x"), _dev(w, "weight")
    if x2d.dtype not in _DT:
        raise ValueError(f"unsupported activation dtype {x2d.dtype}")
    if w.dtype != torch.int8 or x2d.dim() != 2 or w.dim() != 2 or x2d.shape[1] != w.shape[1] or not x2d.is_contiguous() or not w.is_contiguous():
        raise ValueError(f"shape/dtype mismatch: x {tuple(x2d.shape)} {x2d.dtype}, weight {tuple(w.shape)} {w.dtype} (both contiguous)")
    M, K = x2d.shape
    N = w.shape[0]
    for name, t in (("s_col", s_col), ("bias", bias)):
        if t is not None:
            _dev(t, name)
            if t.dtype != torch.float32 or t.numel() != N:
                raise ValueError(f"{name} must be float32 with {N} elements")
    dev = _same_device(x2d, w, s_col, bias)
    out = torch.empty((M, N), dtype=x2d.dtype, device=dev)
    with _on(dev):
        L.check(L.lib().asq_linear_w8a8_forward_fused(x2d.data_ptr(), _DT[x2d.dtype], w.data_ptr(), out.data_ptr(), M, N, K, _ACT[act_mode], float(quant_scale), float(s_scalar),
                                                      _ptr(s_col), _ptr(bias), _stream(x2d)), "asq_linear_w8a8_forward_fused")
    return out


def gate_up_supported(M, F, K, dtype):
    """True when linear_w8a8_gate_up can run [M, K] x (gate || up of F channels each) with `dtype` outputs (asq_gate_up_supported)."""
    return dtype in _DT and bool(L.lib().asq_gate_up_supported(int(M), int(F), int(K), _DT[dtype]))


def interleave_gate_up(w_gate, w_up, out=None, block=None):
    """[F, K] gate and up -> the [2 F, K] operand of the gate || up GEMMs: blocks of `block` gate channels alternate with the same channels of up (include/asq_hip.h).
    int8 weights: block 16 (the 16 x 16 x 64 instruction's accumulator tile); float8 weights: block 32 (the 32 x 32 block-scaled instruction's)."""
    F_, K = w_gate.shape
    f8 = (torch.float8_e4m3fn, torch.float8_e5m2)
    if w_up.shape != w_gate.shape or w_up.dtype != w_gate.dtype or w_gate.dtype not in (torch.int8,) + f8:
        raise ValueError("gate / up must be int8 (or float8) [F, K] tensors of one shape and dtype")
    if block is None:
        block = 16 if w_gate.dtype == torch.int8 else 32
    if F_ % block != 0:
        raise ValueError(f"gate / up: F % {block} == 0 expected")
    if out is None:
        out = torch.empty((2 * F_, K), dtype=w_gate.dtype, device=w_gate.device)
    v = out.view(torch.uint8).view(F_ // block, 2, block, K)
    v[:, 0].copy_(w_gate.view(torch.uint8).view(F_ // block, block, K))
    v[:, 1].copy_(w_up.view(torch.uint8).view(F_ // block, block, K))
    return out


def linear_w8a8_gate_up(xq, w_gu, out_dtype, s_gate, s_up, s_row=None, fast=None, row_off=None, col_off=None, out=None):
    """SiLU(gate(x)) * up(x) in ONE GEMM over the interleaved weight (asq_linear_w8a8_gate_up): out [M, F] in out_dtype, bit-identical to
    linear_w8a8 (gate), linear_w8a8 (up) and the SiLU * up of silu_mul_quantize(fast=...).  row_off / col_off: offset operand images of xq and w_gu."""
    _dev(xq, "xq"), _dev(w_gu, "w_gu")
    if xq.dtype != torch.int8 or w_gu.dtype != torch.int8 or xq.dim() != 2 or w_gu.dim() != 2 or xq.shape[1] != w_gu.shape[1] or w_gu.shape[0] % 2:
        raise ValueError("xq [M,K] and w_gu [2F,K] must be int8 with equal K")
    if not (xq.is_contiguous() and w_gu.is_contiguous()):
        raise ValueError("xq / w_gu must be contiguous")
    M, K = xq.shape
    F_ = w_gu.shape[0] // 2
    if s_row is not None:
        _dev(s_row, "s_row")
        if s_row.dtype != torch.float32 or s_row.numel() != M:
            raise ValueError(f"s_row must be float32 with {M} elements")
    if (row_off is None) != (col_off is None):
        raise ValueError("row_off and col_off come together")
    if row_off is not None and (row_off.dtype != torch.int32 or row_off.numel() != 2 * M or col_off.dtype != torch.int32 or col_off.numel() != 4 * F_):
        raise ValueError("row_off must be int32 [M,2] and col_off int32 [2F,2]")
    if fast is None:
        fast = not SILU_EXACT_DEFAULT
    if out is None:
        out = torch.empty((M, F_), dtype=out_dtype, device=xq.device)
    dev = _same_device(xq, w_gu, out, s_row, row_off, col_off)
    with _on(dev):
        L.check(L.lib().asq_linear_w8a8_gate_up(xq.data_ptr(), w_gu.data_ptr(), out.data_ptr(), _DT[out_dtype], M, F_, K, float(s_gate), float(s_up), _ptr(s_row),
                                                L.ASQ_SILU_FAST if fast else 0, _ptr(row_off), _ptr(col_off), _stream(xq)), "asq_linear_w8a8_gate_up")
    return out


def linear_w8a8_gate_up_q8(xq, w_gu, act_dtype, s_gate, s_up, quant_scale, s_row=None, fast=None, row_off=None, col_off=None):
    """linear_w8a8_gate_up with the consumer's per-tensor quantiser as part of the epilogue (asq_linear_w8a8_gate_up_q8): int8 [M, F] =
    quantize_act(linear_w8a8_gate_up(..., act_dtype), "per-tensor-div", quant_scale)[0], bit for bit; the fp tensor never exists."""
    _dev(xq, "xq"), _dev(w_gu, "w_gu")
    if xq.dtype != torch.int8 or w_gu.dtype != torch.int8 or xq.dim() != 2 or w_gu.dim() != 2 or xq.shape[1] != w_gu.shape[1] or w_gu.shape[0] % 2:
        raise ValueError("xq [M,K] and w_gu [2F,K] must be int8 with equal K")
    if not (xq.is_contiguous() and w_gu.is_contiguous()):
        raise ValueError("xq / w_gu must be contiguous")
    M, K = xq.shape
    F_ = w_gu.shape[0] // 2
    if s_row is not None:
        _dev(s_row, "s_row")
        if s_row.dtype != torch.float32 or s_row.numel() != M:
            raise ValueError(f"s_row must be float32 with {M} elements")
    if (row_off is None) != (col_off is None):
        raise ValueError("row_off and col_off come together")
    if row_off is not None and (row_off.dtype != torch.int32 or row_off.numel() != 2 * M or col_off.dtype != torch.int32 or col_off.numel() != 4 * F_):
        raise ValueError("row_off must be int32 [M,2] and col_off int32 [2F,2]")
    if fast is None:
        fast = not SILU_EXACT_DEFAULT
    out = torch.empty((M, F_), dtype=torch.int8, device=xq.device)
    dev = _same_device(xq, w_gu, out, s_row, row_off, col_off)
    with _on(dev):
        L.check(L.lib().asq_linear_w8a8_gate_up_q8(xq.data_ptr(), w_gu.data_ptr(), out.data_ptr(), _DT[act_dtype], M, F_, K, float(s_gate), float(s_up), _ptr(s_row),
                                                   L.ASQ_SILU_FAST if fast else 0, float(quant_scale), _ptr(row_off), _ptr(col_off), _stream(xq)), "asq_linear_w8a8_gate_up_q8")
    return out


def grouped_gate_up_supported(M, F_, K, out_dtype):
    return out_dtype in (torch.float16, torch.bfloat16) and bool(L.lib().asq_grouped_gate_up_supported(M, F_, K, _DT[out_dtype]))


def interleave_gate_up_stack(w1, w3, out=None):
    """[G, 2F, K] int8: every group's gate (w1[g]) and up (w3[g]) rows interleaved in blocks of 16 channels (interleave_gate_up per group).
    out: rebuild INTO this [G, 2F, K] buffer (a hipGraph captured on it then replays the new weights)."""
    if w1.shape != w3.shape or w1.dim() != 3:
        raise ValueError("w1 / w3 must be [G, F, K] stacks of one shape")
    G, F_, K = w1.shape
    if out is None:
        out = torch.empty((G, 2 * F_, K), dtype=w1.dtype, device=w1.device)
    elif tuple(out.shape) != (G, 2 * F_, K) or out.dtype != w1.dtype or out.device != w1.device or not out.is_contiguous():
        raise ValueError("out must be a contiguous [G, 2F, K] buffer of the stacks' dtype on their device")
    for g in range(G):
        interleave_gate_up(w1[g], w3[g], out=out[g])
    return out


def linear_w8a8_grouped_gate_up(xq, w_gu, group_offsets, s_gate, s_up, out_dtype, fast=None, row_off=None, col_off=None):
    """SiLU(w1 x) * (w3 x) of ALL groups in ONE grouped launch over the interleaved stacks w_gu [G, 2F, K] (asq_linear_w8a8_grouped_gate_up; reference
    models/mixtral.py:99-101,142-145): out [M, F], bit-identical to linear_w8a8_grouped (w1), (w3) and the SiLU * up of silu_mul_quantize(fast=...).  Per-tensor
    activations.  row_off / col_off: offset operand images of xq and of the stack viewed as [G * 2F, K]."""
    _dev(xq, "xq"), _dev(w_gu, "w_gu"), _dev(group_offsets, "group_offsets"), _dev(s_gate, "s_gate"), _dev(s_up, "s_up")
    if xq.dtype != torch.int8 or w_gu.dtype != torch.int8 or xq.dim() != 2 or w_gu.dim() != 3 or xq.shape[1] != w_gu.shape[2] or w_gu.shape[1] % 2:
        raise ValueError("xq [M,K] int8 and w_gu [G,2F,K] int8 with equal K expected")
    if not (xq.is_contiguous() and w_gu.is_contiguous()):
        raise ValueError("xq / w_gu must be contiguous")
    M, K = xq.shape
    G, F_ = w_gu.shape[0], w_gu.shape[1] // 2
    if group_offsets.dtype != torch.int32 or group_offsets.numel() != G + 1:
        raise ValueError("group_offsets must be int32 [G+1]")
    for name, t in (("s_gate", s_gate), ("s_up", s_up)):
        if t.dtype != torch.float32 or t.numel() != G:
            raise ValueError(f"{name} must be float32 [G]")
    if (row_off is None) != (col_off is None):
        raise ValueError("row_off and col_off come together")
    if row_off is not None and (row_off.dtype != torch.int32 or row_off.numel() != 2 * M or col_off.dtype != torch.int32 or col_off.numel() != 4 * G * F_):
        raise ValueError("row_off must be int32 [M,2] and col_off int32 [G,2F,2]")
    if fast is None:
        fast = not SILU_EXACT_DEFAULT
    out = torch.empty((M, F_), dtype=out_dtype, device=xq.device)
    dev = _same_device(xq, w_gu, group_offsets, s_gate, s_up, row_off, col_off)
    with _on(dev):
        lib, st = L.lib(), _stream(xq)
        ws, n = _grouped_ws(lib, M, 2 * F_, K, G, dev, st)
        L.check(lib.asq_linear_w8a8_grouped_gate_up(xq.data_ptr(), w_gu.data_ptr(), out.data_ptr(), _DT[out_dtype], group_offsets.data_ptr(), G, M, F_, K, s_gate.data_ptr(),
                                                    s_up.data_ptr(), L.ASQ_SILU_FAST if fast else 0, _ptr(row_off), _ptr(col_off), _ptr(ws), n, st), "asq_linear_w8a8_grouped_gate_up")
    return out


def _check_image(image, N, K, device):
    """an offset operand image handed to a C-ABI call: int8 [N,K] + int32 [N,2], contiguous, on `device` (the kernels read both through raw pointers)"""
    w_off, col_off = image
    if (w_off.dtype != torch.int8 or tuple(w_off.shape) != (N, K) or not w_off.is_contiguous() or w_off.device != device
            or col_off.dtype != torch.int32 or tuple(col_off.shape) != (N, 2) or not col_off.is_contiguous() or col_off.device != device):
        raise ValueError(f"offset image must be (int8 [{N}, {K}], int32 [{N}, 2]), contiguous, on {device}; got "
                         f"{tuple(w_off.shape)} {w_off.dtype} {w_off.device} / {tuple(col_off.shape)} {col_off.dtype} {col_off.device}")


def weight_offset_image(w, out=None):
    """w int8 [N,K] -> (w_off int8 [N,K], col_off int32 [N,2] = {cw[n], sum_k w[n,k]}): the weight's offset operand image (asq_weight_offset_image), built once per weight.
    out = (w_off, col_off): rebuild INTO these buffers (a hipGraph captured on them then sees the new image at its next replay)."""
    _dev(w, "weight")
    if w.dtype != torch.int8 or w.dim() != 2:
        raise ValueError("weight must be int8 [N,K]")
    N, K = w.shape
    if out is not None:
        _check_image(out, N, K, w.device)
        w_off, col_off = out
    else:
        w_off = torch.empty_like(w)
        col_off = torch.empty((N, 2), dtype=torch.int32, device=w.device)
    with _on(w.device):
        L.check(L.lib().asq_weight_offset_image(w.data_ptr(), N, K, w_off.data_ptr(), col_off.data_ptr(), _stream(w)), "asq_weight_offset_image")
    return w_off, col_off


def quantize_act_off(x, mode, quant_scale=1.0):
    """quantize_act emitting the offset image: (xq' int8 [M,K], s_row f32 [M] or None, row_off int32 [M,2] = {cx[m], sum_k xq'[m,k]});
    xq' - cx[:, None] is exactly quantize_act's xq."""
    _dev(x, "x")
    if x.dtype not in _DT or x.dim() != 2:
        raise ValueError("x must be a 2-D float32/float16/bfloat16 tensor")
    M, K = x.shape
    xq = torch.empty((M, K), dtype=torch.int8, device=x.device)
    s_row = torch.empty((M,), dtype=torch.float32, device=x.device) if mode == "per-token" else None
    row_off = torch.empty((M, 2), dtype=torch.int32, device=x.device)
    with _on(x.device):
        L.check(L.lib().asq_quantize_act_off(x.data_ptr(), _DT[x.dtype], _ACT[mode], float(quant_scale), xq.data_ptr(), _ptr(s_row), row_off.data_ptr(),
                                             M, K, _stream(x)), "asq_quantize_act_off")
    return xq, s_row, row_off


def linear_w8a8_off(xq_off, w_off, row_off, col_off, out_dtype, s_scalar=1.0, s_row=None, s_col=None, bias=None, order="scale_first", out=None):
    """linear_w8a8 on offset operand images: bit-identical to linear_w8a8 on the plain operands (asq_linear_w8a8_off)."""
    _dev(xq_off, "xq"), _dev(w_off, "weight"), _dev(row_off, "row_off"), _dev(col_off, "col_off")
    if xq_off.dtype != torch.int8 or w_off.dtype != torch.int8 or xq_off.dim() != 2 or w_off.dim() != 2 or xq_off.shape[1] != w_off.shape[1]:
        raise ValueError("xq [M,K] and weight [N,K] must be int8 with equal K")
    M, K = xq_off.shape
    N = w_off.shape[0]
    if row_off.dtype != torch.int32 or row_off.numel() != 2 * M or col_off.dtype != torch.int32 or col_off.numel() != 2 * N:
        raise ValueError("row_off must be int32 [M,2] and col_off int32 [N,2]")
    for name, t, n in (("s_row", s_row, M), ("s_col", s_col, N), ("bias", bias, N)):
        if t is not None:
            _dev(t, name)
            if t.dtype != torch.float32 or t.numel() != n:
                raise ValueError(f"{name} must be float32 with {n} elements")
    if out is None:
        out = torch.empty((M, N), dtype=out_dtype, device=xq_off.device)
    else:
        _dev(out, "out")
        if out.dtype != out_dtype or tuple(out.shape) != (M, N):
            raise ValueError("out has wrong dtype/shape")
        _bump_version(out)
    dev = _same_device(xq_off, w_off, out, row_off, col_off, s_row, s_col, bias)
    with _on(dev):
        L.check(L.lib().asq_linear_w8a8_off(xq_off.data_ptr(), w_off.data_ptr(), out.data_ptr(), _DT[out_dtype], M, N, K, float(s_scalar),
                                            _ptr(s_row), _ptr(s_col), _ptr(bias), L.ASQ_EPI_SCALE_FIRST if order == "scale_first" else L.ASQ_EPI_ACC_FIRST,
                                            row_off.data_ptr(), col_off.data_ptr(), _stream(xq_off)), "asq_linear_w8a8_off")
    return out


def _bump_version(t):
    """A raw C-ABI write into a caller-provided tensor is made visible to torch's version counter (autograd's saved-tensor checks and
    any caller that keys on ._version).  Inference tensors have no counter: nothing to do."""
    try:
        torch.autograd.graph.increment_version(t)
    except Exception:
        pass


def norm_quantize(x, weight, bias=None, eps=1e-5, per_token=False, offsets=False):
    """Fused (scale-folded) RMSNorm / LayerNorm -> int8 activation.  x [M,K]; weight (and bias for
    LayerNorm) [K] in x's dtype.  Returns (xq int8 [M,K], s_row f32 [M] or None); offsets=True emits the offset image
    (asq_norm_quantize_off) and returns (xq', s_row, row_off int32 [M,2])."""
    _dev(x, "x"), _dev(weight, "weight")
    if x.dtype not in _DT or x.dim() != 2 or weight.dtype != x.dtype or weight.numel() != x.shape[1]:
        raise ValueError("x must be 2-D float and weight a [K] tensor of the same dtype")
    if bias is not None:
        _dev(bias, "bias")
        if bias.dtype != x.dtype or bias.numel() != x.shape[1]:
            raise ValueError("bias must match weight")
    M, K = x.shape
    xq = torch.empty((M, K), dtype=torch.int8, device=x.device)
    s_row = torch.empty((M,), dtype=torch.float32, device=x.device) if per_token else None
    row_off = torch.empty((M, 2), dtype=torch.int32, device=x.device) if offsets else None
    with _on(x.device):
        if offsets:
            L.check(L.lib().asq_norm_quantize_off(x.data_ptr(), _DT[x.dtype], weight.data_ptr(), _ptr(bias), float(eps), 1 if per_token else 0,
                                                  xq.data_ptr(), _ptr(s_row), row_off.data_ptr(), M, K, _stream(x)), "asq_norm_quantize_off")
        else:
            L.check(L.lib().asq_norm_quantize(x.data_ptr(), _DT[x.dtype], weight.data_ptr(), _ptr(bias), float(eps), 1 if per_token else 0,
                                              xq.data_ptr(), _ptr(s_row), M, K, _stream(x)), "asq_norm_quantize")
    return (xq, s_row, row_off) if offsets else (xq, s_row)


def add_norm_quantize(x, residual, weight, bias=None, eps=1e-5, per_token=False, out=None, offsets=False):
    """h = residual + x (in x's dtype) and the fused norm -> int8 of h in one pass (the reference's dq_add_layernorm_q,
    csrc/kernels/fused.cu:5-25, on a floating x).  Returns (h [M,K], xq int8 [M,K], s_row f32 [M] or None) -- with offsets=True (h, xq', s_row, row_off int32 [M,2]); `out`
    may be `residual` itself (the residual stream is updated in place).  offsets=True: xq is the offset image (asq_add_norm_quantize_off)."""
    _dev(x, "x"), _dev(residual, "residual"), _dev(weight, "weight")
    if x.dtype not in _DT or x.dim() != 2 or residual.shape != x.shape or residual.dtype != x.dtype:
        raise ValueError("x and residual must be 2-D float tensors of equal shape and dtype")
    if weight.dtype != x.dtype or weight.numel() != x.shape[1]:
        raise ValueError("weight must be a [K] tensor of x's dtype")
    if bias is not None:
        _dev(bias, "bias")
        if bias.dtype != x.dtype or bias.numel() != x.shape[1]:
            raise ValueError("bias must match weight")
    M, K = x.shape
    h = torch.empty_like(x) if out is None else _dev(out, "out")
    if h.shape != x.shape or h.dtype != x.dtype:
        raise ValueError("out has wrong dtype/shape")
    if out is not None:
        _bump_version(out)
    xq = torch.empty((M, K), dtype=torch.int8, device=x.device)
    s_row = torch.empty((M,), dtype=torch.float32, device=x.device) if per_token else None
    row_off = torch.empty((M, 2), dtype=torch.int32, device=x.device) if offsets else None
    with _on(x.device):
        if offsets:
            L.check(L.lib().asq_add_norm_quantize_off(x.data_ptr(), residual.data_ptr(), h.data_ptr(), _DT[x.dtype], weight.data_ptr(), _ptr(bias), float(eps),
                                                      1 if per_token else 0, xq.data_ptr(), _ptr(s_row), row_off.data_ptr(), M, K, _stream(x)), "asq_add_norm_quantize_off")
        else:
            L.check(L.lib().asq_add_norm_quantize(x.data_ptr(), residual.data_ptr(), h.data_ptr(), _DT[x.dtype], weight.data_ptr(), _ptr(bias), float(eps),
                                                  1 if per_token else 0, xq.data_ptr(), _ptr(s_row), M, K, _stream(x)), "asq_add_norm_quantize")
    return (h, xq, s_row, row_off) if offsets else (h, xq, s_row)


SILU_EXACT_DEFAULT = os.environ.get("ASQ_SILU_EXACT", "0") == "1"   # the bit-reproducible SiLU everywhere `fast` is left to the default


def silu_mul_quantize(gate, up, per_token=True, quant_scale=1.0, fast=None, offsets=False):
    """int8(quantise(silu(gate) * up)) in one pass; returns (xq int8 [M,K], s_row f32 [M] or None); offsets=True: (xq', s_row, row_off int32 [M,2]) with xq' the offset image.
    fast (default True since round 5; ASQ_SILU_EXACT=1 flips the default): silu from the hardware transcendentals (C-ABI flag ASQ_SILU_FAST, ~1.5x the throughput: 5.3 vs 3.5 TB/s at
    65536 x 11008).  fast=False: the fixed-operation-order sequence oracle/n1.py reproduces bit for bit.  BOTH forms meet the same bound against what the reference's composition
    computes -- torch F.silu(gate) * up, then the consumer's quantiser: |diff| <= 1 int8 on < 2e-3 of the elements (tests/test_hip_harness.py::test_silu_mul_quant_matches_two_step_path)."""
    if fast is None:
        fast = not SILU_EXACT_DEFAULT
    _dev(gate, "gate"), _dev(up, "up")
    if gate.dtype not in _DT or gate.dim() != 2 or up.dtype != gate.dtype or up.shape != gate.shape:
        raise ValueError("gate and up must be 2-D float tensors of equal shape and dtype")
    M, K = gate.shape
    xq = torch.empty((M, K), dtype=torch.int8, device=gate.device)
    s_row = torch.empty((M,), dtype=torch.float32, device=gate.device) if per_token else None
    row_off = torch.empty((M, 2), dtype=torch.int32, device=gate.device) if offsets else None
    with _on(gate.device):
        if offsets:
            L.check(L.lib().asq_silu_mul_quantize_off(gate.data_ptr(), up.data_ptr(), _DT[gate.dtype], (1 if per_token else 0) | (2 if fast else 0), float(quant_scale),
                                                      xq.data_ptr(), _ptr(s_row), row_off.data_ptr(), M, K, _stream(gate)), "asq_silu_mul_quantize_off")
        else:
            L.check(L.lib().asq_silu_mul_quantize(gate.data_ptr(), up.data_ptr(), _DT[gate.dtype], (1 if per_token else 0) | (2 if fast else 0), float(quant_scale),
                                                  xq.data_ptr(), _ptr(s_row), M, K, _stream(gate)), "asq_silu_mul_quantize")
    return (xq, s_row, row_off) if offsets else (xq, s_row)


def linear_w8a8(xq, w, out_dtype, s_scalar=1.0, s_row=None, s_col=None, bias=None, order="scale_first", out=None):
    """Fused GEMM + dequant/bias epilogue: out[M,N] = (s_col|s_scalar)[*s_row] * f32(xq.w^T) + bias."""
    _dev(xq, "xq"), _dev(w, "weight")
    if xq.dtype != torch.int8 or w.dtype != torch.int8 or xq.dim() != 2 or w.dim() != 2 or xq.shape[1] != w.shape[1]:
        raise ValueError("xq [M,K] and weight [N,K] must be int8 with equal K")
    M, K = xq.shape
    N = w.shape[0]
    for name, t, n in (("s_row", s_row, M), ("s_col", s_col, N), ("bias", bias, N)):
        if t is not None:
            _dev(t, name)
            if t.dtype != torch.float32 or t.numel() != n:
                raise ValueError(f"{name} must be float32 with {n} elements")
    if out is None:
        out = torch.empty((M, N), dtype=out_dtype, device=xq.device)
    else:
        _dev(out, "out")
        if out.dtype != out_dtype or tuple(out.shape) != (M, N):
            raise ValueError("out has wrong dtype/shape")
        _bump_version(out)
    dev = _same_device(xq, w, out, s_row, s_col, bias)
    with _on(dev):
        ws, n = _gemm_ws(M, N, K, dev, _stream(xq))
        L.check(L.lib().asq_linear_w8a8(xq.data_ptr(), w.data_ptr(), out.data_ptr(), _DT[out_dtype], M, N, K, float(s_scalar),
                                        _ptr(s_row), _ptr(s_col), _ptr(bias),
                                        L.ASQ_EPI_SCALE_FIRST if order == "scale_first" else L.ASQ_EPI_ACC_FIRST, _ptr(ws), n, _stream(xq)),
                "asq_linear_w8a8")
    return out


def linear_w8a8_q8(xq, w, mid_dtype, s_scalar=1.0, s_row=None, s_col=None, bias=None, act=None, qmode="per-tensor-round", quant_scale=1.0,
                   order="scale_first"):
    """GEMM + dequant/bias + activation + the NEXT linear's per-tensor quantiser in one launch: returns int8 [M,N], bit-identical
    to linear_w8a8(..., mid_dtype) -> act -> quantize_act(..., qmode, quant_scale).  act in {None, "relu"}."""
    _dev(xq, "xq"), _dev(w, "weight")
    if xq.dtype != torch.int8 or w.dtype != torch.int8 or xq.dim() != 2 or w.dim() != 2 or xq.shape[1] != w.shape[1]:
        raise ValueError("xq [M,K] and weight [N,K] must be int8 with equal K")
    if qmode not in ("per-tensor-round", "per-tensor-div") or act not in (None, "relu") or mid_dtype not in _DT:
        raise ValueError("qmode must be per-tensor-round / per-tensor-div, act None / 'relu', mid_dtype a float dtype")
    M, K = xq.shape
    N = w.shape[0]
    for name, t, n in (("s_row", s_row, M), ("s_col", s_col, N), ("bias", bias, N)):
        if t is not None:
            _dev(t, name)
            if t.dtype != torch.float32 or t.numel() != n:
                raise ValueError(f"{name} must be float32 with {n} elements")
    out = torch.empty((M, N), dtype=torch.int8, device=xq.device)
    dev = _same_device(xq, w, s_row, s_col, bias)
    with _on(dev):
        ws, n = _gemm_ws(M, N, K, dev, _stream(xq))
        L.check(L.lib().asq_linear_w8a8_q8(xq.data_ptr(), w.data_ptr(), out.data_ptr(), _DT[mid_dtype], M, N, K, float(s_scalar), _ptr(s_row), _ptr(s_col),
                                           _ptr(bias), L.ASQ_EPI_SCALE_FIRST if order == "scale_first" else L.ASQ_EPI_ACC_FIRST, 1 if act == "relu" else 0,
                                           _ACT[qmode], float(quant_scale), _ptr(ws), n, _stream(xq)), "asq_linear_w8a8_q8")
    return out


def linear_w8a8_grouped(xq, w, group_offsets, s_group, out_dtype, s_row=None, bias=None):
    """ngroups independent W8A8 linears in one launch (Mixtral experts).  xq int8 [M,K] rows sorted by group,
    w int8 [G,N,K], group_offsets int32 [G+1] on the device, s_group f32 [G] on the device (each expert's
    dequant_scale), s_row f32 [M] (per-token) or None, bias f32 [G,N] or None.  Row m of the result is
    bit-identical to linear_w8a8 on its group's slice."""
    _dev(xq, "xq"), _dev(w, "weight"), _dev(group_offsets, "group_offsets"), _dev(s_group, "s_group")
    if xq.dtype != torch.int8 or w.dtype != torch.int8 or xq.dim() != 2 or w.dim() != 3 or xq.shape[1] != w.shape[2]:
        raise ValueError("xq [M,K] int8 and weight [G,N,K] int8 with equal K expected")
    M, K = xq.shape
    G, N = w.shape[0], w.shape[1]
    if group_offsets.dtype != torch.int32 or group_offsets.numel() != G + 1 or s_group.dtype != torch.float32 or s_group.numel() != G:
        raise ValueError("group_offsets must be int32 [G+1] and s_group float32 [G]")
    for name, t, n in (("s_row", s_row, M), ("bias", bias, G * N)):
        if t is not None:
            _dev(t, name)
            if t.dtype != torch.float32 or t.numel() != n:
                raise ValueError(f"{name} must be float32 with {n} elements")
    out = torch.empty((M, N), dtype=out_dtype, device=xq.device)
    dev = _same_device(xq, w, group_offsets, s_group, s_row, bias)
    with _on(dev):
        lib, st = L.lib(), _stream(xq)
        ws, n = _grouped_ws(lib, M, N, K, G, dev, st)
        L.check(lib.asq_linear_w8a8_grouped_ws(xq.data_ptr(), w.data_ptr(), out.data_ptr(), _DT[out_dtype], group_offsets.data_ptr(), G, M, N, K,
                                               s_group.data_ptr(), _ptr(s_row), _ptr(bias), _ptr(ws), n, st), "asq_linear_w8a8_grouped")
    return out


def grouped_offsets_supported(M, N, K, out_dtype):
    """grouped launches always run on the 256 x 256 kernel; images pay from a few thousand routed rows on (the matrix cores' energy is the limit there)"""
    return (out_dtype in (torch.float16, torch.bfloat16) and M >= 2048 and K % 128 == 0 and 128 <= K <= 65536 and N % 4 == 0
            and bool(L.lib().asq_offsets_supported(4096, 4096, 4096, _DT[out_dtype])))   # (the process-wide ASQ_OFFSETS switch)


def linear_w8a8_grouped_off(xq_off, w_off, row_off, col_off, group_offsets, s_group, out_dtype, s_row=None, bias=None):
    """linear_w8a8_grouped on offset operand images: xq_off / row_off from a *_off quantiser, w_off [G,N,K] / col_off [G,N,2] from weight_offset_image on the
    stack viewed as [G*N, K].  Bit-identical to linear_w8a8_grouped on the plain operands."""
    _dev(xq_off, "xq"), _dev(w_off, "weight"), _dev(group_offsets, "group_offsets"), _dev(s_group, "s_group"), _dev(row_off, "row_off"), _dev(col_off, "col_off")
    if xq_off.dtype != torch.int8 or w_off.dtype != torch.int8 or xq_off.dim() != 2 or w_off.dim() != 3 or xq_off.shape[1] != w_off.shape[2]:
        raise ValueError("xq [M,K] int8 and weight [G,N,K] int8 with equal K expected")
    M, K = xq_off.shape
    G, N = w_off.shape[0], w_off.shape[1]
    if group_offsets.dtype != torch.int32 or group_offsets.numel() != G + 1 or s_group.dtype != torch.float32 or s_group.numel() != G:
        raise ValueError("group_offsets must be int32 [G+1] and s_group float32 [G]")
    if row_off.dtype != torch.int32 or row_off.numel() != 2 * M or col_off.dtype != torch.int32 or col_off.numel() != 2 * G * N:
        raise ValueError("row_off must be int32 [M,2] and col_off int32 [G,N,2]")
    for name, t, n in (("s_row", s_row, M), ("bias", bias, G * N)):
        if t is not None:
            _dev(t, name)
            if t.dtype != torch.float32 or t.numel() != n:
                raise ValueError(f"{name} must be float32 with {n} elements")
    out = torch.empty((M, N), dtype=out_dtype, device=xq_off.device)
    dev = _same_device(xq_off, w_off, group_offsets, s_group, s_row, bias, row_off, col_off)
    with _on(dev):
        lib, st = L.lib(), _stream(xq_off)
        ws, n = _grouped_ws(lib, M, N, K, G, dev, st)
        L.check(lib.asq_linear_w8a8_grouped_off(xq_off.data_ptr(), w_off.data_ptr(), out.data_ptr(), _DT[out_dtype], group_offsets.data_ptr(), G, M, N, K,
                                                s_group.data_ptr(), _ptr(s_row), _ptr(bias), row_off.data_ptr(), col_off.data_ptr(), _ptr(ws), n, st), "asq_linear_w8a8_grouped_off")
    return out


def linear_w8a8_forward(x2d, w, act_mode, quant_scale, s_scalar, s_col=None, bias=None, image=None):
    """Whole module forward on a 2-D activation: quantise -> GEMM + epilogue, one stream,
    no int32 round trip.  Returns out [M,N] in x's dtype.  image = (w_off, col_off) of this weight (weight_offset_image) or None: with it the C-ABI
    runs the shapes it gives to the 256 x 256 kernel on offset operand images (same bits, less energy)."""
    _dev(x2d, "x"), _dev(w, "weight")
    if x2d.dtype not in _DT:
        raise ValueError(f"unsupported activation dtype {x2d.dtype}")
    if w.dtype != torch.int8 or x2d.dim() != 2 or w.dim() != 2 or x2d.shape[1] != w.shape[1]:
        raise ValueError(f"shape/dtype mismatch: x {tuple(x2d.shape)} {x2d.dtype}, weight {tuple(w.shape)} {w.dtype}")
    M, K = x2d.shape
    N = w.shape[0]
    for name, t in (("s_col", s_col), ("bias", bias)):
        if t is not None:
            _dev(t, name)
            if t.dtype != torch.float32 or t.numel() != N:
                raise ValueError(f"{name} must be float32 with {N} elements")
    dev = _same_device(x2d, w, s_col, bias)
    if image is not None:
        _check_image(image, N, K, dev)
    out = torch.empty((M, N), dtype=x2d.dtype, device=dev)
    if M == 0 or N == 0:
        return out
    lib = L.lib()
    stream = _stream(x2d)
    ws, nbytes = _forward_ws(lib, M, N, K, dev, stream)
    with _on(dev):
        if image is not None:
            L.check(lib.asq_linear_w8a8_forward_off(x2d.data_ptr(), _DT[x2d.dtype], w.data_ptr(), image[0].data_ptr(), image[1].data_ptr(), out.data_ptr(), M, N, K,
                                                    _ACT[act_mode], float(quant_scale), float(s_scalar), _ptr(s_col), _ptr(bias),
                                                    ws.data_ptr(), nbytes, stream), "asq_linear_w8a8_forward_off")
        else:
            L.check(lib.asq_linear_w8a8_forward(x2d.data_ptr(), _DT[x2d.dtype], w.data_ptr(), out.data_ptr(), M, N, K,
                                                _ACT[act_mode], float(quant_scale), float(s_scalar), _ptr(s_col), _ptr(bias),
                                                ws.data_ptr(), nbytes, stream), "asq_linear_w8a8_forward")
    return out


def linear_w8a8_forward_trusted(x2d, w, act_code, quant_scale, s_scalar, s_col, bias, image, N, K):
    """linear_w8a8_forward without the per-call validation of the module's own state, for callers that check it where it can change (the nn modules:
    weight / bias / scale vector are validated once per tensor object, layers/nn/linear.py::_module_forward).  Still checked here, per call: the activation
    (HIP tensor of a supported dtype; contiguity and K are the caller's `_flatten`; `image` must come from the module's own cache, whose buffers
    weight_offset_image allocated or validated).  At decode sizes the module call is host-bound (12.9 us at 32 x 4096 x 4096,
    5.6 us of it the two launches): this path takes ~2 us of Python out of it (tools/pyoverhead.py).  act_code: _lib.ASQ_ACT_*; quant_scale, s_scalar: Python floats."""
    dt = _DT.get(x2d.dtype)
    if dt is None or not x2d.is_cuda:
        raise (ValueError(f"unsupported activation dtype {x2d.dtype}") if x2d.is_cuda else
               RuntimeError(f"x is on {x2d.device}: autosmoothquant_amd ops need a HIP (cuda:N) tensor; there is no CPU fallback"))
    dev = x2d.device
    if w.device != dev:
        raise RuntimeError(f"tensors on different devices: {dev} vs {w.device}")
    M = x2d.shape[0]
    out = torch.empty((M, N), dtype=x2d.dtype, device=dev)
    if M == 0 or N == 0:
        return out
    lib = L.lib()
    stream = _stream(x2d)
    ws, nbytes = _forward_ws(lib, M, N, K, dev, stream)
    with _on(dev):
        if image is not None:
            rc = lib.asq_linear_w8a8_forward_off(x2d.data_ptr(), dt, w.data_ptr(), image[0].data_ptr(), image[1].data_ptr(), out.data_ptr(), M, N, K, act_code, quant_scale, s_scalar,
                                                 None if s_col is None else s_col.data_ptr(), None if bias is None else bias.data_ptr(), ws.data_ptr(), nbytes, stream)
        else:
            rc = lib.asq_linear_w8a8_forward(x2d.data_ptr(), dt, w.data_ptr(), out.data_ptr(), M, N, K, act_code, quant_scale, s_scalar,
                                             None if s_col is None else s_col.data_ptr(), None if bias is None else bias.data_ptr(), ws.data_ptr(), nbytes, stream)
    if rc:
        L.check(rc, "asq_linear_w8a8_forward")
    return out


_FP8 = {"per-token": L.ASQ_FP8_PER_TOKEN, "per-tensor": L.ASQ_FP8_PER_TENSOR, "static": L.ASQ_FP8_STATIC}


def quantize_act_fp8(x, mode, static_scale=1.0):
    """x [M,K] f32/f16/bf16 -> (xq float8_e4m3fn [M,K], scale).  mode "per-token": scale f32 [M,1] on the
    device; "per-tensor" (dynamic): scale f32 0-dim ON THE DEVICE (no host sync, like the reference's
    0-dim scale tensor); "static": scale is the given host value (returned as a python float).
    Reference: layers/functional/quantization.py:144-211."""
    _dev(x, "x")
    if x.dtype not in _DT or x.dim() != 2:
        raise ValueError("x must be a 2-D float32/float16/bfloat16 tensor")
    M, K = x.shape
    xq = torch.empty((M, K), dtype=torch.uint8, device=x.device)
    if mode == "per-token":
        sc = torch.empty((M,), dtype=torch.float32, device=x.device)
    elif mode == "per-tensor":
        sc = torch.empty((2,), dtype=torch.float32, device=x.device)
    else:
        sc = None
    with _on(x.device):
        L.check(L.lib().asq_quantize_act_fp8(x.data_ptr(), _DT[x.dtype], _FP8[mode], float(static_scale), xq.data_ptr(), _ptr(sc), M, K,
                                             _stream(x)), "asq_quantize_act_fp8")
    xq = xq.view(torch.float8_e4m3fn)
    if mode == "per-token":
        return xq, sc.view(M, 1)
    if mode == "per-tensor":
        return xq, sc[0]
    return xq, float(static_scale)


def rmsnorm(x, weight, eps=1e-5):
    """y = weight * dt(f32(x) * rsqrt(mean(x^2) + eps)) in ONE pass (asq_rmsnorm): HF LlamaRMSNorm's arithmetic with a floating output -- the norm as a module of its own, as
    the reference's composition has it (models/llama.py:27-37); the N1 form norm_quantize emits int8 instead.  x [M,K], weight [K] of x's dtype."""
    _dev(x, "x"), _dev(weight, "weight")
    if x.dtype not in _DT or x.dim() != 2 or weight.dtype != x.dtype or weight.numel() != x.shape[1]:
        raise ValueError("x must be 2-D float and weight a [K] tensor of the same dtype")
    M, K = x.shape
    y = torch.empty_like(x)
    with _on(x.device):
        L.check(L.lib().asq_rmsnorm(x.data_ptr(), _DT[x.dtype], weight.data_ptr(), float(eps), y.data_ptr(), M, K, _stream(x)), "asq_rmsnorm")
    return y


def silu_mul(gate, up, fast=None):
    """silu(gate) * up in the activation dtype in ONE pass (asq_silu_mul): dt(dt(silu(g)) * u), the two roundings of F.silu(gate) * up -- the gated activation as an op of
    its own, as the reference's composition has it (HF LlamaMLP); the N1 form silu_mul_quantize emits int8 instead.  gate / up: contiguous tensors of one shape and dtype."""
    if fast is None:
        fast = not SILU_EXACT_DEFAULT
    _dev(gate, "gate"), _dev(up, "up")
    if gate.dtype not in _DT or up.dtype != gate.dtype or up.shape != gate.shape:
        raise ValueError("gate and up must be float tensors of equal shape and dtype")
    out = torch.empty_like(gate)
    with _on(gate.device):
        L.check(L.lib().asq_silu_mul(gate.data_ptr(), up.data_ptr(), _DT[gate.dtype], L.ASQ_SILU_FAST if fast else 0, out.data_ptr(), gate.numel(), _stream(gate)), "asq_silu_mul")
    return out


def rope(x, cos, sin, out=None):
    """Rotary embedding of x [B, S, H, D] in one pass (asq_rope): x is a q / k projection's output viewed per head -- dense, or a slice of a fused q || k || v output
    (strides (S * ld, ld, D, 1) with ld >= H * D); cos / sin [S, D/2] of x's dtype, positions 0 .. S-1, rotate_half convention.  Returns a DENSE [B, S, H, D] tensor
    (out=x rotates a dense x in place).  fp16: bit-identical to the torch composition addcmul(x1 * cos, x2, sin, value=-1) / addcmul(x2 * cos, x1, sin) it replaces
    (harness._rope_torch); bf16: the same operations at fp32 width."""
    _dev(cos, "cos"), _dev(sin, "sin")
    if not isinstance(x, torch.Tensor) or not x.is_cuda:   # (x may be a strided slice: _dev's contiguity check does not apply)
        raise RuntimeError("x must be a HIP (cuda:N) tensor; there is no CPU fallback")
    if x.dtype not in _DT or x.dim() != 4:
        raise ValueError("x must be a [B, S, H, D] float tensor")
    B, S, H, D = x.shape
    st = x.stride()
    ld = st[1] if S > 1 else (st[0] if B > 1 else H * D)
    if not (st[3] == 1 and st[2] == D and ld >= H * D and (B == 1 or S == 1 or st[0] == S * ld)) or (S == 1 and B > 1 and st[0] < H * D):
        raise ValueError("x must be [B, S, H, D] with H * D contiguous and one row pitch (a projection output or a slice of a fused one)")
    if cos.dtype != x.dtype or sin.dtype != x.dtype or cos.numel() != S * (D // 2) or sin.numel() != S * (D // 2) or not (cos.is_contiguous() and sin.is_contiguous()):
        raise ValueError("cos / sin must be contiguous [S, D/2] tables of x's dtype")
    if out is None:
        out = torch.empty((B, S, H, D), dtype=x.dtype, device=x.device)
    elif out.shape != x.shape or out.dtype != x.dtype or not out.is_contiguous() or out.device != x.device:
        raise ValueError("out must be a dense tensor of x's shape, dtype and device")
    with _on(x.device):
        L.check(L.lib().asq_rope(x.data_ptr(), ld, out.data_ptr(), _DT[x.dtype], cos.data_ptr(), sin.data_ptr(), B, S, H, D, _stream(x)), "asq_rope")
    return out


def silu_mul_quantize_fp8(gate, up, fast=None):
    """e4m3(per-token quantise(silu(gate) * up)) in ONE pass (asq_silu_mul_quantize_fp8): (xq float8_e4m3fn [M,K], scale f32 [M,1]) -- what
    quantize_act_fp8(F.silu(gate) * up, "per-token") returns from three.  fast as in silu_mul_quantize (default: the hardware-transcendental SiLU; False: the
    fixed-operation-order form oracle/n1.py::silu_mul_quant_fp8_kernel_order repeats bit for bit).  Reference: models/mixtral.py:99-101 on FP8LinearDynamic modules."""
    if fast is None:
        fast = not SILU_EXACT_DEFAULT
    _dev(gate, "gate"), _dev(up, "up")
    if gate.dtype not in _DT or gate.dim() != 2 or up.dtype != gate.dtype or up.shape != gate.shape:
        raise ValueError("gate and up must be 2-D float tensors of equal shape and dtype")
    if not (gate.is_contiguous() and up.is_contiguous()):
        raise ValueError("gate / up must be contiguous")
    M, K = gate.shape
    xq = torch.empty((M, K), dtype=torch.uint8, device=gate.device)
    sc = torch.empty((M,), dtype=torch.float32, device=gate.device)
    with _on(gate.device):
        L.check(L.lib().asq_silu_mul_quantize_fp8(gate.data_ptr(), up.data_ptr(), _DT[gate.dtype], L.ASQ_SILU_FAST if fast else 0, xq.data_ptr(), sc.data_ptr(), M, K,
                                                  _stream(gate)), "asq_silu_mul_quantize_fp8")
    return xq.view(torch.float8_e4m3fn), sc.view(M, 1)


def fp8_grouped_gate_up_supported(M, F, K, out_dtype):
    return out_dtype in _DT and bool(L.lib().asq_fp8_grouped_gate_up_supported(int(M), int(F), int(K), _DT[out_dtype]))


def linear_fp8_grouped_gate_up(xq, a_scale, w_gu, group_offsets, s_gate, s_up, out_dtype, fast=None):
    """SiLU(w1 x) * (w3 x) of ALL groups in ONE grouped fp8 launch over the interleaved stacks w_gu [G, 2F, K] (interleave_gate_up_stack of two float8_e4m3fn stacks: blocks
    of 32 channels; asq_linear_fp8_grouped_gate_up): out [M, F], bit-identical to linear_fp8_grouped (w1), (w3) and the SiLU * up of silu_mul_quantize_fp8(fast=...).
    xq float8_e4m3fn [M, K] + a_scale f32 [M] or [M, 1] (quantize_act_fp8 per-token); s_gate / s_up f32 [G] on the device."""
    if fast is None:
        fast = not SILU_EXACT_DEFAULT
    _dev(xq, "xq"), _dev(w_gu, "w_gu"), _dev(group_offsets, "group_offsets"), _dev(s_gate, "s_gate"), _dev(s_up, "s_up"), _dev(a_scale, "a_scale")
    if xq.dtype != torch.float8_e4m3fn or w_gu.dtype != torch.float8_e4m3fn or xq.dim() != 2 or w_gu.dim() != 3 or xq.shape[1] != w_gu.shape[2] or w_gu.shape[1] % 2:
        raise ValueError("xq [M,K] and w_gu [G,2F,K] must be float8_e4m3fn with equal K")
    M, K = xq.shape
    G, F_ = w_gu.shape[0], w_gu.shape[1] // 2
    if group_offsets.dtype != torch.int32 or group_offsets.numel() != G + 1:
        raise ValueError("group_offsets must be int32 with G + 1 elements")
    for name, t, n in (("a_scale", a_scale, M), ("s_gate", s_gate, G), ("s_up", s_up, G)):
        if t.dtype != torch.float32 or t.numel() != n:
            raise ValueError(f"{name} must be float32 with {n} elements")
    if not fp8_grouped_gate_up_supported(M, F_, K, out_dtype):
        raise ValueError("shape not supported by the grouped fp8 gate || up launch (fp8_grouped_gate_up_supported)")
    out = torch.empty((M, F_), dtype=out_dtype, device=xq.device)
    with _on(xq.device):
        L.check(L.lib().asq_linear_fp8_grouped_gate_up(xq.data_ptr(), w_gu.data_ptr(), out.data_ptr(), _DT[out_dtype], group_offsets.data_ptr(), G, M, F_, K, a_scale.data_ptr(),
                                                       s_gate.data_ptr(), s_up.data_ptr(), L.ASQ_SILU_FAST if fast else 0, _stream(xq)), "asq_linear_fp8_grouped_gate_up")
    return out


def quantize_mxfp8(x):
    """MX (OCP Microscaling) e4m3 with real block scales -- opt-in extension, see include/asq_hip.h.  x [M,K] f32/f16/bf16, K % 32 == 0
    -> (codes float8_e4m3fn [M,K], scales uint8 [M,K/32] as E8M0 bytes)."""
    _dev(x, "x")
    if x.dtype not in _DT or x.dim() != 2 or x.shape[1] % 32 != 0:
        raise ValueError("x must be a 2-D float32/float16/bfloat16 tensor whose last dim is a multiple of 32")
    M, K = x.shape
    xq = torch.empty((M, K), dtype=torch.uint8, device=x.device)
    sc = torch.empty((M, K // 32), dtype=torch.uint8, device=x.device)
    with _on(x.device):
        L.check(L.lib().asq_quantize_mxfp8(x.data_ptr(), _DT[x.dtype], xq.data_ptr(), sc.data_ptr(), M, K, _stream(x)), "asq_quantize_mxfp8")
    return xq.view(torch.float8_e4m3fn), sc


def linear_mxfp8(xq, x_scales, wq, w_scales, out_dtype, bias=None):
    """out[M,N] = block-scaled e4m3 product of (xq, x_scales) [M,K] and (wq, w_scales) [N,K] (+ bias) on the scaled matrix-core instruction."""
    for name, t in (("xq", xq), ("x_scales", x_scales), ("wq", wq), ("w_scales", w_scales)):
        _dev(t, name)
    if xq.dim() != 2 or wq.dim() != 2 or xq.shape[1] != wq.shape[1] or xq.shape[1] % 64 != 0 or xq.element_size() != 1 or wq.element_size() != 1:
        raise ValueError("xq [M,K] and wq [N,K] must be 1-byte e4m3 tensors with equal K, K % 64 == 0")
    M, K = xq.shape
    N = wq.shape[0]
    if tuple(x_scales.shape) != (M, K // 32) or tuple(w_scales.shape) != (N, K // 32) or x_scales.dtype != torch.uint8 or w_scales.dtype != torch.uint8:
        raise ValueError("scales must be uint8 [rows, K/32]")
    if bias is not None:
        _dev(bias, "bias")
        if bias.dtype != torch.float32 or bias.numel() != N:
            raise ValueError(f"bias must be float32 with {N} elements")
    out = torch.empty((M, N), dtype=out_dtype, device=xq.device)
    if M == 0 or N == 0:
        return out
    dev = _same_device(xq, x_scales, wq, w_scales, bias)
    with _on(dev):
        L.check(L.lib().asq_linear_mxfp8(xq.data_ptr(), x_scales.data_ptr(), wq.data_ptr(), w_scales.data_ptr(), out.data_ptr(), _DT[out_dtype], M, N, K,
                                         _ptr(bias), _stream(xq)), "asq_linear_mxfp8")
    return out


def cast_e5m2(x):
    """x [M,K] f32/f16/bf16 -> float8_e5m2 [M,K], plain round-to-nearest-even cast (FP8E5M2Linear, linear.py:612)."""
    _dev(x, "x")
    if x.dtype not in _DT or x.dim() != 2:
        raise ValueError("x must be a 2-D float32/float16/bfloat16 tensor")
    M, K = x.shape
    xq = torch.empty((M, K), dtype=torch.uint8, device=x.device)
    with _on(x.device):
        L.check(L.lib().asq_cast_e5m2(x.data_ptr(), _DT[x.dtype], xq.data_ptr(), M * K, _stream(x)), "asq_cast_e5m2")
    return xq.view(torch.float8_e5m2)


def linear_fp8(xq, a_scale, w, w_scale, bias, out_dtype):
    """easy_fp8_gemm on the fp8 matrix cores (reference linear.py:336-369): out = (xq . w^T) * a_scale * w_scale (+ bias).
    a_scale: device f32 tensor ([M,1] per-token or 0-dim per-tensor) or a python float."""
    _dev(xq, "xq"), _dev(w, "weight")
    f8 = (torch.float8_e4m3fn, torch.float8_e5m2)
    if xq.dtype not in f8 or w.dtype != xq.dtype or xq.dim() != 2 or w.dim() != 2 or xq.shape[1] != w.shape[1]:
        raise ValueError("xq [M,K] and weight [N,K] must both be float8_e4m3fn (or both float8_e5m2) with equal K")
    fmt = 0 if xq.dtype == torch.float8_e4m3fn else 1
    M, K = xq.shape
    N = w.shape[0]
    out = torch.empty((M, N), dtype=out_dtype, device=xq.device)
    if M == 0 or N == 0:
        return out
    a_dev, per_token, a_host = None, 0, 1.0
    if isinstance(a_scale, torch.Tensor):
        if a_scale.is_cuda:
            a_dev = a_scale.to(torch.float32).contiguous()
            per_token = 1 if (a_scale.dim() >= 1 and a_scale.numel() == M) else 0
            if not per_token and a_dev.numel() != 1:
                raise ValueError("a_scale must have 1 or M elements")
        else:
            a_host = float(a_scale)
    else:
        a_host = float(a_scale)
    if bias is not None:
        _dev(bias, "bias")
        if bias.dtype != torch.float32 or bias.numel() != N:
            raise ValueError(f"bias must be float32 with {N} elements")
    dev = _same_device(xq, w, a_dev, bias)
    with _on(dev):
        L.check(L.lib().asq_linear_fp8(xq.data_ptr(), w.data_ptr(), fmt, out.data_ptr(), _DT[out_dtype], M, N, K, _ptr(a_dev), per_token,
                                       a_host, float(w_scale), _ptr(bias), _stream(xq)), "asq_linear_fp8")
    return out


def gemm_kernel_name(M, N, K):
    return L.lib().asq_gemm_kernel_name(M, N, K).decode()


def linear_fp8_grouped(xq, a_scale, w, w_scale_group, group_offsets, out_dtype, bias=None):
    """ngroups independent e4m3 linears in one launch (Mixtral experts, FP8LinearDynamic math).  xq float8_e4m3fn [M,K]
    rows sorted by group, a_scale f32 [M] or [M,1] (per-token, device), w float8_e4m3fn [G,N,K], w_scale_group f32 [G]
    (device), group_offsets int32 [G+1] (device), bias f32 [G,N] or None."""
    _dev(xq, "xq"), _dev(w, "weight"), _dev(group_offsets, "group_offsets"), _dev(w_scale_group, "w_scale_group"), _dev(a_scale, "a_scale")
    f8 = torch.float8_e4m3fn
    if xq.dtype != f8 or w.dtype != f8 or xq.dim() != 2 or w.dim() != 3 or xq.shape[1] != w.shape[2]:
        raise ValueError("xq [M,K] and weight [G,N,K] must be float8_e4m3fn with equal K")
    M, K = xq.shape
    G, N = w.shape[0], w.shape[1]
    if group_offsets.dtype != torch.int32 or group_offsets.numel() != G + 1 or w_scale_group.dtype != torch.float32 or w_scale_group.numel() != G:
        raise ValueError("group_offsets must be int32 [G+1] and w_scale_group float32 [G]")
    if a_scale.dtype != torch.float32 or a_scale.numel() != M:
        raise ValueError("a_scale must be float32 with M elements (per-token)")
    if bias is not None:
        _dev(bias, "bias")
        if bias.dtype != torch.float32 or bias.numel() != G * N:
            raise ValueError(f"bias must be float32 with {G * N} elements")
    out = torch.empty((M, N), dtype=out_dtype, device=xq.device)
    if M == 0 or N == 0:
        return out
    dev = _same_device(xq, w, group_offsets, w_scale_group, a_scale, bias)
    with _on(dev):
        L.check(L.lib().asq_linear_fp8_grouped(xq.data_ptr(), w.data_ptr(), out.data_ptr(), _DT[out_dtype], group_offsets.data_ptr(), G, M, N, K,
                                               a_scale.data_ptr(), w_scale_group.data_ptr(), _ptr(bias), _stream(xq)), "asq_linear_fp8_grouped")
    return out
