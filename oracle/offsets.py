"""TEST INFRASTRUCTURE (oracle): NumPy restatement of the offset operand images of include/asq_hip.h ("offset operand images").

There is no reference counterpart -- the reference hands the raw int8 operands to cuBLASLt (csrc/int8gemm/cublasINT8MMWrapper.cc:224-354) -- so this
file pins only what the images must satisfy: they are derived from the plain operands by the two stated rules, and the plain int32 product
(oracle/w8a8.py::gemm_i8_i32, the reference's arithmetic) is recovered from them exactly.  Only tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg may import this package."""
import numpy as np

CX = 3     # activation row offset magnitude (ASQ_OFF_CX)
CW = 64    # cap of the weight row offset (ASQ_OFF_CW)


def weight_image(w, cw_cap=CW):
    """w int8 [N,K] -> (w_off int8 [N,K], col_off int32 [N,2] = {cw[n], sum_k w[n,k] of the ORIGINAL row}); cw[n] = min(cap, 127 - max_k w[n,k])."""
    w = np.asarray(w, dtype=np.int8)
    cw = np.minimum(cw_cap, 127 - w.max(axis=1).astype(np.int32)).astype(np.int32)
    w_off = (w.astype(np.int32) + cw[:, None])
    assert w_off.min() >= -128 and w_off.max() <= 127
    return w_off.astype(np.int8), np.stack([cw, w.astype(np.int64).sum(axis=1).astype(np.int32)], axis=1).astype(np.int32)


def act_offsets(xq, c=CX):
    """cx[m] = +c if max_k xq <= 127 - c, else -c if min_k xq >= -128 + c, else 0."""
    xq = np.asarray(xq, dtype=np.int8).astype(np.int32)
    rmax, rmin = xq.max(axis=1), xq.min(axis=1)
    return np.where(rmax <= 127 - c, c, np.where(rmin >= -128 + c, -c, 0)).astype(np.int32)


def act_image(xq, c=CX):
    """xq int8 [M,K] -> (x_off int8 [M,K], row_off int32 [M,2] = {cx[m], sum_k x_off[m,k]})."""
    xq = np.asarray(xq, dtype=np.int8)
    cx = act_offsets(xq, c)
    x_off = xq.astype(np.int32) + cx[:, None]
    assert x_off.min() >= -128 and x_off.max() <= 127
    return x_off.astype(np.int8), np.stack([cx, x_off.sum(axis=1).astype(np.int32)], axis=1).astype(np.int32)


def product_from_images(x_off, row_off, w_off, col_off):
    """The int32 product of the PLAIN operands recovered from the images, in two's-complement int32 arithmetic as the kernel forms it:
    accumulators start at -(cx[m] * wsum[n] + cw[n] * xsum'[m]) and add sum_k x_off[m,k] w_off[n,k]."""
    x_off, w_off = np.asarray(x_off, dtype=np.int8), np.asarray(w_off, dtype=np.int8)
    cx, xs = row_off[:, 0].astype(np.int64), row_off[:, 1].astype(np.int64)
    cw, ws = col_off[:, 0].astype(np.int64), col_off[:, 1].astype(np.int64)
    start = -(cx[:, None] * ws[None, :] + cw[None, :] * xs[:, None])
    acc = start + x_off.astype(np.int64) @ w_off.astype(np.int64).T
    return ((acc + 2 ** 31) % 2 ** 32 - 2 ** 31).astype(np.int32)   # wrap-around, as the hardware
