"""CPU: oracle/mx.py against hand-derived known answers of the OCP Microscaling spec (MXFP8 E4M3: shared exponent = floor(log2 max) - 8, saturating
round-to-nearest-even elements).  The reference has no block-scaled mode, so these spec-derived vectors are what pins the oracle."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import fp8 as F8
from oracle import mx as MX


def _block(vals):
    x = np.zeros((1, 32), np.float32)
    x[0, :len(vals)] = vals
    return x


def test_known_answers():
    # max = 1.0 -> shared exponent 0 - 8: scale byte 119 (2^-8); 1.0 / 2^-8 = 256 = 1.0 x 2^8 -> e4m3 0b1111_000
    q, s = MX.mx_quantize_e4m3(_block([1.0, -0.5, 0.75]))
    assert s[0, 0] == 119 and q[0, 0] == 0x78 and q[0, 1] == 0xF0 and q[0, 2] == 0x74 and q[0, 3] == 0x00
    # max = 448 (the largest e4m3): floor(log2 448) = 8 -> scale 2^0; 448 -> 0x7E
    q, s = MX.mx_quantize_e4m3(_block([448.0, 1.0]))
    assert s[0, 0] == 127 and q[0, 0] == 0x7E and q[0, 1] == 0x38
    # max = 500: same scale, 500 > 448 saturates (the spec's clamp) instead of becoming NaN
    q, s = MX.mx_quantize_e4m3(_block([500.0, -511.0]))
    assert s[0, 0] == 127 and q[0, 0] == 0x7E and q[0, 1] == 0xFE
    # all-zero block: smallest scale, zero codes; NaN / inf in a block -> NaN scale byte
    q, s = MX.mx_quantize_e4m3(_block([]))
    assert s[0, 0] == 0 and not q.any()
    assert MX.mx_quantize_e4m3(_block([np.inf]))[1][0, 0] == 255 and MX.mx_quantize_e4m3(_block([1.0, np.nan]))[1][0, 0] == 255


def test_round_trip_and_linear():
    rng = np.random.default_rng(3)
    x = (rng.standard_normal((7, 128)) * np.exp(rng.standard_normal((7, 1)) * 3)).astype(np.float32)
    q, s = MX.mx_quantize_e4m3(x)
    d = MX.mx_dequantize(q, s)
    # a block's elements are within half an e4m3 ulp at the block's top binade (2^-3 relative to 2^floor(log2 max)), or saturated at 448 * scale
    xb, db = x.reshape(7, 4, 32).astype(np.float64), d.reshape(7, 4, 32)
    top = 2.0 ** np.floor(np.log2(np.abs(xb).max(axis=-1, keepdims=True)))
    assert np.all(np.abs(db - np.clip(xb, -1.75 * top, 1.75 * top)) <= top / 16 + 1e-30)
    w = rng.standard_normal((5, 128)).astype(np.float32)
    qw, sw = MX.mx_quantize_e4m3(w)
    y = MX.mx_linear(q, s, qw, sw, None, "f32")
    assert np.allclose(y, (d @ MX.mx_dequantize(qw, sw).T).astype(np.float32))
    assert np.array_equal(F8.e4m3fn_to_f32(np.array([0x78, 0x7E, 0x38], np.uint8)), np.array([256.0, 448.0, 1.0], np.float32))
