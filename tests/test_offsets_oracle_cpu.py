"""CPU: the offset-image identity in oracle/offsets.py against the reference's integer product (oracle/w8a8.py::igemm, the restatement of
cublasINT8MMWrapper.cc:224-354 pinned by the golden fixtures): for every operand pattern the plain int32 product is recovered from the images
exactly, with two's-complement wrap-around where the sums leave int32."""
import numpy as np

from oracle import offsets as OFF, w8a8 as O


def _check(x, w):
    xo, ro = OFF.act_image(x)
    wo, co = OFF.weight_image(w)
    assert np.array_equal(xo.astype(np.int32) - ro[:, :1], x.astype(np.int32))
    assert np.array_equal(wo.astype(np.int32) - co[:, :1], w.astype(np.int32))
    assert np.array_equal(ro[:, 1], xo.astype(np.int64).sum(1)) and np.array_equal(co[:, 1], w.astype(np.int64).sum(1))
    assert np.array_equal(OFF.product_from_images(xo, ro, wo, co), O.igemm(x, w))
    return ro, co


def test_images_recover_the_plain_product():
    rng = np.random.default_rng(3)
    for (M, N, K) in ((1, 4, 16), (37, 20, 256), (130, 64, 1024)):
        x = rng.integers(-128, 128, (M, K), dtype=np.int8)
        w = rng.integers(-128, 128, (N, K), dtype=np.int8)
        _check(x, w)
        xs = np.clip(np.rint(rng.standard_normal((M, K)) * 1.4), -128, 127).astype(np.int8)
        ws = np.clip(np.rint(rng.standard_normal((N, K)) * 22), -128, 127).astype(np.int8)
        ro, co = _check(xs, ws)
        assert (ro[:, 0] == OFF.CX).all() and (co[:, 0] > 20).all()      # SmoothQuant-like rows take the full activation offset and a large weight offset


def test_offset_rules_on_extreme_rows():
    K = 64
    x = np.zeros((6, K), np.int8)
    x[0] = 127; x[1] = -128; x[2, 0] = 127; x[2, 1] = -128; x[3, 0] = 125; x[4, 0] = 124; x[5, 0] = -126
    assert OFF.act_offsets(x).tolist() == [-3, 3, 0, -3, 3, 3]
    w = np.zeros((4, K), np.int8)
    w[0] = 127; w[1] = -128; w[2, 0] = 100
    assert OFF.weight_image(w)[1][:, 0].tolist() == [0, 64, 27, 64]
    _check(x, np.ascontiguousarray(np.resize(w, (8, K))))
    # the int32 edge: 65536 products of (-128) x (-128) = 2^30; images and start values wrap, the result does not
    xe = np.full((2, 65536), -128, np.int8)
    we = np.full((4, 65536), -128, np.int8)
    we[1] = 127
    _check(xe, we)
