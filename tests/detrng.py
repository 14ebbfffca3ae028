"""Build-owned counter-based RNG (splitmix64 over an index grid).

Generates identical bytes on any machine / NumPy / torch version, so config-sized
golden cases (tests/golden G4) can be regenerated on the GPU box from a seed
instead of being shipped as multi-megabyte tensors."""
import numpy as np

_M64 = np.uint64(0xFFFFFFFFFFFFFFFF)


def _splitmix64(z: np.ndarray) -> np.ndarray:
    with np.errstate(over="ignore"):
        z = (z + np.uint64(0x9E3779B97F4A7C15)) & _M64
        z = ((z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)) & _M64
        z = ((z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)) & _M64
        return z ^ (z >> np.uint64(31))


def u64(seed: int, stream: int, n: int) -> np.ndarray:
    idx = np.arange(n, dtype=np.uint64)
    with np.errstate(over="ignore"):
        base = _splitmix64(np.array([seed * 1000003 + stream], dtype=np.uint64))[0]
        return _splitmix64(idx * np.uint64(0xD1342543DE82EF95) + base)


def int8_uniform(seed: int, stream: int, shape) -> np.ndarray:
    """uniform over [-128, 127]"""
    n = int(np.prod(shape))
    return (u64(seed, stream, n) >> np.uint64(56)).astype(np.uint8).view(np.int8).reshape(shape)


def uniform01(seed: int, stream: int, shape) -> np.ndarray:
    """float64 uniform in [0,1) with 53 random bits"""
    n = int(np.prod(shape))
    return ((u64(seed, stream, n) >> np.uint64(11)).astype(np.float64) / float(1 << 53)).reshape(shape)


def normal(seed: int, stream: int, shape) -> np.ndarray:
    """float32 standard normal via Box-Muller on two uniform streams (float64 math,
    rounded once to float32 -> identical everywhere libm's log/cos/sqrt agree to
    float64 ulp; inputs are then snapped to a coarse grid by `grid` helpers)."""
    u1 = uniform01(seed, 2 * stream, shape)
    u2 = uniform01(seed, 2 * stream + 1, shape)
    z = np.sqrt(-2.0 * np.log(1.0 - u1)) * np.cos(2.0 * np.pi * u2)
    return z.astype(np.float32)


def act_like(seed: int, stream: int, shape, scale=48.0, outlier_frac=0.01, outlier_gain=20.0,
             grid=1.0 / 8) -> np.ndarray:
    """Activation-shaped values, snapped to multiples of `grid` (exact in fp16/bf16
    for the magnitudes used, and immune to last-ulp libm differences): N(0,1)*scale
    with `outlier_frac` of the channels amplified (SmoothQuant-like outliers)."""
    z = normal(seed, stream, shape).astype(np.float64) * scale
    K = shape[-1]
    ch = uniform01(seed, 7919 + stream, (K,)) < outlier_frac
    z[..., ch] *= outlier_gain
    z = np.round(z / grid) * grid
    return z.astype(np.float32)
