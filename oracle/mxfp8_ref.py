"""TEST ORACLE (not product code) for BASELINE.json configs[4] (fp8 FFN GEMMs): numpy restatement of the OCP
microscaling FP8 format the HIP path uses (csrc/common.h mx_e8m0 / pack_fp8x4, csrc/gemm_mxfp8.hip).

There is no fp8 path in the reference (sylber/model/sylber.py runs HubertModel in fp32); the config asks for a
"tolerance vs the bf16 reference", so this oracle pins the QUANTISER and the block-scaled contraction exactly and the
end-to-end tolerance is stated in tests/test_gpu_fp8.py.

Format: e4m3 (OCP "e4m3fn": bias 7, max 448, no infinities) elements, one E8M0 scale 2^(s-127) per 32 consecutive
elements along K.  Scale rule: the smallest power of two 2^e with amax <= 448 * 2^e (so nothing saturates), an all-zero
block gets 2^0; elements are x / 2^e rounded to nearest, ties to even."""
from __future__ import annotations

import numpy as np


def _e4m3_table():
    codes = np.arange(127, dtype=np.uint8)                       # 0x00 .. 0x7E: finite, non-negative, monotonic
    e = (codes >> 3).astype(np.int32)
    m = (codes & 7).astype(np.float64)
    vals = np.where(e == 0, m / 8.0 * 2.0 ** -6, (1.0 + m / 8.0) * 2.0 ** (e - 7.0))
    return codes, vals


_CODES, _VALS = _e4m3_table()


def e4m3_decode(b: np.ndarray) -> np.ndarray:
    b = np.asarray(b, dtype=np.uint8)
    mag = _VALS[np.minimum(b & 0x7F, 126)]
    return np.where(b & 0x80, -mag, mag)


def e4m3_encode(x: np.ndarray) -> np.ndarray:
    """round to nearest even onto the e4m3 grid; |x| must be <= 448 (guaranteed by the scale rule)"""
    x = np.asarray(x, dtype=np.float64)
    a = np.abs(x)
    hi = np.clip(np.searchsorted(_VALS, a, side="left"), 0, 126)
    lo = np.clip(hi - 1, 0, 126)
    dlo, dhi = a - _VALS[lo], _VALS[hi] - a
    pick_hi = (dhi < dlo) | ((dhi == dlo) & ((_CODES[hi] & 1) == 0))
    code = np.where(pick_hi, _CODES[hi], _CODES[lo]).astype(np.uint8)
    code = np.where(a >= 448.0, np.uint8(126), code)
    return code | np.where(np.signbit(x), np.uint8(0x80), np.uint8(0)).astype(np.uint8)


def block_scale(amax: np.ndarray) -> np.ndarray:
    """biased E8M0 byte of the smallest 2^e with amax <= 448 * 2^e (float32 bit arithmetic, like the kernel)"""
    u = np.asarray(amax, dtype=np.float32).view(np.uint32).astype(np.int64)
    b = (u >> 23) - 8 + ((u & 0x7FFFFF) > 0x600000)
    b = np.clip(b, 0, 254)
    return np.where(u == 0, 127, b).astype(np.uint8)


def quantize(x: np.ndarray):
    """x float32 [R, K] (K % 32 == 0) -> (e4m3 bytes [R, K], E8M0 bytes [R, K // 32])"""
    x = np.ascontiguousarray(x, dtype=np.float32)
    R, K = x.shape
    blocks = x.reshape(R, K // 32, 32)
    scale = block_scale(np.abs(blocks).max(-1))
    inv = (2.0 ** (127.0 - scale.astype(np.float64)))[..., None]
    scaled = (blocks.astype(np.float64) * inv).astype(np.float32)      # exact: power-of-two scaling (or flush region)
    return e4m3_encode(scaled).reshape(R, K), scale


def dequantize(data: np.ndarray, scale: np.ndarray) -> np.ndarray:
    R, K = data.shape
    v = e4m3_decode(data).reshape(R, K // 32, 32) * (2.0 ** (scale.astype(np.float64) - 127.0))[..., None]
    return v.reshape(R, K)


def linear(a: np.ndarray, w: np.ndarray, bias=None) -> np.ndarray:
    """what the MXFP8 GEMM computes, in float64: dequant(quant(a)) @ dequant(quant(w)).T + bias"""
    y = dequantize(*quantize(a)) @ dequantize(*quantize(w)).T
    return y if bias is None else y + np.asarray(bias, np.float64)[None, :]
