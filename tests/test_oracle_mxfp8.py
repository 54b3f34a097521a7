"""CPU tier, BASELINE configs[4]: the MXFP8 oracle quantiser itself (format facts of OCP e4m3 / E8M0)."""
import numpy as np

from oracle import mxfp8_ref as Q


def test_e4m3_grid_and_rounding():
    assert Q.e4m3_decode(np.array([0x00, 0x38, 0xB8, 0x40, 0x7E, 0x01, 0x08], np.uint8)).tolist() == \
        [0.0, 1.0, -1.0, 2.0, 448.0, 2.0 ** -9, 2.0 ** -6]
    # every code is a fixed point; midpoints go to the even code
    codes = np.arange(127, dtype=np.uint8)
    vals = Q.e4m3_decode(codes)
    assert np.array_equal(Q.e4m3_encode(vals), codes)
    mid = (vals[:-1] + vals[1:]) / 2
    enc = Q.e4m3_encode(mid)
    assert np.all((enc & 1) == 0) and np.all((enc == codes[:-1]) | (enc == codes[1:]))
    assert np.array_equal(Q.e4m3_encode(-vals[1:]), codes[1:] | 0x80)


def test_block_scale_rule():
    am = np.array([0.0, 448.0, 448.0001, 1.0, 1.75, 1.7500001, 3e-30, 1e30], np.float32)
    b = Q.block_scale(am).astype(np.int32)
    e = b - 127
    for a, ee in zip(am[1:], e[1:]):
        assert a <= 448.0 * 2.0 ** ee and (a > 448.0 * 2.0 ** (ee - 1) or ee == -127)
    assert b[0] == 127 and e[1] == 0 and e[2] == 1 and e[3] == -8 and e[4] == -8 and e[5] == -7


def test_quantize_roundtrip_error():
    rng = np.random.default_rng(0)
    x = (rng.standard_normal((64, 256)) * np.exp(rng.uniform(-6, 6, (64, 1)))).astype(np.float32)
    x[3, 32:64] = 0.0
    d, s = Q.quantize(x)
    assert d.shape == x.shape and s.shape == (64, 8) and s[3, 1] == 127
    y = Q.dequantize(d, s)
    blocks = np.abs(x.reshape(64, 8, 32)).max(-1, keepdims=True)
    err = np.abs(y.reshape(64, 8, 32) - x.reshape(64, 8, 32))
    # 3 mantissa bits: half an ulp is 2^-4 relative for normals; small elements of a block sit on its subnormal grid
    assert np.all(err <= np.maximum(np.abs(x.reshape(64, 8, 32)) * 2.0 ** -4, blocks * 2.0 ** -9) * 1.0001)
    assert np.array_equal(Q.dequantize(*Q.quantize(y.astype(np.float32))), y)      # idempotent
