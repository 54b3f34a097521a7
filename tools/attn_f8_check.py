"""fp8 attention core (csrc/attention.hip attention_f8_kernel) against fp32 softmax attention on the SAME quantised operands
(oracle/mxfp8_ref.py quantiser: the error left is P's e4m3 rounding + bf16 output), against the unquantised fp32 result, and
its kernel-only time against the bf16 core.  Test infrastructure (imports oracle/)."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from sylber_amd import _lib
from oracle import mxfp8_ref as Q
lib = _lib.load()
p = lambda t: ctypes.c_void_p(t.data_ptr()) if t is not None else None

def fq(x, axis):                       # quantise -> dequantise, blocks of 32 along `axis`
    x = np.moveaxis(x, axis, -1)
    d, s = Q.quantize(np.ascontiguousarray(x).reshape(-1, x.shape[-1]))
    return np.moveaxis(Q.dequantize(d, s).reshape(x.shape), -1, axis)

for (B, T, valid) in [(2, 64, None), (3, 143, [143, 100, 1]), (2, 499, [499, 300]), (1, 700, None), (6, 499, None)]:
    g = torch.Generator().manual_seed(B * 1000 + T)
    q = torch.randn(B, T, 768, generator=g); k = torch.randn(B, T, 768, generator=g); v = torch.randn(B, T, 768, generator=g)
    k[0, T // 2, :64] = 4.0 * q[0, 3, :64] / 8
    vd = torch.tensor(valid, dtype=torch.int32).cuda() if valid else None
    o = torch.full((B, T, 768), float("nan"), device="cuda")
    qd, kd, vdev = q.cuda(), k.cuda(), v.cuda()
    _lib.check(lib.sylber_op_attention(p(qd), p(kd), p(vdev), p(vd), p(o), B, T, 2, 0, None), "op_attention fp8")
    def ref(qq, kk, vv):
        qh = qq.view(B, T, 12, 64).transpose(1, 2); kh = kk.view(B, T, 12, 64).transpose(1, 2); vh = vv.view(B, T, 12, 64).transpose(1, 2)
        s = qh @ kh.transpose(-1, -2)
        if valid:
            mask = torch.arange(T)[None, :] >= torch.tensor(valid)[:, None]
            s = s.masked_fill(mask[:, None, None, :], float("-inf"))
        return (torch.softmax(s, -1) @ vh).transpose(1, 2).reshape(B, T, 768)
    exact = ref(q * 0.125, k, v)
    # same operands as the kernel: q, k blocks of 32 along the head dim; v blocks of 32 along the keys (zero padded)
    qq = torch.from_numpy(fq((q * 0.125).numpy().reshape(B, T, 24, 32), -1).reshape(B, T, 768))
    kq = torch.from_numpy(fq(k.numpy().reshape(B, T, 24, 32), -1).reshape(B, T, 768))
    Tpad = (T + 31) // 32 * 32
    vpad = np.zeros((B, Tpad, 768), np.float32); vpad[:, :T] = v.numpy()
    vq = torch.from_numpy(fq(vpad.reshape(B, Tpad // 32, 32, 768), 2).reshape(B, Tpad, 768)[:, :T].copy())
    same = ref(qq, kq, vq)
    got = o.cpu()
    e1 = (got - same).abs().max().item(); r1 = ((got - same).pow(2).mean() / same.pow(2).mean()).sqrt().item()
    r2 = ((got - exact).pow(2).mean() / exact.pow(2).mean()).sqrt().item()
    print("B %d T %4d valid %-16s | vs same operands: max-abs %.3e rel-rms %.3e | vs unquantised: rel-rms %.3e | finite %s" %
          (B, T, valid, e1, r1, r2, bool(torch.isfinite(got).all())), flush=True)
ms = ctypes.c_float()
for (B, T) in [(32, 499), (8, 2999)]:
    r = []
    for prec in (0, 2, 0, 2):
        _lib.check(lib.sylber_debug_attention_bench(B, T, prec, 20, ctypes.byref(ms)), "attention bench")
        r.append(ms.value * 1e3)
    fl = 4.0 * B * 12 * T * T * 64
    print("B %d T %d: bf16 %.1f / %.1f us (%.0f TF)   fp8 %.1f / %.1f us (%.0f TF)" % (B, T, r[0], r[2], fl / (min(r[0], r[2]) * 1e-6) / 1e12, r[1], r[3], fl / (min(r[1], r[3]) * 1e-6) / 1e12))
