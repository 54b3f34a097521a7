"""CPU study: what would MXFP8 operands in the attention core (QK^T and PV) cost in accuracy on top of the shipped
fp8 mode (MXFP8 operands in the four weight GEMMs of every encoder layer)?

BASELINE configs[4] words the mode as "fp8 MFMA for attention + FFN GEMMs"; the shipped precision="fp8" keeps the
attention core on bf16 operands.  This script fake-quantises operands inside the fp32 oracle forward
(oracle/hubert_ref.py restated inline for the encoder layers) and reports the hidden-state error against the
unquantised fp32 forward, per variant.  Test infrastructure only (imports oracle/).

    python tools/fp8_attention_sim.py [--clips 4 --seconds 4]
"""
import argparse
import os
import sys

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import hubert_ref as R                                     # noqa: E402
from sylber_amd.synth import syllable_wave                             # noqa: E402
from sylber_amd.weights import synthetic_state_dict                    # noqa: E402


def mx_fake(x: torch.Tensor, dim: int = -1, block: int = 32) -> torch.Tensor:
    """quantise -> dequantise: e4m3 elements, one power-of-two scale per `block` elements along `dim`
    (the smallest 2^e with amax <= 448 * 2^e, as csrc/common.h mx_e8m0)"""
    x = x.transpose(dim, -1)
    shp = x.shape
    n = shp[-1]
    pad = (-n) % block
    xp = F.pad(x, (0, pad)).reshape(*shp[:-1], -1, block)
    amax = xp.abs().amax(-1, keepdim=True).clamp_min(1e-30)
    e = torch.ceil(torch.log2(amax / 448.0))
    s = torch.exp2(e)
    q = (xp / s).to(torch.float8_e4m3fn).float() * s
    return q.reshape(*shp[:-1], -1)[..., :n].transpose(dim, -1)


def bf16(x):
    return x.to(torch.bfloat16).float()


def p_fake(p: torch.Tensor, mode: str) -> torch.Tensor:
    """P = exp(s - rowmax) in [0, 1]; 'e4m3x256': one fixed power-of-two scale (P * 256 <= 448, underflow below 2^-17)"""
    if mode == "bf16":
        return bf16(p)
    return (p * 256.0).to(torch.float8_e4m3fn).float() / 256.0


def encoder(sd, h, gemm_q, attn):
    """encoder layers of oracle/hubert_ref.py:131-155 with operand fake-quantisation.
    gemm_q: quantiser applied to both operands of the four weight GEMMs (along K); attn: None | 'bf16' | 'fp8'"""
    B, T, _ = h.shape
    H, D = R.HEADS, R.HEAD_DIM

    def lin(x, w, b):
        return F.linear(gemm_q(x), gemm_q(w), b)

    for l in range(9):
        p = f"encoder.layers.{l}."
        q = lin(h, sd[p + "attention.q_proj.weight"], sd[p + "attention.q_proj.bias"]).view(B, T, H, D).transpose(1, 2)
        k = lin(h, sd[p + "attention.k_proj.weight"], sd[p + "attention.k_proj.bias"]).view(B, T, H, D).transpose(1, 2)
        v = lin(h, sd[p + "attention.v_proj.weight"], sd[p + "attention.v_proj.bias"]).view(B, T, H, D).transpose(1, 2)
        if attn == "bf16":
            q, k, v = bf16(q * D ** -0.5), bf16(k), bf16(v)
            s = q @ k.transpose(-1, -2)
        elif attn == "fp8":
            q, k = mx_fake(q * D ** -0.5), mx_fake(k)          # blocks of 32 along the head dim (the contraction)
            v = mx_fake(v, dim=-2)                              # blocks of 32 along the keys (the contraction of PV)
            s = q @ k.transpose(-1, -2)
        else:
            s = (q @ k.transpose(-1, -2)) * D ** -0.5
        m = s.amax(-1, keepdim=True)
        e = torch.exp(s - m)
        l_sum = e.sum(-1, keepdim=True)                         # the row sum stays fp32 (as in csrc/attention.hip)
        if attn is not None:
            e = p_fake(e, "bf16" if attn == "bf16" else "e4m3x256")
        ctx = (e @ v) / l_sum
        ctx = ctx.transpose(1, 2).reshape(B, T, R.HIDDEN)
        a = lin(ctx, sd[p + "attention.out_proj.weight"], sd[p + "attention.out_proj.bias"])
        h = F.layer_norm(h + a, (R.HIDDEN,), sd[p + "layer_norm.weight"], sd[p + "layer_norm.bias"], R.LN_EPS)
        ff = F.gelu(lin(h, sd[p + "feed_forward.intermediate_dense.weight"], sd[p + "feed_forward.intermediate_dense.bias"]))
        ff = lin(ff, sd[p + "feed_forward.output_dense.weight"], sd[p + "feed_forward.output_dense.bias"])
        h = F.layer_norm(h + ff, (R.HIDDEN,), sd[p + "final_layer_norm.weight"], sd[p + "final_layer_norm.bias"], R.LN_EPS)
    return h


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--clips", type=int, default=4)
    ap.add_argument("--seconds", type=float, default=4.0)
    a = ap.parse_args()
    torch.manual_seed(0)
    sd = {k: v.float() for k, v in synthetic_state_dict(0).items()}
    n = int(a.seconds * 16000)
    wav = torch.stack([syllable_wave(n, seed=9000 + i).reshape(-1) for i in range(a.clips)])
    st = R.forward(sd, wav, None, collect=True)
    h0, ref = st["enc_in"], st["hidden"]
    ident = lambda x: x                                                                     # noqa: E731
    chk = encoder(sd, h0, ident, None)
    assert torch.allclose(chk, ref, atol=1e-4), "the inline restatement must reproduce oracle/hubert_ref.py"

    def rel(x):
        return float(((x - ref).pow(2).mean() / ref.pow(2).mean()).sqrt())

    rows = [("bf16 operands everywhere (the bf16 mode's encoder)", bf16, "bf16"),
            ("shipped fp8 mode: MXFP8 weight GEMMs, bf16 attention core", mx_fake, "bf16"),
            ("configs[4] as worded: MXFP8 weight GEMMs + fp8 QK^T and PV", mx_fake, "fp8"),
            ("fp8 attention core only (bf16 weight GEMMs)", bf16, "fp8")]
    print("hidden-state relative RMS error against the fp32 forward (%d clips x %g s, synthetic weights seed 0)" % (a.clips, a.seconds))
    for name, gq, at in rows:
        print("  %-62s %.3e" % (name, rel(encoder(sd, h0, gq, at))))


if __name__ == "__main__":
    main()
