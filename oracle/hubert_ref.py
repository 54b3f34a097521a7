"""ORACLE (test infrastructure, never shipped, never on the product path).

CPU fp32 restatement, in plain torch functional ops, of the arithmetic the reference runs at
``sylber/model/sylber.py:122``::

    self.speech_model(batch_tensor, attention_mask=attention_mask).last_hidden_state

where ``speech_model`` is ``transformers.HubertModel`` built with ``num_hidden_layers=9``
(sylber/model/sylber.py:41).  ``transformers`` is a third-party dependency of the reference
(pinned ==4.45.2 in requirements.txt:59, unpinned in setup.py:25; 5.15.0 in the build image) and
is NOT under /root/reference, so this file restates its published algorithm
(TP = transformers/models/hubert/modeling_hubert.py):

    feature encoder   TP:154-213  (conv0 + GroupNorm(512 groups) + GELU, 6 x (conv + GELU), no conv bias)
    frame mask        TP:664-689  (valid frames = conv-length formula of attention_mask.sum(-1))
    feature proj      TP:216-231  (LayerNorm(512) -> Linear(512->768))
    encoder prologue  TP:417-442  (zero padded frames; weight-normed grouped pos-conv k=128 g=16 pad=64,
                                   drop last frame, GELU, residual; LayerNorm(768))
    encoder layer x9  TP:371-404  (post-LN: x = LN(x + Attn(x)); x = LN2(x + FFN(x)))
    attention         TP:234-344  (softmax(q k^T / 8 + key-padding mask) v, 12 heads x 64)

Parity pinning: the reference has no tests or golden vectors for this path, so this
restatement is pinned against the reference itself, imported in the build container by
``tools/gen_golden.py`` (max-abs deviation recorded in tests/golden/manifest.json), and against the
committed per-stage golden activations under tests/golden/ produced by that script.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import this module.
"""
from __future__ import annotations

import math
from typing import Dict, List, Optional, Sequence

import torch
import torch.nn.functional as F

CONV_KERNELS = (10, 3, 3, 3, 3, 2, 2)
CONV_STRIDES = (5, 2, 2, 2, 2, 2, 2)
CONV_DIM = 512
HIDDEN = 768
HEADS = 12
HEAD_DIM = 64
FFN = 3072
POS_K = 128
POS_GROUPS = 16
LN_EPS = 1e-5


def conv_out_lengths(n: int) -> List[int]:
    """Per-layer output lengths of the 7-layer feature encoder (TP:664-677)."""
    out = []
    for k, s in zip(CONV_KERNELS, CONV_STRIDES):
        n = (n - k) // s + 1
        out.append(n)
    return out


def num_frames(n: int) -> int:
    return conv_out_lengths(n)[-1]


def pos_conv_weight(sd: Dict[str, torch.Tensor]) -> torch.Tensor:
    """Effective pos-conv weight: weight_norm(dim=2): w = g * v / ||v||_(0,1)  (TP:63-80)."""
    p = "encoder.pos_conv_embed.conv."
    if p + "parametrizations.weight.original0" in sd:
        g = sd[p + "parametrizations.weight.original0"]
        v = sd[p + "parametrizations.weight.original1"]
    elif p + "weight_g" in sd:
        g, v = sd[p + "weight_g"], sd[p + "weight_v"]
    else:
        return sd[p + "weight"]
    norm = v.float().pow(2).sum(dim=(0, 1), keepdim=True).sqrt()
    return (g.float() * v.float() / norm)


def forward(sd: Dict[str, torch.Tensor], wav: torch.Tensor, lengths: Optional[Sequence[int]] = None,
            num_layers: int = 9, collect: bool = False) -> Dict[str, torch.Tensor]:
    """wav: [B, N] float32 (already padded with zeros); lengths: valid samples per row or None
    (None == no attention_mask, numerically identical to an all-ones mask).

    Returns {"hidden": [B,T,768]} plus, with collect=True, every stage the goldens pin.
    """
    sd = {k: v.float() for k, v in sd.items()}
    out: Dict[str, torch.Tensor] = {}
    B, N = wav.shape
    x = wav.float()[:, None, :]                                               # [B,1,N]
    # ---- feature encoder (TP:203-213)
    for i, (k, s) in enumerate(zip(CONV_KERNELS, CONV_STRIDES)):
        w = sd[f"feature_extractor.conv_layers.{i}.conv.weight"]
        x = F.conv1d(x, w, None, stride=s)
        if i == 0:
            if collect:
                out["conv0_raw"] = x
            x = F.group_norm(x, CONV_DIM, sd["feature_extractor.conv_layers.0.layer_norm.weight"],
                             sd["feature_extractor.conv_layers.0.layer_norm.bias"], eps=LN_EPS)
        x = F.gelu(x)
        if collect:
            out[f"conv{i}"] = x
    feats = x.transpose(1, 2)                                                  # [B,T,512]
    T = feats.shape[1]
    # ---- frame-level validity (TP:679-689)
    if lengths is not None:
        valid = torch.tensor([num_frames(int(n)) for n in lengths], dtype=torch.long)
        frame_mask = torch.arange(T)[None, :] < valid[:, None]                # [B,T] bool
    else:
        valid = torch.full((B,), T, dtype=torch.long)
        frame_mask = None
    # ---- feature projection (TP:225-231)
    h = F.layer_norm(feats, (CONV_DIM,), sd["feature_projection.layer_norm.weight"],
                     sd["feature_projection.layer_norm.bias"], LN_EPS)
    h = F.linear(h, sd["feature_projection.projection.weight"], sd["feature_projection.projection.bias"])
    if collect:
        out["proj"] = h
    # ---- encoder prologue (TP:428-442)
    if frame_mask is not None:
        h = h * frame_mask[:, :, None].to(h.dtype)
    pw = pos_conv_weight(sd)
    pos = F.conv1d(h.transpose(1, 2), pw, sd["encoder.pos_conv_embed.conv.bias"], padding=POS_K // 2,
                   groups=POS_GROUPS)
    pos = F.gelu(pos[:, :, :-1]).transpose(1, 2)
    h = h + pos
    h = F.layer_norm(h, (HIDDEN,), sd["encoder.layer_norm.weight"], sd["encoder.layer_norm.bias"], LN_EPS)
    if collect:
        out["enc_in"] = h
    # additive key-padding mask
    if frame_mask is not None:
        add_mask = torch.zeros(B, 1, 1, T)
        add_mask.masked_fill_(~frame_mask[:, None, None, :], float("-inf"))
    else:
        add_mask = None
    # ---- encoder layers (TP:371-404)
    for l in range(num_layers):
        p = f"encoder.layers.{l}."
        q = F.linear(h, sd[p + "attention.q_proj.weight"], sd[p + "attention.q_proj.bias"])
        k = F.linear(h, sd[p + "attention.k_proj.weight"], sd[p + "attention.k_proj.bias"])
        v = F.linear(h, sd[p + "attention.v_proj.weight"], sd[p + "attention.v_proj.bias"])
        q = q.view(B, T, HEADS, HEAD_DIM).transpose(1, 2)
        k = k.view(B, T, HEADS, HEAD_DIM).transpose(1, 2)
        v = v.view(B, T, HEADS, HEAD_DIM).transpose(1, 2)
        scores = torch.matmul(q, k.transpose(-1, -2)) * (HEAD_DIM ** -0.5)
        if add_mask is not None:
            scores = scores + add_mask
        probs = torch.softmax(scores, dim=-1)
        ctx = torch.matmul(probs, v).transpose(1, 2).reshape(B, T, HIDDEN)
        attn = F.linear(ctx, sd[p + "attention.out_proj.weight"], sd[p + "attention.out_proj.bias"])
        h = F.layer_norm(h + attn, (HIDDEN,), sd[p + "layer_norm.weight"], sd[p + "layer_norm.bias"], LN_EPS)
        ff = F.linear(h, sd[p + "feed_forward.intermediate_dense.weight"],
                      sd[p + "feed_forward.intermediate_dense.bias"])
        ff = F.gelu(ff)
        ff = F.linear(ff, sd[p + "feed_forward.output_dense.weight"], sd[p + "feed_forward.output_dense.bias"])
        h = F.layer_norm(h + ff, (HIDDEN,), sd[p + "final_layer_norm.weight"], sd[p + "final_layer_norm.bias"],
                         LN_EPS)
        if collect:
            out[f"layer{l}"] = h
    out["hidden"] = h
    out["valid_frames"] = valid
    return out


def flops_per_clip(n_samples: int, num_layers: int = 9) -> Dict[str, float]:
    """Algorithmic FLOPs (2*MAC) per clip — SURVEY.md §8(d) denominators."""
    L = conv_out_lengths(n_samples)
    T = L[-1]
    conv = [2.0 * L[0] * CONV_DIM * CONV_KERNELS[0]]
    for i in range(1, 7):
        conv.append(2.0 * L[i] * CONV_DIM * CONV_DIM * CONV_KERNELS[i])
    proj = 2.0 * T * CONV_DIM * HIDDEN
    pos = 2.0 * T * HIDDEN * (HIDDEN // POS_GROUPS) * POS_K
    qkvo = 4 * 2.0 * T * HIDDEN * HIDDEN
    attn = 2 * 2.0 * T * T * HIDDEN
    ffn = 2 * 2.0 * T * HIDDEN * FFN
    return {"conv": conv, "proj": proj, "pos": pos, "qkvo": qkvo, "attn": attn, "ffn": ffn,
            "layer": qkvo + attn + ffn,
            "total": sum(conv) + proj + pos + num_layers * (qkvo + attn + ffn), "frames": T}
