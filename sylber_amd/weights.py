"""Weight schema of the hot path and a deterministic synthetic checkpoint.

The reference loads a flat ``state_dict`` with ``load_state_dict(strict=False)`` into
``transformers.HubertModel`` (sylber/model/sylber.py:41-52).  The real checkpoint (``sylber.ckpt``
on the HF hub, sylber.py:47-50) is not obtainable offline, so benchmarks, tests and golden
vectors use the seeded synthetic checkpoint built here.  Key names and shapes follow
``HubertModel.state_dict()`` for the 9-layer hubert-base geometry.
"""
from __future__ import annotations

import math
from typing import Dict

import torch

CONV_KERNELS = (10, 3, 3, 3, 3, 2, 2)
CONV_STRIDES = (5, 2, 2, 2, 2, 2, 2)
CONV_DIM = 512
HIDDEN = 768
HEADS = 12
FFN = 3072
POS_K = 128
POS_GROUPS = 16
NUM_LAYERS = 9

POS_G_KEYS = ("encoder.pos_conv_embed.conv.parametrizations.weight.original0",
              "encoder.pos_conv_embed.conv.weight_g")
POS_V_KEYS = ("encoder.pos_conv_embed.conv.parametrizations.weight.original1",
              "encoder.pos_conv_embed.conv.weight_v")


def expected_shapes(num_layers: int = NUM_LAYERS) -> Dict[str, tuple]:
    s: Dict[str, tuple] = {"masked_spec_embed": (HIDDEN,)}
    for i, k in enumerate(CONV_KERNELS):
        s[f"feature_extractor.conv_layers.{i}.conv.weight"] = (CONV_DIM, 1 if i == 0 else CONV_DIM, k)
    s["feature_extractor.conv_layers.0.layer_norm.weight"] = (CONV_DIM,)
    s["feature_extractor.conv_layers.0.layer_norm.bias"] = (CONV_DIM,)
    s["feature_projection.layer_norm.weight"] = (CONV_DIM,)
    s["feature_projection.layer_norm.bias"] = (CONV_DIM,)
    s["feature_projection.projection.weight"] = (HIDDEN, CONV_DIM)
    s["feature_projection.projection.bias"] = (HIDDEN,)
    s["encoder.pos_conv_embed.conv.bias"] = (HIDDEN,)
    s[POS_G_KEYS[0]] = (1, 1, POS_K)
    s[POS_V_KEYS[0]] = (HIDDEN, HIDDEN // POS_GROUPS, POS_K)
    s["encoder.layer_norm.weight"] = (HIDDEN,)
    s["encoder.layer_norm.bias"] = (HIDDEN,)
    for l in range(num_layers):
        p = f"encoder.layers.{l}."
        for n in ("q_proj", "k_proj", "v_proj", "out_proj"):
            s[p + f"attention.{n}.weight"] = (HIDDEN, HIDDEN)
            s[p + f"attention.{n}.bias"] = (HIDDEN,)
        s[p + "layer_norm.weight"] = (HIDDEN,)
        s[p + "layer_norm.bias"] = (HIDDEN,)
        s[p + "feed_forward.intermediate_dense.weight"] = (FFN, HIDDEN)
        s[p + "feed_forward.intermediate_dense.bias"] = (FFN,)
        s[p + "feed_forward.output_dense.weight"] = (HIDDEN, FFN)
        s[p + "feed_forward.output_dense.bias"] = (HIDDEN,)
        s[p + "final_layer_norm.weight"] = (HIDDEN,)
        s[p + "final_layer_norm.bias"] = (HIDDEN,)
    return s


def synthetic_state_dict(seed: int = 0, num_layers: int = NUM_LAYERS, final_gain: float = 0.038,
                         final_bias: float = 1.0, final_spread: float = 1.0) -> Dict[str, torch.Tensor]:
    """Seeded fp32 checkpoint with HF-init-like scales.

    Linear ~ N(0, 0.02)... scaled up so that attention and FFN actually move the residual stream
    (pure 0.02 init makes every layer a near-identity, which would hide kernel bugs); convs
    kaiming-normal; LayerNorm/GroupNorm affine drawn around (1, 0).  The LAST layer's
    ``final_layer_norm`` is given a small gain and a fixed bias direction so that frame norms
    straddle the reference's ``norm_threshold=2.6`` and adjacent-frame cosines straddle
    ``merge_threshold=0.8`` (sylber.py:35-36): with unit gain every frame norm is ~27.7 and the
    segmenter would see one all-speech segment.
    """
    g = torch.Generator().manual_seed(seed)

    def randn(*shape, std=1.0):
        return torch.randn(*shape, generator=g, dtype=torch.float32) * std

    sd: Dict[str, torch.Tensor] = {}
    shapes = expected_shapes(num_layers)
    for name, shape in shapes.items():
        if name == "masked_spec_embed":
            sd[name] = torch.rand(*shape, generator=g)
        elif name.endswith("conv.weight") and name.startswith("feature_extractor"):
            fan_in = shape[1] * shape[2]
            sd[name] = randn(*shape, std=math.sqrt(2.0 / fan_in))
        elif "layer_norm.weight" in name:
            sd[name] = 1.0 + randn(*shape, std=0.1)
        elif "layer_norm.bias" in name:
            sd[name] = randn(*shape, std=0.1)
        elif name == POS_G_KEYS[0]:
            sd[name] = 1.0 + 0.25 * torch.rand(*shape, generator=g)
        elif name == POS_V_KEYS[0]:
            sd[name] = randn(*shape, std=math.sqrt(4.0 / (POS_K * HIDDEN)))
        elif name.endswith(".bias"):
            sd[name] = randn(*shape, std=0.02)
        elif name.endswith(".weight"):
            # linear layers: fan-in scaled so activations keep O(1) scale through the stack; the
            # two residual-branch output projections are damped so that 9 random layers do not
            # collapse every frame onto one common direction (adjacent-frame cosine -> 1).
            damp = 0.3 if ("out_proj" in name or "output_dense" in name) else 1.0
            sd[name] = randn(*shape, std=damp / math.sqrt(shape[1]))
        else:
            raise KeyError(name)
    last = f"encoder.layers.{num_layers - 1}.final_layer_norm."
    sd[last + "weight"] = final_gain * torch.exp(final_spread * randn(HIDDEN))
    direction = randn(HIDDEN)
    sd[last + "bias"] = final_bias * direction / direction.norm()
    return sd


def normalize_keys(sd: Dict[str, torch.Tensor]) -> Dict[str, torch.Tensor]:
    """Accept both the parametrized (``parametrizations.weight.original{0,1}``) and the legacy
    (``weight_g`` / ``weight_v``) pos-conv names, and an optional ``speech_model.`` prefix."""
    out = {}
    for k, v in sd.items():
        for pre in ("speech_model.", "hubert."):
            if k.startswith(pre):
                k = k[len(pre):]
        if k == POS_G_KEYS[1]:
            k = POS_G_KEYS[0]
        if k == POS_V_KEYS[1]:
            k = POS_V_KEYS[0]
        out[k] = v
    return out


def fold_pos_conv_weight(sd: Dict[str, torch.Tensor]) -> torch.Tensor:
    """Effective positional-conv weight ``g * v / ||v||`` (weight_norm with dim=2: the norm runs over
    dims (0, 1) for every tap; TP:63-80).  Folded once at load time; also accepts a plain
    ``conv.weight`` entry."""
    sd = normalize_keys(sd)
    if POS_G_KEYS[0] in sd:
        g = sd[POS_G_KEYS[0]].float()
        v = sd[POS_V_KEYS[0]].float()
        return (g * v / v.pow(2).sum(dim=(0, 1), keepdim=True).sqrt()).contiguous()
    return sd["encoder.pos_conv_embed.conv.weight"].float().contiguous()


def synthetic_mlp_state_dict(seed: int = 0, input_dim: int = HIDDEN, output_dim: int = 256,
                             hidden_dims=(512, 512)) -> Dict[str, torch.Tensor]:
    """Seeded weights of the resynthesis conditioner ``MLP`` in the key layout of its ``state_dict()``
    (sylber/model/segment_synthesis.py:35-53: ``mlp.{2i}`` Linear, ``mlp.{2i+1}`` RFF with ``linear1``, ``linear2``,
    ``norm``; last entry the output Linear; sylber_configs/sylber_resynthesis.yaml gives 768 -> [512, 512] -> 256)."""
    g = torch.Generator().manual_seed(10_000 + seed)

    def randn(*shape, std=1.0):
        return torch.randn(*shape, generator=g, dtype=torch.float32) * std

    sd: Dict[str, torch.Tensor] = {}
    d_in = input_dim
    for i, d in enumerate(hidden_dims):
        sd["mlp.%d.weight" % (2 * i)] = randn(d, d_in, std=1.0 / math.sqrt(d_in))
        sd["mlp.%d.bias" % (2 * i)] = randn(d, std=0.1)
        for nm in ("linear1", "linear2"):
            sd["mlp.%d.%s.weight" % (2 * i + 1, nm)] = randn(d, d, std=1.0 / math.sqrt(d))
            sd["mlp.%d.%s.bias" % (2 * i + 1, nm)] = randn(d, std=0.1)
        sd["mlp.%d.norm.weight" % (2 * i + 1)] = 1.0 + randn(d, std=0.1)
        sd["mlp.%d.norm.bias" % (2 * i + 1)] = randn(d, std=0.1)
        d_in = d
    sd["mlp.%d.weight" % (2 * len(hidden_dims))] = randn(output_dim, d_in, std=1.0 / math.sqrt(d_in))
    sd["mlp.%d.bias" % (2 * len(hidden_dims))] = randn(output_dim, std=0.1)
    return sd
