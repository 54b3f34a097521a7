"""Golden vector for row N3 (survey container only): runs the REFERENCE's own ``MLP`` class
(sylber/model/segment_synthesis.py:35-53) on seeded inputs with the seeded weights of
``sylber_amd.weights.synthetic_mlp_state_dict`` and stores inputs + outputs in tests/golden/mlp_front.npz.

Importing that module drags in the flow-matching decoder's dependencies (torchode, torchdiffeq, beartype,
gateloop_transformer, vector_quantize_pytorch, lightning), none of which is installed; they are replaced by EMPTY
stand-in modules here — the ``MLP`` / ``RFF`` classes exercised below are plain torch.nn code and never touch them.
Contains no reference code; never runs on the GPU box."""
import importlib
import os
import sys
import types
import typing

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import downstream_ref                        # noqa: E402
from sylber_amd.weights import synthetic_mlp_state_dict  # noqa: E402
from tools import ref_shim                               # noqa: E402


class _Anything(types.ModuleType):
    """empty stand-in: any attribute is a placeholder object (default arguments like ``to.Tsit5`` only need to exist)"""
    def __getattr__(self, item):
        if item.startswith("__"):
            raise AttributeError(item)
        return object


def _stub(name, **attrs):
    m = _Anything(name)
    for k, v in attrs.items():
        setattr(m, k, v)
    sys.modules.setdefault(name, m)
    return sys.modules[name]


def main():
    ref_shim.load()
    _stub("torchode")
    _stub("torchdiffeq", odeint=None)
    bt = _stub("beartype", beartype=lambda f: f)
    bt.typing = _stub("beartype.typing", Tuple=typing.Tuple, Union=typing.Union, Optional=typing.Optional, List=typing.List)
    _stub("gateloop_transformer", SimpleGateLoopLayer=object)
    _stub("vector_quantize_pytorch", GroupedResidualVQ=object)
    _stub("lightning", LightningModule=torch.nn.Module)
    mod = importlib.import_module("sylber.model.segment_synthesis")
    mlp = mod.MLP(768, output_dim=256, hidden_dims=[512, 512]).eval()      # sylber_configs/sylber_resynthesis.yaml
    sd = synthetic_mlp_state_dict(0)
    mlp.load_state_dict(sd, strict=True)
    g = torch.Generator().manual_seed(77)
    x = torch.randn(24, 768, generator=g) * torch.exp(torch.rand(24, 1, generator=g) * 3 - 1.5)
    x[5] = 0.0                                                             # frames outside every segment feed zeros
    with torch.no_grad():
        y = mlp(x)
        yo = downstream_ref.mlp_forward(sd, x)
    dev = float((y - yo).abs().max())
    print("reference MLP vs oracle restatement: max abs", dev)
    assert dev < 1e-5
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", "mlp_front.npz"), x=x.numpy(), y=y.numpy(),
                        oracle_vs_reference_max_abs=np.float64(dev))
    print("wrote tests/golden/mlp_front.npz")


if __name__ == "__main__":
    main()
