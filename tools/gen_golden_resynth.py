"""Golden vectors for row N3's ORCHESTRATION (survey container only): runs the REFERENCE's own
``SegmentSynthesis.resynthesize`` (sylber/model/segment_synthesis.py:103-146) -- segmentation of the hidden states, per-segment mean,
optional quantiser hook, broadcast over the segment's frames, ``MLP`` conditioner, silence mask, both branches (hidden states /
caller-supplied ``features``) -- and stores inputs + outputs in tests/golden/resynth_front.npz.

What is real: the method body and the ``MLP`` / ``RFF`` classes, executed unmodified; ``get_segment`` (the reference's).
What is stubbed (none of it is on the path this repository replaces, and none of it is installed here):
  * the object is built with ``object.__new__`` -- ``__init__`` would construct the flow-matching decoder;
  * ``speech_model`` returns the seeded hidden states handed to it (the encoder is pinned by its own goldens);
  * ``cfm_wrapper.sample`` returns its ``cond_emb`` argument, i.e. the conditioning input the front half produces; ``pitch_amp`` = 1
    so that line :144 leaves it untouched;
  * the quantiser hook (:121-125) is exercised with an object that has the two members the reference calls
    (``get_indices``, ``vq.get_output_from_indices``) and performs a float64 nearest-centroid look-up: this pins WHERE the decoded
    vector goes (shapes, broadcast), not vector_quantize_pytorch's look-up itself (row N4 stays "parity unpinned").
Contains no reference code; never runs on the GPU box."""
import importlib
import os
import sys
import types

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import downstream_ref                        # noqa: E402
from sylber_amd.synth_states import syllable_states      # noqa: E402
from sylber_amd.weights import synthetic_mlp_state_dict  # noqa: E402
from tools import ref_shim                               # noqa: E402
from tools.gen_golden_mlp import _stub                   # noqa: E402
import typing                                            # noqa: E402


class _Speech(torch.nn.Module):
    def __init__(self, hidden):
        super().__init__()
        self.hidden = hidden

    def forward(self, input_values, attention_mask=None):
        return types.SimpleNamespace(last_hidden_state=self.hidden)


class _Sampler:
    def sample(self, cond_emb=None, steps=None, rand_scale=None):
        return cond_emb


class _VQ:
    def __init__(self, cent):
        self.cent = cent

    def get_output_from_indices(self, idx):              # idx [1, 1, 1] -> [1, 1, 768]
        return self.cent[idx.reshape(-1)].reshape(1, 1, -1)


class _Quantizer:
    def __init__(self, cent):
        self.cent = torch.from_numpy(cent)
        self.vq = _VQ(self.cent)

    def get_indices(self, token):                        # [1, 768] -> [1, 1]
        i, _ = downstream_ref.km_indices(token.numpy(), self.cent.numpy())
        return torch.from_numpy(np.asarray(i)).reshape(1, 1)


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
    msd = synthetic_mlp_state_dict(1)
    mlp = mod.MLP(768, output_dim=256, hidden_dims=[512, 512]).eval()
    mlp.load_state_dict(msd, strict=True)

    def make(hidden, quantizer=None):
        m = object.__new__(mod.SegmentSynthesis)
        torch.nn.Module.__init__(m)
        m.speech_model = _Speech(hidden)
        m.input_model = mlp
        m.cfm_wrapper = _Sampler()
        m.quantizer = quantizer
        m.pitch_amp = 1.0
        m.thresholder = None
        return m

    h = torch.from_numpy(np.stack([syllable_states(120, 3), syllable_states(120, 4), syllable_states(120, 5, mode="silence"),
                                   syllable_states(120, 6, mode="edge")]))
    out = {}                                             # (inputs are seeds + generators: tests/test_oracle_downstream.py rebuilds them)
    with torch.no_grad():
        art, segs = make(h).resynthesize(input_values=torch.zeros(4, 1), normthreshold=2.6, merge_threshold=0.8)
    out["cond"] = art.numpy()
    out["nseg"] = np.array([len(s) for s in segs], np.int32)
    out["segments"] = np.concatenate([np.asarray(s, np.int64).reshape(-1, 2) for s in segs], 0)
    # quantiser hook
    rng = np.random.default_rng(3)
    cent = (rng.standard_normal((500, 768)) * 0.25).astype(np.float32)
    with torch.no_grad():
        art_q, segs_q = make(h, _Quantizer(cent)).resynthesize(input_values=torch.zeros(4, 1), normthreshold=2.6, merge_threshold=0.8)
    out["cond_quantized"] = art_q.numpy()
    # features branch (:135-139)
    g = torch.Generator().manual_seed(5)
    f = torch.randn(2, 37, 768, generator=g)
    f[0, 3] = 0.0
    f[1, 10] = 5e-6
    f[1, 11] = 3e-6
    with torch.no_grad():
        art_f, segs_f = make(h).resynthesize(features=f)
    assert segs_f is None
    out["cond_features"] = art_f.numpy()
    # the repository's restatement against the reference's own method
    o_inp, o_avg, o_segs = downstream_ref.resynth_front(msd, h, 2.6, 0.8)
    d1 = float((o_inp - art).abs().max())
    assert all(np.array_equal(np.asarray(a).reshape(-1, 2), np.asarray(b).reshape(-1, 2)) for a, b in zip(o_segs, segs))
    assert np.array_equal(o_inp.numpy() == 0.0, art.numpy() == 0.0)
    q_inp, _, _ = downstream_ref.resynth_front(msd, h, 2.6, 0.8, centroids=cent)
    d2 = float((q_inp - art_q).abs().max())
    d3 = float((downstream_ref.resynth_front_features(msd, f) - art_f).abs().max())
    print("oracle restatement vs the reference's resynthesize: max abs %.2e (plain) %.2e (quantiser hook) %.2e (features branch)" % (d1, d2, d3))
    assert max(d1, d2, d3) < 1e-5
    out["oracle_vs_reference_max_abs"] = np.float64(max(d1, d2, d3))
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", "resynth_front.npz"), **out)
    print("wrote tests/golden/resynth_front.npz", {k: v.shape for k, v in out.items() if hasattr(v, "shape")})


if __name__ == "__main__":
    main()
