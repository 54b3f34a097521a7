"""Host mirrors of the two callers right behind the Segmenter path, running on the
device-resident outputs of ``HubertEncoderHIP.segment`` through the C-ABI (csrc/downstream.hip):

* ``KMQuantizer`` — sylber/model/quantizer.py:86-135: ``get_indices`` (nearest centroid) and ``decode``;
* ``SegmentConditioner`` — the front half of ``SegmentSynthesis.resynthesize`` (sylber/model/segment_synthesis.py:
  103-140): segment means broadcast back to frames -> ``MLP`` conditioner -> silence mask."""
from __future__ import annotations

import ctypes
from typing import Dict, Optional, Sequence

import numpy as np
import torch

from . import _lib


def _vp(t: torch.Tensor) -> ctypes.c_void_p:
    return ctypes.c_void_p(t.data_ptr())


def _stream(dev) -> ctypes.c_void_p:
    return ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream)


class KMQuantizer:
    """``KMQuantizer(centroids, normalize=False)`` — ``centroids`` a ``.npy`` path (like the reference) or an
    array/tensor ``[K, 768]``."""

    def __init__(self, centroids, normalize: bool = False, device="cuda"):
        self.lib = _lib.load()
        if not torch.cuda.is_available():
            raise _lib.SylberHipError("no MI355X visible to PyTorch-ROCm; the HIP path has no CPU fallback")
        if isinstance(centroids, str):
            centroids = np.load(centroids)
        c = torch.as_tensor(np.asarray(centroids) if not torch.is_tensor(centroids) else centroids, dtype=torch.float32)
        if c.dim() != 2:
            raise ValueError("centroids must be [K, D]")
        self.device = torch.device(device if device != "cuda" else "cuda:%d" % torch.cuda.current_device())
        self.centroids = c.contiguous().to(self.device)
        self.normalize = normalize

    def get_indices(self, token: torch.Tensor) -> torch.Tensor:
        """token ``[..., D]`` -> int64 indices ``[..., 1]`` (the reference's ``outputs['indices']`` for one codebook)"""
        lead = tuple(token.shape[:-1])
        x = token.reshape(-1, token.shape[-1]).to(self.device, torch.float32).contiguous()
        n, D = x.shape
        K = self.centroids.shape[0]
        idx = torch.empty(n, dtype=torch.int32, device=self.device)
        ws = torch.empty(int(self.lib.sylber_km_workspace_floats(n, K, D)), dtype=torch.float32, device=self.device)
        with torch.cuda.device(self.device):
            _lib.check(self.lib.sylber_km_assign(_vp(x), n, _vp(self.centroids), K, D, 1 if self.normalize else 0, _vp(idx), _vp(ws),
                                                 _stream(self.device)), "sylber_km_assign")
        return idx.to(torch.int64).reshape(lead + (1,))

    def decode(self, indices: torch.Tensor) -> torch.Tensor:
        """indices ``[..., >=1]`` -> centroid rows ``[..., D]`` (negative indices clipped to 0, quantizer.py:129-130)"""
        ind = indices[..., :1]
        lead = tuple(ind.shape[:-1])
        flat = ind.reshape(-1).to(self.device, torch.int32).contiguous()
        K, D = self.centroids.shape
        out = torch.empty(flat.numel(), D, dtype=torch.float32, device=self.device)
        with torch.cuda.device(self.device):
            _lib.check(self.lib.sylber_km_decode(_vp(flat), flat.numel(), _vp(self.centroids), K, D, _vp(out), _stream(self.device)),
                       "sylber_km_decode")
        return out.reshape(lead + (D,))

    def __call__(self, token: torch.Tensor) -> Dict[str, torch.Tensor]:
        idx = self.get_indices(token)
        return {"indices": idx, "quantize": self.decode(idx), "non_quantized": token}


class SegmentConditioner:
    """``MLP(input_dim, output_dim, hidden_dims)`` of segment_synthesis.py:35-53 + the frame broadcast / silence mask
    of ``resynthesize``.  ``state_dict`` uses the keys of the reference module (``mlp.0.weight`` ...; an
    ``input_model.`` prefix, as in a SegmentSynthesis checkpoint, is accepted)."""

    def __init__(self, state_dict: Dict[str, torch.Tensor], device="cuda"):
        self.lib = _lib.load()
        if not torch.cuda.is_available():
            raise _lib.SylberHipError("no MI355X visible to PyTorch-ROCm; the HIP path has no CPU fallback")
        sd = {(k[len("input_model."):] if k.startswith("input_model.") else k): v for k, v in state_dict.items()}
        lin = sorted({int(k.split(".")[1]) for k in sd if k.startswith("mlp.") and k.count(".") == 2})
        if not lin or lin != list(range(0, 2 * (len(lin) - 1) + 1, 2)):
            raise KeyError("state_dict does not look like MLP.state_dict(): %s" % sorted(sd)[:4])
        nh = len(lin) - 1
        keep = []

        def ptr(name):
            t = sd[name].detach().to("cpu", torch.float32).contiguous()
            keep.append(t)
            return ctypes.cast(t.data_ptr(), _lib.c_float_p)

        w = _lib.SylberMlpWeights()
        w.input_dim = int(sd["mlp.0.weight"].shape[1]); w.output_dim = int(sd["mlp.%d.weight" % (2 * nh)].shape[0]); w.num_hidden = nh
        for i in range(nh):
            w.hidden_dims[i] = int(sd["mlp.%d.weight" % (2 * i)].shape[0])
            h = w.hidden[i]
            h.lin_w = ptr("mlp.%d.weight" % (2 * i)); h.lin_b = ptr("mlp.%d.bias" % (2 * i))
            h.ff1_w = ptr("mlp.%d.linear1.weight" % (2 * i + 1)); h.ff1_b = ptr("mlp.%d.linear1.bias" % (2 * i + 1))
            h.ff2_w = ptr("mlp.%d.linear2.weight" % (2 * i + 1)); h.ff2_b = ptr("mlp.%d.linear2.bias" % (2 * i + 1))
            h.ln_w = ptr("mlp.%d.norm.weight" % (2 * i + 1)); h.ln_b = ptr("mlp.%d.norm.bias" % (2 * i + 1))
        w.out_w = ptr("mlp.%d.weight" % (2 * nh)); w.out_b = ptr("mlp.%d.bias" % (2 * nh))
        self.device = torch.device(device if device != "cuda" else "cuda:%d" % torch.cuda.current_device())
        self.input_dim, self.output_dim = w.input_dim, w.output_dim
        self.handle = ctypes.c_void_p()
        _lib.check(self.lib.sylber_mlp_create(ctypes.byref(w), self.device.index or 0, ctypes.byref(self.handle)), "sylber_mlp_create")

    def __del__(self):
        h = getattr(self, "handle", None)
        if h:
            self.lib.sylber_mlp_destroy(h)
            self.handle = None

    def __call__(self, hidden: torch.Tensor, seg: torch.Tensor, nseg: torch.Tensor, feats: torch.Tensor, normthreshold: float,
                 max_segments: Optional[int] = None, quantizer: Optional["KMQuantizer"] = None):
        """hidden ``[B,T,768]``, (seg, nseg, feats) as returned by ``HubertEncoderHIP.segment`` -> ``(input [B,T,out],
        averaged_target_hidden_states [B,T,768])`` — the tensors named so at segment_synthesis.py:115,138-139.
        ``quantizer``: the optional substitution inside the averaging loop (segment_synthesis.py:121-125): every segment
        mean is replaced by its nearest codebook entry (``get_indices`` -> ``get_output_from_indices``) before it is
        broadcast to its frames and fed to the MLP."""
        B, T, D = hidden.shape
        if D != self.input_dim:
            raise ValueError("hidden dim %d != MLP input dim %d" % (D, self.input_dim))
        S = int(max_segments) if max_segments is not None else max(1, int(nseg.max().item()))
        S = min(max(S, 1), T)
        if quantizer is not None:
            # only the first S slots per utterance are ever read; slots beyond an utterance's own count are ignored by
            # sylber_condition, so they may hold anything (NaN rows of unused slots map to index 0 and stay unused)
            head = torch.nan_to_num(feats[:, :S].contiguous())
            q = quantizer.decode(quantizer.get_indices(head))
            feats = feats.clone()
            feats[:, :S] = q
        cond = torch.empty(B, T, self.output_dim, dtype=torch.float32, device=self.device)
        avg = torch.empty(B, T, D, dtype=torch.float32, device=self.device)
        ws = torch.empty(int(self.lib.sylber_condition_workspace_floats(self.handle, B, S)), dtype=torch.float32, device=self.device)
        with torch.cuda.device(self.device):
            _lib.check(self.lib.sylber_condition(self.handle, _vp(hidden), _vp(seg), _vp(nseg), _vp(feats), B, T, S,
                                                 ctypes.c_float(float(np.float32(normthreshold))), _vp(avg), _vp(cond), _vp(ws),
                                                 _stream(self.device)), "sylber_condition")
        return cond, avg

    def from_features(self, features: torch.Tensor) -> torch.Tensor:
        """the ``features is not None`` branch of ``resynthesize`` (segment_synthesis.py:135-140): ``features [B,T,768]``
        (already averaged / decoded by the caller) -> ``input [B,T,out]`` = MLP(features) with the frames whose
        ``((features**2).sum(-1))**.5 < 1e-4`` zeroed (threshold and missing 1e-8 exactly as the reference)."""
        lead = tuple(features.shape[:-1])
        x = features.reshape(-1, features.shape[-1]).to(self.device, torch.float32).contiguous()
        if x.shape[1] != self.input_dim:
            raise ValueError("feature dim %d != MLP input dim %d" % (x.shape[1], self.input_dim))
        rows = x.shape[0]
        cond = torch.empty(rows, self.output_dim, dtype=torch.float32, device=self.device)
        ws = torch.empty(int(self.lib.sylber_condition_workspace_floats(self.handle, rows, 1)), dtype=torch.float32, device=self.device)
        with torch.cuda.device(self.device):
            _lib.check(self.lib.sylber_condition_features(self.handle, _vp(x), rows, _vp(cond), _vp(ws), _stream(self.device)),
                       "sylber_condition_features")
        return cond.reshape(lead + (self.output_dim,))
