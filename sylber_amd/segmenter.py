"""Drop-in ``Segmenter`` for the MI355X: same constructor and ``__call__`` contract as the reference
(sylber/model/sylber.py:28-138), with the arithmetic on hand-written HIP kernels behind the C-ABI of
include/sylber_hip.h.  Host code stays Python on PyTorch-ROCm (device memory, streams)."""
from __future__ import annotations

import ctypes
import os
import collections
import threading
import time
import weakref
from pathlib import Path
from typing import Dict, List, Optional, Sequence, Union

import numpy as np
import torch

from . import _lib
from .ingest import ingest_file
from .weights import (NUM_LAYERS, expected_shapes, fold_pos_conv_weight, normalize_keys, synthetic_state_dict)

FRAME_RATE = 50  # sylber.py:132


def _ptr(t: torch.Tensor):
    return ctypes.cast(t.data_ptr(), _lib.c_float_p)


def _stream_ptr(device: torch.device) -> ctypes.c_void_p:
    return ctypes.c_void_p(torch.cuda.current_stream(device).cuda_stream)


def load_wav_file(path: str) -> torch.Tensor:
    """PCM ``.wav`` -> float32 [C, N] in [-1, 1) on the HOST (what torchaudio.load returns at sylber.py:83), 16 kHz
    files only.  Kept for tests and tools; ``Segmenter.__call__(wav_file=...)`` ingests on the device instead
    (sylber_amd/ingest.py: decode + resample + normalise in HIP)."""
    from .ingest import read_pcm
    pcm = read_pcm(path)
    if pcm.sample_rate != 16000:
        raise NotImplementedError("load_wav_file is the host-side 16 kHz reader (got %d Hz); use sylber_amd.ingest" % pcm.sample_rate)
    raw, width = pcm.data, pcm.sample_width
    if width == 2:
        x = raw.view("<i2").astype(np.float32) / 32768.0
    elif width == 4:
        x = raw.view("<i4").astype(np.float32) / 2147483648.0
    elif width == 1:
        x = (raw.astype(np.float32) - 128.0) / 128.0
    else:
        b = raw.reshape(-1, 3).astype(np.int32)
        v = b[:, 0] | (b[:, 1] << 8) | (b[:, 2] << 16)
        x = (v - ((v & 0x800000) << 1)).astype(np.float32) / 8388608.0
    return torch.from_numpy(x.reshape(-1, pcm.channels).T.copy())


class PinnedOutputPool:
    """Persistent page-locked host blocks for the results of ``Segmenter.__call__`` (reference: the D2H of
    sylber.py:122-138, where ``.cpu().numpy()`` lands in pageable memory).

    Page-locking is the expensive part of a pinned allocation (tens to hundreds of ms for the 49 MB of hidden states of
    a 32 x 10 s batch), so blocks are allocated once and LEASED: a call's numpy results are views of one block, and the
    block returns to the pool when the last of those views is garbage collected (a finalizer on the owning ndarray --
    every numpy view of it keeps that owner alive through ``.base``).  Nothing is ever overwritten while a caller can
    still see it.  At most ``max_leased`` blocks are out at a time: beyond that (a corpus loop that keeps every result)
    ``lease`` returns None and the caller falls back to pageable per-utterance copies, so retained results never pin
    more than ``max_leased`` batches of host memory."""

    GRANULE = 1 << 20

    def __init__(self, max_leased: int = 4, alloc=None):
        self.max_leased = int(max_leased)
        self._free: List[torch.Tensor] = []
        self._leased = 0
        self._lock = threading.Lock()
        # blocks handed back by finalizers.  A finalizer can run at ANY allocation point -- including inside lease() on the
        # same thread while it holds the (non-reentrant) lock, when a collection frees owner arrays caught in a reference
        # cycle -- so the release path takes no lock: deque.append is atomic, and lease() drains the deque under the lock
        self._returned = collections.deque()
        self._alloc = alloc or (lambda cap: torch.empty(cap, dtype=torch.uint8, pin_memory=True))   # (tests inject pageable memory)
        self.allocations = 0                      # page-locking events so far (tests / bench: must stop growing)

    def _release(self, blk: torch.Tensor) -> None:
        self._returned.append(blk)

    def _drain(self) -> None:                     # (lock held)
        while True:
            try:
                blk = self._returned.popleft()
            except IndexError:
                return
            self._leased -= 1
            self._free.append(blk)

    @property
    def leased(self) -> int:
        with self._lock:
            self._drain()
            return self._leased

    def trim(self) -> None:
        """drop free blocks beyond what ``max_leased`` can ever hand out again (after a temporarily raised budget: Segmenter.stream)"""
        with self._lock:
            self._drain()
            while self._free and len(self._free) + self._leased > self.max_leased:
                self._free.pop(0)

    def lease(self, nbytes: int):
        """-> (owner ndarray uint8 [cap], block tensor) or None when ``max_leased`` blocks are already out"""
        with self._lock:
            self._drain()
            if self._leased >= self.max_leased:
                return None
            pick = None
            for i, b in enumerate(self._free):
                if b.numel() >= nbytes and (pick is None or b.numel() < self._free[pick].numel()):
                    pick = i
            blk = self._free.pop(pick) if pick is not None else None
            if blk is None and len(self._free) + self._leased >= self.max_leased and self._free:
                self._free.pop(0)                 # too small for this batch shape: let it go instead of hoarding
            self._leased += 1
        if blk is None:
            cap = (int(nbytes) + self.GRANULE - 1) // self.GRANULE * self.GRANULE
            try:
                blk = self._alloc(cap)
            except BaseException:
                with self._lock:
                    self._leased -= 1
                raise
            self.allocations += 1
        owner = blk.numpy()
        weakref.finalize(owner, self._release, blk)
        return owner, blk


class HubertEncoderHIP:
    """The (1) boundary of include/sylber_hip.h: weights -> handle, waveform batch -> hidden states."""

    supports_out = True          # forward / segment take caller-owned output tensors (out=): ShardedSegmenter.run_stream's buffer ring

    def __init__(self, state_dict: Dict[str, torch.Tensor], num_layers: int = NUM_LAYERS, device: str = "cuda",
                 precision: str = "bf16"):
        self.lib = _lib.load()
        if not torch.cuda.is_available():
            raise _lib.SylberHipError("no MI355X visible to PyTorch-ROCm; the HIP path has no CPU fallback")
        self.device = torch.device(device if device != "cuda" else "cuda:%d" % torch.cuda.current_device())
        self.num_layers = num_layers
        sd = normalize_keys(state_dict)
        shapes = expected_shapes(num_layers)
        keep = {}

        def get(name):
            if name not in sd:
                raise KeyError("checkpoint is missing %s" % name)
            t = sd[name].detach().to("cpu", torch.float32).contiguous()
            if name in shapes and tuple(t.shape) != shapes[name]:
                raise ValueError("%s has shape %s, expected %s" % (name, tuple(t.shape), shapes[name]))
            keep[name] = t
            return _ptr(t)

        w = _lib.SylberWeights()
        w.num_layers = num_layers
        for i in range(7):
            w.conv_w[i] = get(f"feature_extractor.conv_layers.{i}.conv.weight")
        w.gn_w = get("feature_extractor.conv_layers.0.layer_norm.weight")
        w.gn_b = get("feature_extractor.conv_layers.0.layer_norm.bias")
        w.fp_ln_w = get("feature_projection.layer_norm.weight")
        w.fp_ln_b = get("feature_projection.layer_norm.bias")
        w.fp_w = get("feature_projection.projection.weight")
        w.fp_b = get("feature_projection.projection.bias")
        pos = fold_pos_conv_weight(sd)
        keep["pos"] = pos
        w.pos_w = _ptr(pos)
        w.pos_b = get("encoder.pos_conv_embed.conv.bias")
        w.enc_ln_w = get("encoder.layer_norm.weight")
        w.enc_ln_b = get("encoder.layer_norm.bias")
        for l in range(num_layers):
            p = f"encoder.layers.{l}."
            L = w.layers[l]
            L.q_w, L.q_b = get(p + "attention.q_proj.weight"), get(p + "attention.q_proj.bias")
            L.k_w, L.k_b = get(p + "attention.k_proj.weight"), get(p + "attention.k_proj.bias")
            L.v_w, L.v_b = get(p + "attention.v_proj.weight"), get(p + "attention.v_proj.bias")
            L.o_w, L.o_b = get(p + "attention.out_proj.weight"), get(p + "attention.out_proj.bias")
            L.ln1_w, L.ln1_b = get(p + "layer_norm.weight"), get(p + "layer_norm.bias")
            L.ff1_w, L.ff1_b = (get(p + "feed_forward.intermediate_dense.weight"),
                                get(p + "feed_forward.intermediate_dense.bias"))
            L.ff2_w, L.ff2_b = get(p + "feed_forward.output_dense.weight"), get(p + "feed_forward.output_dense.bias")
            L.ln2_w, L.ln2_b = get(p + "final_layer_norm.weight"), get(p + "final_layer_norm.bias")
        h = ctypes.c_void_p()
        prec = {"bf16": 0, "fp32": 1, "fp8": 2, "fp16": 3, "mixed16": 4, "split16": 5}[precision]
        _lib.check(self.lib.sylber_create(ctypes.byref(w), self.device.index or 0, prec, ctypes.byref(h)),
                   "sylber_create")                 # (the library restores the caller's current device itself)
        self.handle = h
        del keep

    def __del__(self):
        h = getattr(self, "handle", None)
        if h:
            self.lib.sylber_destroy(h)
            self.handle = None

    def num_frames(self, n_samples: int) -> int:
        return int(self.lib.sylber_num_frames(int(n_samples)))

    def padded_frames(self, n_samples: int) -> int:
        """frame pitch per utterance of the library's activation buffers (>= num_frames, multiple of 32)"""
        return int(self.lib.sylber_padded_frames(int(n_samples)))

    def set_option(self, key: int, value: int) -> None:
        """per-handle tuning / test override (include/sylber_hip.h SYLBER_OPT_*; value < 0 = automatic)"""
        _lib.check(self.lib.sylber_set_option(self.handle, int(key), int(value)), "sylber_set_option")

    def forward(self, wav: torch.Tensor, lengths: Optional[Sequence[int]] = None, stop_stage: int = 0,
                out: Optional[torch.Tensor] = None) -> torch.Tensor:
        """wav: [B, Lmax] float32 on this device, zero padded.  Returns [B, T, 768] float32 (device)."""
        assert wav.is_cuda and wav.dtype == torch.float32 and wav.dim() == 2 and wav.is_contiguous()
        B, Lmax = wav.shape
        T = self.num_frames(Lmax)
        width = 512 if stop_stage == 1 else 768
        if out is None:
            out = torch.empty(B, T, width, dtype=torch.float32, device=wav.device)
        larr = None
        if lengths is not None:
            larr = (ctypes.c_int32 * B)(*[int(x) for x in lengths])
        self.lib.sylber_set_stop_stage(self.handle, int(stop_stage))
        cur = torch.cuda.current_stream(wav.device)
        use = cur
        if getattr(self, "_graph_stream", None) is not None and cur.cuda_stream == 0:
            # a hipGraph cannot be captured on the default stream: run on the handle's own stream, ordered both ways
            use = self._graph_stream
            use.wait_stream(cur)
        with torch.cuda.device(wav.device):
            st = self.lib.sylber_forward(self.handle, ctypes.c_void_p(wav.data_ptr()), larr, B, Lmax,
                                         ctypes.c_void_p(out.data_ptr()), ctypes.c_void_p(use.cuda_stream))
        if use is not cur:
            cur.wait_stream(use)
        self.lib.sylber_set_stop_stage(self.handle, 0)
        _lib.check(st, "sylber_forward")
        return out

    def set_batches_in_flight(self, n: int) -> None:
        """tell the handle whether it owns the chip (n = 1, the default: a synchronous caller) or shares it with other in-flight batches on other
        handles / streams (n >= 2: bench.py's pipeline, ShardedSegmenter's engines).  Only the GEMM tile choice depends on it
        (SYLBER_OPT_GEMM_MODEL, include/sylber_hip.h); results are bit-identical."""
        self.set_option(12, 5 if int(n) >= 2 else 0)

    def set_graph_mode(self, enable: bool = True) -> None:
        """Replay the forward's ~110 kernel launches from a captured hipGraph (per (B, Lmax, input, output buffers);
        captured on the second call with the same key).  For launch-bound small batches; pass ``out=`` and reuse the
        input buffer so that the key repeats."""
        _lib.check(self.lib.sylber_set_graph_mode(self.handle, 1 if enable else 0), "sylber_set_graph_mode")
        self._graph_stream = torch.cuda.Stream(device=self.device) if enable else None

    def segment(self, hidden: torch.Tensor, norm_threshold: float, merge_threshold: float, with_features: bool = True,
                out=None):
        """hidden: [B, T, 768] float32 device.  Returns (segments [B,T,2] int64, nseg [B] int32, feats [B,T,768]).
        Runs on the CURRENT torch stream; it touches no encoder workspace, so a caller may run it on a side
        stream concurrently with the next batch's forward; it uses the handle's own scratch slab (frame norms, slot table), so the
        segment calls of one handle must be ordered on one stream."""
        assert hidden.is_cuda and hidden.dtype == torch.float32 and hidden.is_contiguous()
        B, T, D = hidden.shape
        if out is not None:
            seg, nseg, feats = out
        else:
            seg = torch.empty(B, T, 2, dtype=torch.int64, device=hidden.device)
            nseg = torch.empty(B, dtype=torch.int32, device=hidden.device)
            feats = torch.empty(B, T, D, dtype=torch.float32, device=hidden.device) if with_features else None
        with torch.cuda.device(hidden.device):
            st = self.lib.sylber_segment(self.handle, ctypes.c_void_p(hidden.data_ptr()), B, T, D,
                                         ctypes.c_float(float(np.float32(norm_threshold))),
                                         ctypes.c_float(float(np.float32(merge_threshold))),
                                         ctypes.c_void_p(seg.data_ptr()), ctypes.c_void_p(nseg.data_ptr()),
                                         ctypes.c_void_p(feats.data_ptr()) if feats is not None else None,
                                         _stream_ptr(hidden.device))
        _lib.check(st, "sylber_segment")
        return seg, nseg, feats

    def set_profiling(self, on: bool) -> None:
        self.lib.sylber_set_profiling(self.handle, 1 if on else 0)

    def get_profile(self) -> Dict[str, float]:
        cap = 64
        names = (ctypes.c_char_p * cap)()
        ms = (ctypes.c_float * cap)()
        n = self.lib.sylber_get_profile(self.handle, names, ms, cap)
        return {names[i].decode(): float(ms[i]) for i in range(max(n, 0))}

    def workspace_bytes(self) -> int:
        return int(self.lib.sylber_workspace_bytes(self.handle))

    def fp16_audit(self, start: bool = False):
        """fp16 headroom audit (include/sylber_hip.h SYLBER_OPT_FP16_AUDIT; precision "fp16" / "mixed16" only -- the other modes have nothing to
        saturate).  ``fp16_audit(start=True)`` (re)starts it: from then on every forward scans each 16-bit activation buffer behind its
        producer.  ``fp16_audit()`` returns ``{stage: {"saturated": values clamped at +-65504, "max_abs": largest magnitude}}`` since the
        start (synchronises the device), or ``{}`` when it never ran.  A non-zero ``saturated`` anywhere means the checkpoint does not fit
        IEEE half there: use "bf16" or "split16"."""
        if start:
            self.set_option(10, 1)
            return None
        cap = 32
        names = (ctypes.c_char_p * cap)()
        sat = (ctypes.c_uint32 * cap)()
        mx = (ctypes.c_float * cap)()
        n = self.lib.sylber_get_fp16_audit(self.handle, names, sat, mx, cap)
        if n < 0:
            _lib.check(1, "sylber_get_fp16_audit")
        return {names[i].decode(): {"saturated": int(sat[i]), "max_abs": float(mx[i])} for i in range(n)}


class Segmenter:
    """Same signature as the reference's ``Segmenter`` (sylber/model/sylber.py:30-39, 63)."""

    def __init__(self, model_ckpt="sylber", speech_upstream="facebook/hubert-base-ls960", ema_decay=0.999,
                 encoding_layer=9, merge_threshold=0.8, norm_threshold=2.6, device="cuda", **kwargs):
        self.encoding_layer = encoding_layer
        self.enc_dim = 768
        state_dict = self._load_state_dict(model_ckpt, encoding_layer)
        if "cuda" not in str(device):
            raise _lib.SylberHipError("sylber_amd.Segmenter runs on the MI355X only (device=%r)" % (device,))
        self.speech_model = HubertEncoderHIP(state_dict, num_layers=encoding_layer, device=device,
                                             precision=kwargs.get("precision", "bf16"))
        self.device = str(self.speech_model.device)
        # __call__ is synchronous: ONE batch in flight on this handle, where the residual GEMMs' K loops prefetching all three
        # fragment columns of the residual rows is the faster setting (-1.0 % of the forward; with two batches in flight on
        # two handles, as bench.py's pipeline runs, one column is: profiles/r04_resln_prefetch.md).  Bit-identical either way.
        self._resln_prefetch = int(kwargs.get("resln_prefetch", 3))
        self.speech_model.set_option(6, self._resln_prefetch)
        self.norm_threshold = norm_threshold
        self.merge_threshold = merge_threshold
        # where __call__'s numpy results live: "pinned" (default) = views of leased page-locked blocks, at most
        # `max_pinned_batches` batches outstanding, pageable copies beyond that; "pageable" = always ordinary arrays
        # (what the reference returns).  See PinnedOutputPool.
        self.output_memory = kwargs.get("output_memory", "pinned")
        if self.output_memory not in ("pinned", "pageable"):
            raise ValueError("output_memory must be 'pinned' or 'pageable'")
        # Round 6 (measured, OFF by default): `call_split = n >= 2` cuts a large batch of host tensors into n sub-batches that go through the machinery of
        # ``stream`` inside the one synchronous call (upload of part 2 under the forward of part 1, download of part 1 under the forward of part 2; every
        # part padded to the WHOLE batch's longest clip, so the call returns exactly the unsplit call's bits).  It is SLOWER on every shape tried
        # (32 x 10 s: 7.4 -> 8.2 ms with two parts, 9.7 with three; 16 x 10 s 4.5 -> 5.5; 8 x 60 s equal): a half batch's forward is not half a forward, and
        # the download of one part runs as a shader copy beside the 160-KiB-LDS GEMM workgroups of the other part's forward (tools/api_split_ab.py).
        self._call_split = max(0, int(kwargs.get("call_split", 0)))
        # (a split call leases one page-locked block per part: `max_pinned_batches` counts blocks)
        self.out_pool = PinnedOutputPool(max_leased=int(kwargs.get("max_pinned_batches", 4)))
        # host padding of tensor inputs (encode_batch).  tools/pad_probe.py on the 256-cpu boxes: the 20 MB copy of a 32 x 10 s batch into the
        # page-locked staging buffer takes 0.49 ms on one thread, 0.36-0.44 on two, MORE on four / eight / sixteen (0.58 / 0.93 / 1.25: waking
        # pool threads costs more than the copy they take over; numpy slice assignment and ctypes.memmove alike)
        self._fill_threads = max(1, int(kwargs.get("host_pad_threads", 2)))
        self._fill_groups = max(1, int(kwargs.get("host_pad_groups", 8)))       # row groups the padding + H2D of a batch is cut into (H2D of group g under the padding of g + 1)
        self._kcap_seen = 128                                                    # segment slots per utterance the next block is sized for
        self._kcap_recent = collections.deque(maxlen=16)                         # per-batch maxima of the last 16 batches (sizing decays with them)
        self._overlap_d2h = bool(kwargs.get("overlap_d2h", True))               # hidden-state D2H under the segmenter (A/B switch)
        # which keys of the reference's dict (sylber.py:134-138) a call returns.  Default = all three, the reference's contract.  A caller
        # that consumes only the tables / pooled features (tokenisation, the resynthesis front half) can drop "hidden_states": that skips
        # the 49 MB device-to-host copy of a 32 x 10 s batch (0.87 ms of a 7.9 ms call) and its page-locked block (round 6)
        outs = tuple(kwargs.get("outputs", ("segments", "segment_features", "hidden_states")))
        bad = [o for o in outs if o not in ("segments", "segment_features", "hidden_states")]
        if bad or "segments" not in outs:
            raise ValueError("outputs must contain 'segments' and may add 'segment_features', 'hidden_states' (got %r)" % (outs,))
        self.outputs = outs
        self._want_hidden = "hidden_states" in outs
        self._want_feats = "segment_features" in outs

    @staticmethod
    def _load_state_dict(model_ckpt, encoding_layer):
        if model_ckpt is None:
            # the reference keeps HubertModel's random init here (sylber.py:41,46); use the seeded one
            return synthetic_state_dict(0, num_layers=encoding_layer)
        if isinstance(model_ckpt, dict):
            return model_ckpt
        if model_ckpt == "sylber":
            model_ckpt = "sylber.ckpt"
        if not Path(model_ckpt).exists():
            from huggingface_hub import hf_hub_download  # sylber.py:49-50
            model_ckpt = hf_hub_download(repo_id="cheoljun95/sylber", filename=model_ckpt)
        sd = torch.load(model_ckpt, map_location="cpu")
        print("Pre-trained checkpoint loaded")
        return sd

    # -- batching exactly like sylber.py:76-118 ---------------------------------------------------
    def _collect(self, wav_file, wav):
        batch_wavs: List[torch.Tensor] = []
        if wav_file is not None:
            is_batch = isinstance(wav_file, list)
            for f in (wav_file if is_batch else [wav_file]):
                # sylber.py:83-86 on the device: decode, resample to 16 kHz, (x - mean) / unbiased std
                batch_wavs.append(ingest_file(f, self.speech_model.device, normalize=True))
        else:
            assert wav is not None
            is_batch = isinstance(wav, list)
            batch_wavs = wav if is_batch else [wav]
        return batch_wavs, is_batch

    @staticmethod
    def _rows(batch_wavs: Sequence[torch.Tensor]):
        rows, lengths = [], []
        for w in batch_wavs:
            if not torch.is_tensor(w) or w.dim() != 2:
                raise ValueError("each wav must be a [channels, N] tensor (sylber.py:96 reads wav.shape[1])")
            for ch in range(w.shape[0]):              # torch.cat(dim=0) makes every channel a batch row
                rows.append(w[ch])
                lengths.append(int(w.shape[1]))
        return rows, lengths

    def encode_batch(self, batch_wavs: Sequence[torch.Tensor]):
        """Pads to the batch max (sylber.py:93-118) and runs the HIP forward.  Returns the device
        hidden states [B,T,768] (full padded T, like the reference) and the per-row lengths."""
        tr = self.__dict__.get("_trace")
        mark = (lambda name: tr.append((name, time.perf_counter()))) if tr is not None else (lambda name: None)
        rows, lengths = self._rows(batch_wavs)
        lmax = max(lengths)
        dev = self.speech_model.device
        mark("  rows collected")
        if all(not r.is_cuda for r in rows):
            # host inputs: pad on the host into a pinned staging buffer and cross PCIe once (a row-by-row copy loop
            # costs a host round trip per utterance).  Two buffers rotate, each with the event of the H2D copy that last
            # read it: encode_batch returns without synchronising, so the host must not refill a buffer whose copy is
            # still queued behind an earlier forward.
            stage, slot = self._stage_buffer((len(rows), lmax))
            # the padding runs through numpy views, NOT torch CPU ops: on a many-core host every small torch op wakes
            # the intra-op thread pool (128 threads on the MI355X boxes), which made one call cost anything from 10 ms
            # to 1.8 s (tools/api_profile.py: 9.4-9.9 ms with one thread, 9.8-188 ms with the default pool)
            stage_np = stage.numpy()
            mark("  staging buffer free")
            batch = torch.empty(len(rows), lmax, dtype=torch.float32, device=dev)
            mark("  device batch allocated")

            def fill(lo, hi):
                for i in range(lo, hi):
                    src = rows[i].detach()
                    stage_np[i, : lengths[i]] = (src if src.dtype == torch.float32 else src.to(torch.float32)).numpy()
                    stage_np[i, lengths[i]:] = 0.0
            # round 4: the 20 MB host copy of a 32 x 10 s batch is 0.47 ms on one thread, 0.26 ms on two, slower again on four or eight
            # (probed on the 256-cpu boxes).  Row groups are padded by a small
            # private thread pool (numpy's copy releases the GIL) and each group crosses PCIe as soon as it is padded, so the
            # H2D of group g runs under the padding of group g + 1
            nrow = len(rows)
            # round 6: twice as many groups as threads -- with one group per thread both groups finish together and the copies only
            # START when the padding is over (tools/api_timeline.py: H2D done 1.25 ms into the call against 0.26 + 0.37 of work)
            ngrp = min(self._fill_groups, max(1, nrow // 4)) if nrow * lmax >= (1 << 20) else 1
            if ngrp <= 1:
                fill(0, nrow)
                batch.copy_(stage, non_blocking=True)
            else:
                bounds = [(g * nrow // ngrp, (g + 1) * nrow // ngrp) for g in range(ngrp)]
                futs = [self._fill_pool().submit(fill, lo, hi) for lo, hi in bounds]
                for (lo, hi), f in zip(bounds, futs):
                    f.result()
                    batch[lo:hi].copy_(stage[lo:hi], non_blocking=True)
            slot["event"].record(torch.cuda.current_stream(dev))
            mark("  padded, H2D issued")
            gtr = self.__dict__.get("_gpu_trace")
            if gtr is not None:
                ev = torch.cuda.Event(enable_timing=True)
                ev.record(torch.cuda.current_stream(dev))
                gtr.append(("H2D done", ev))
        else:
            batch = torch.zeros(len(rows), lmax, dtype=torch.float32, device=dev)
            for i, r in enumerate(rows):
                batch[i, : lengths[i]] = r.to(dev, torch.float32, non_blocking=True)
        hidden = self.speech_model.forward(batch, lengths)
        return hidden, lengths

    def _fill_pool(self):
        pool = self.__dict__.get("_fill_executor")
        if pool is None:
            from concurrent.futures import ThreadPoolExecutor
            pool = ThreadPoolExecutor(max_workers=self._fill_threads, thread_name_prefix="sylber-pad")
            self._fill_executor = pool
        return pool

    def _stage_buffer(self, shape):
        """next of two grow-only pinned H2D staging buffers; waits (host side) for the copy that last read it"""
        ring = self.__dict__.setdefault("_stage_ring", [{"buf": None, "event": None}, {"buf": None, "event": None}])
        k = self.__dict__.get("_stage_next", 0)
        self._stage_next = (k + 1) % len(ring)
        slot = ring[k]
        n = 1
        for d in shape:
            n *= int(d)
        if slot["event"] is not None:
            slot["event"].synchronize()
        else:
            slot["event"] = torch.cuda.Event()
        if slot["buf"] is None or slot["buf"].numel() < n:
            slot["buf"] = torch.empty(max(n, 1), dtype=torch.float32, pin_memory=True)
        return slot["buf"][:n].view(*shape), slot

    def segment(self, input_values=None, features=None, attention_mask=None, mergethreshold=None, normthreshold=None,
                **kwargs):
        """Tensor-native sibling of ``__call__`` with the signature of the reference's ``Sylber.segment``
        (sylber/model/sylber.py:208-247): a padded ``[B, N]`` waveform batch (+ 0/1 ``attention_mask``) or
        precomputed ``features [B, T, 768]`` in, ``(features, segments, avg_fts)`` out — ``segments`` a list of
        int64 ``[n, 2]`` arrays, ``avg_fts`` the segment means zero-padded to ``[B, max(n, 1), 768]`` on the
        device (an utterance without segments contributes one zero row, sylber.py:238-241)."""
        dev = self.speech_model.device
        if features is None:
            x = input_values.to(dev, torch.float32).contiguous()
            lengths = None if attention_mask is None else [int(v) for v in attention_mask.sum(-1).tolist()]
            features = self.speech_model.forward(x, lengths)
        else:
            features = features.to(dev, torch.float32).contiguous()
        nt = self.norm_threshold if normthreshold is None else normthreshold
        mt = self.merge_threshold if mergethreshold is None else mergethreshold
        seg, nseg, feats = self.speech_model.segment(features, nt, mt)
        nseg_h = nseg.cpu().numpy()
        nmax = max(int(nseg_h.max()), 1)
        seg_h = seg[:, :nmax].cpu().numpy()
        segments = [seg_h[i, : int(nseg_h[i])].copy() if nseg_h[i] > 0 else np.array([]) for i in range(len(nseg_h))]
        keep = torch.arange(nmax, device=dev)[None, :] < nseg[:, None]
        avg_fts = torch.where(keep[:, :, None], feats[:, :nmax], torch.zeros((), device=dev))
        return features, segments, avg_fts

    def __call__(self, wav_file=None, wav=None, in_second=True):
        tr = self.__dict__.get("_trace")                     # tools/api_timeline.py: list of (label, host time) marks
        mark = (lambda name: tr.append((name, time.perf_counter()))) if tr is not None else (lambda name: None)
        gtr = self.__dict__.get("_gpu_trace")                # tools/api_timeline.py: list of (label, timing event) recorded on the streams involved
        def gmark(name, stream=None):
            if gtr is not None:
                ev = torch.cuda.Event(enable_timing=True)
                ev.record(stream if stream is not None else torch.cuda.current_stream(self.speech_model.device))
                gtr.append((name, ev))
        mark("enter")
        gmark("enter")
        batch_wavs, is_batch = self._collect(wav_file, wav)
        parts = self._split_plan(batch_wavs) if (is_batch and tr is None and gtr is None) else None
        if parts is not None:
            return self._call_in_parts(parts, in_second)
        hidden, _ = self.encode_batch(batch_wavs)
        gmark("forward done")
        mark("padded, H2D and forward issued")
        # D2H (sylber.py:122-138's .cpu().numpy()) into ONE leased page-locked block (PinnedOutputPool: persistent blocks, no
        # page-locking per call); the numpy results are views of that block -- no second host copy of the 49 MB of hidden
        # states per 32 x 10 s batch -- and the block goes back to the pool when the caller drops them.
        # Round 4: the block is leased BEFORE the segmenter runs, sized for the largest segment-slot count seen so far, so
        # that the hidden states (the bulk: 0.87 ms over PCIe) leave on a copy stream WHILE boundary detection runs
        # (0.24 ms + the host's wait for the counts); a batch with more segments than that gets a second block for its
        # tables (rare, and the sizes repeat from then on).
        dev = hidden.device
        cur = torch.cuda.current_stream(dev)
        B, T, D = hidden.shape

        def al(n):
            return (n + 255) & ~255
        want_h, want_f = self._want_hidden, self._want_feats
        hid_bytes = al(B * T * D * 4) if want_h else 0

        def sizes(kc):
            return hid_bytes, hid_bytes + al(B * kc * 2 * 8), hid_bytes + al(B * kc * 2 * 8) + (al(B * kc * D * 4) if want_f else 0)
        kcap = min(T, self._kcap_seen)
        o_seg, o_feat, need = sizes(kcap)
        lease = self.out_pool.lease(need) if self.output_memory == "pinned" else None
        handed = lease is not None
        owner, blk = lease if handed else self._scratch_block(need)
        copy_s = self.__dict__.get("_copy_stream")
        if copy_s is None or copy_s.device != dev:
            copy_s = self._copy_stream = torch.cuda.Stream(device=dev)
        fwd_done = self.__dict__.setdefault("_ev_fwd", torch.cuda.Event())
        fwd_done.record(cur)
        if self._overlap_d2h and want_h:
            with torch.cuda.stream(copy_s):
                copy_s.wait_event(fwd_done)
                blk[:B * T * D * 4].view(torch.float32).view(B, T, D).copy_(hidden, non_blocking=True)
                gmark("hidden states D2H done", copy_s)
            hidden.record_stream(copy_s)
        seg, nseg, feats = self.speech_model.segment(hidden, self.norm_threshold, self.merge_threshold, with_features=want_f)
        gmark("boundary detection done")
        nseg_pin = self._nseg_pinned(B)
        nseg_pin.copy_(nseg, non_blocking=True)
        counted = self.__dict__.setdefault("_ev_counts", torch.cuda.Event())
        counted.record(cur)
        mark("lease, segmenter issued")
        counted.synchronize()                                # the counts are on the host
        mark("counts on the host (forward + boundary detection done)")
        nseg_h = nseg_pin.numpy().copy()
        nmax = int(nseg_h.max()) if len(nseg_h) else 0
        k = max(nmax, 1)
        towner, tblk = owner, blk
        self._note_segments(k)
        if k > kcap:                                         # more segments than the recent batches had: the tables get their own block
            kcap = min(T, (k + 63) & ~63)
            t_seg, t_feat, t_need = al(0), al(B * kcap * 2 * 8), al(B * kcap * 2 * 8) + (al(B * kcap * D * 4) if want_f else 0)
            tl = self.out_pool.lease(t_need) if handed else None
            if tl is not None:
                towner, tblk = tl
            else:                                            # pageable mode, or the pool is exhausted: a private pinned bounce block
                pb = torch.empty(t_need, dtype=torch.uint8, pin_memory=True)
                towner, tblk = pb.numpy(), pb
            o_seg, o_feat = t_seg, t_feat
        if not self._overlap_d2h and want_h:
            blk[:B * T * D * 4].view(torch.float32).view(B, T, D).copy_(hidden, non_blocking=True)
        tblk[o_seg:o_seg + B * k * 2 * 8].view(torch.int64).view(B, k, 2).copy_(seg[:, :k], non_blocking=True)
        if want_f:
            tblk[o_feat:o_feat + B * k * D * 4].view(torch.float32).view(B, k, D).copy_(feats[:, :k], non_blocking=True)
        gmark("tables D2H done")
        cur.wait_stream(copy_s)
        cur.synchronize()
        mark("all D2H done")
        hidden_h = owner[:B * T * D * 4].view(np.float32).reshape(B, T, D) if want_h else None
        seg_h = towner[o_seg:o_seg + B * k * 2 * 8].view(np.int64).reshape(B, k, 2)
        feats_h = towner[o_feat:o_feat + B * k * D * 4].view(np.float32).reshape(B, k, D) if want_f else None
        outputs = []
        for i in range(B):
            n = int(nseg_h[i])
            segments = seg_h[i, :n].copy() if n > 0 else np.array([])
            o = {"segments": segments * 1.0 / FRAME_RATE if in_second else segments}
            if want_f:
                # (a view of the leased block, like hidden_states; a scratch block is reused by the next call, so its rows are copied)
                o["segment_features"] = (feats_h[i, :n] if handed else feats_h[i, :n].copy()) if n > 0 else np.array([])
            if want_h:
                o["hidden_states"] = hidden_h[i] if handed else hidden_h[i].copy()
            outputs.append(o)
        mark("dicts built")
        return outputs if is_batch else outputs[0]

    # -- one synchronous call on a large host batch, pipelined over sub-batches (round 6) -----------------
    def _split_plan(self, batch_wavs):
        """sub-batches (lists of the caller's tensors, in order) or None: only host batches large enough that a part still fills the chip
        (at least 8 rows and ~65 s of padded audio per part)"""
        n = self._call_split
        if n < 2 or len(batch_wavs) < 2:
            return None
        rows, lmax = 0, 0
        for w in batch_wavs:
            if not torch.is_tensor(w) or w.dim() != 2 or w.is_cuda:
                return None
            rows += int(w.shape[0])
            lmax = max(lmax, int(w.shape[1]))
        n = min(n, rows // 8, (rows * lmax) >> 20)
        if n < 2:
            return None
        parts, cur, acc, k = [], [], 0, 1
        for w in batch_wavs:
            cur.append(w)
            acc += int(w.shape[0])
            if acc * n >= rows * k and k < n:
                parts.append(cur)
                cur, k = [], k + 1
        if cur:
            parts.append(cur)
        return parts if len(parts) >= 2 else None

    def _call_in_parts(self, parts, in_second):
        self._force_lmax = max(int(w.shape[1]) for p in parts for w in p)
        outputs = []
        try:
            for res in self.stream(parts, in_second=in_second):
                outputs.extend(res)
        finally:
            self._force_lmax = 0
        return outputs

    # -- a stream of batches: the PCIe-inclusive path at (nearly) the resident rate -----------------------
    def stream(self, batches, in_second=True):
        """Generator over an iterable of batches (each what ``__call__`` takes as ``wav=``: a list of host ``[channels, N]`` tensors).
        Yields, in order, exactly what ``self(wav=batch, in_second=in_second)`` returns for each of them (same bits: the same
        kernels on the same data) -- but a synchronous call spends ~2.5 of its ~7.6 ms (32 x 10 s) outside the GPU's forward
        (padding + H2D in front, D2H + slicing behind), and here those run under the NEIGHBOURING batches' forwards: batch
        i + 1 is padded and uploaded on a copy stream while batch i computes, batch i's results leave on a second copy stream
        while batch i + 1 computes, and the host slices batch i - 1 meanwhile.  The reference has no counterpart (its
        ``__call__`` is one synchronous batch, sylber.py:76-138); a corpus loop over it is what this replaces."""
        it = iter(batches)
        dev = self.speech_model.device
        cur = torch.cuda.current_stream(dev)
        st = self.__dict__.get("_stream_streams")
        if st is None or st[0].device != dev:
            from .streams import concurrent_streams
            st = self._stream_streams = concurrent_streams(2, dev, avoid=[cur])        # H2D, D2H
        h2d, d2h = st
        want_h, want_f = self._want_hidden, self._want_feats
        counter = {"in": 0}
        NSET = 3                                              # device buffer sets = batches issued ahead + 1

        def ring_set(j):
            ring = self.__dict__.setdefault("_stream_dev", [None] * 8)
            if ring[j] is None:
                ring[j] = {"in_free": None, "out_free": None}
            return ring[j]

        def flat(d, key, numel, dtype):
            buf = d.get(key)
            if buf is None or buf.numel() < numel or buf.device != dev:
                if buf is not None:
                    torch.cuda.synchronize(dev)               # (grow-only: a larger batch shape than any before)
                buf = d[key] = torch.empty(int(numel * 1.25) + 64, dtype=dtype, device=dev)
            return buf[:numel]

        def issue_input(batch_wavs):
            rows, lengths = self._rows(batch_wavs if isinstance(batch_wavs, (list, tuple)) else [batch_wavs])
            if any(r.is_cuda for r in rows):
                raise ValueError("Segmenter.stream takes host tensors (device batches have nothing to overlap: use __call__)")
            lmax = max(max(lengths), int(self.__dict__.get("_force_lmax") or 0))      # (a split __call__ pads every part to the whole batch's max)
            stage, slot = self._stage_buffer((len(rows), lmax))
            stage_np = stage.numpy()

            def fill(lo, hi):
                for i in range(lo, hi):
                    src = rows[i].detach()
                    stage_np[i, : lengths[i]] = (src if src.dtype == torch.float32 else src.to(torch.float32)).numpy()
                    stage_np[i, lengths[i]:] = 0.0
            n = len(rows)
            if self._fill_threads > 1 and n >= 16:
                fut = self._fill_pool().submit(fill, 0, n // 2)
                fill(n // 2, n)
                fut.result()
            else:
                fill(0, n)
            tr = self.__dict__.get("_trace")
            if tr is not None:
                tr.append(("padded", time.perf_counter()))
            # device buffers come from a ring of three grow-only sets with explicit events, not from the caching allocator: blocks
            # handed between streams with record_stream come back late, and the hipMalloc that then happens every third batch
            # synchronises the device (measured: GPU gaps 5.3 / 5.3 / 6.9 ms, tools/api_stream_timeline.py)
            d = ring_set(counter["in"] % NSET)
            counter["in"] += 1
            with torch.cuda.stream(h2d):
                if d["in_free"] is not None:
                    h2d.wait_event(d["in_free"])
                batch = flat(d, "in", n * lmax, torch.float32).view(n, lmax)
                batch.copy_(stage, non_blocking=True)
                slot["event"].record(h2d)
                ev = torch.cuda.Event()
                ev.record(h2d)
            return {"batch": batch, "lengths": lengths, "uploaded": ev, "single": not isinstance(batch_wavs, (list, tuple)), "set": d}

        def issue_compute(t, slot_id):
            batch, lengths, d = t["batch"], t["lengths"], t["set"]
            cur.wait_event(t["uploaded"])
            if d["out_free"] is not None:
                cur.wait_event(d["out_free"])             # the copies of the batch that last used this set have left
            B_, T_ = batch.shape[0], self.speech_model.num_frames(batch.shape[1])
            # forward + boundary detection run on `cur`, the stream that was current when the generator STARTED, whatever the
            # caller's current stream is at this resumption: `done` below is recorded on the stream the work was issued to
            with torch.cuda.stream(cur):
                hidden = self.speech_model.forward(batch, lengths, out=flat(d, "hid", B_ * T_ * 768, torch.float32).view(B_, T_, 768))
                out = (flat(d, "seg", B_ * T_ * 2, torch.int64).view(B_, T_, 2), flat(d, "nseg", B_, torch.int32),
                       flat(d, "feat", B_ * T_ * 768, torch.float32).view(B_, T_, 768) if want_f else None)
                seg, nseg, feats = self.speech_model.segment(hidden, self.norm_threshold, self.merge_threshold, out=out)
            tr = self.__dict__.get("_trace")                  # tools/api_stream_timeline.py: (label, host time[, event]) marks
            done = torch.cuda.Event(enable_timing=tr is not None)
            done.record(cur)
            if tr is not None:
                tr.append(("compute issued", time.perf_counter(), done))
            B, T, D = hidden.shape

            def al(n):
                return (n + 255) & ~255
            kcap = min(T, self._kcap_seen)
            o_cnt = al(B * T * D * 4) if want_h else 0
            o_seg = o_cnt + al(B * 4)
            o_feat = o_seg + al(B * kcap * 2 * 8)
            need = o_feat + (al(B * kcap * D * 4) if want_f else 0)
            lease = self.out_pool.lease(need) if self.output_memory == "pinned" else None
            handed = lease is not None
            if handed:
                owner, blk = lease
            else:                                         # pageable mode / pool exhausted: one private bounce block per in-flight slot (three)
                ring = self.__dict__.setdefault("_stream_scratch", [None] * 8)
                if ring[slot_id] is None or ring[slot_id].numel() < need:
                    ring[slot_id] = torch.empty((need + (1 << 20) - 1) >> 20 << 20, dtype=torch.uint8, pin_memory=True)
                blk = ring[slot_id]
                owner = blk.numpy()
            with torch.cuda.stream(d2h):
                d2h.wait_event(done)
                if want_h:
                    blk[:B * T * D * 4].view(torch.float32).view(B, T, D).copy_(hidden, non_blocking=True)
                blk[o_cnt:o_cnt + B * 4].view(torch.int32).copy_(nseg, non_blocking=True)
                blk[o_seg:o_seg + B * kcap * 2 * 8].view(torch.int64).view(B, kcap, 2).copy_(seg[:, :kcap], non_blocking=True)
                if want_f:
                    blk[o_feat:o_feat + B * kcap * D * 4].view(torch.float32).view(B, kcap, D).copy_(feats[:, :kcap], non_blocking=True)
                out_ev = torch.cuda.Event()
                out_ev.record(d2h)
            d["in_free"], d["out_free"] = done, out_ev
            t.update(dict(shape=(B, T, D), kcap=kcap, offs=(o_cnt, o_seg, o_feat), owner=owner, handed=handed, out_ev=out_ev,
                          dev=(seg, feats)))
            t.pop("batch")
            return t

        def finish(t):
            tr = self.__dict__.get("_trace")
            if tr is not None:
                tr.append(("finish enter", time.perf_counter()))
            t["out_ev"].synchronize()
            if tr is not None:
                tr.append(("results on the host", time.perf_counter()))
            B, T, D = t["shape"]
            o_cnt, o_seg, o_feat = t["offs"]
            owner, handed, kslots = t["owner"], t["handed"], t["kcap"]
            nseg_h = owner[o_cnt:o_cnt + B * 4].view(np.int32).copy()
            k = max(int(nseg_h.max()) if B else 0, 1)
            towner = owner
            self._note_segments(k)
            if k > kslots:                                # more segments than the recent batches had: fetch the tables again (rare)
                seg, feats = t["dev"]
                kslots = min(T, (k + 63) & ~63)
                o_seg, o_feat = 0, (B * kslots * 2 * 8 + 255) & ~255
                pb = torch.empty(o_feat + (B * kslots * D * 4 if want_f else 0), dtype=torch.uint8, pin_memory=True)
                pb[o_seg:o_seg + B * kslots * 2 * 8].view(torch.int64).view(B, kslots, 2).copy_(seg[:, :kslots], non_blocking=True)
                if want_f:
                    pb[o_feat:o_feat + B * kslots * D * 4].view(torch.float32).view(B, kslots, D).copy_(feats[:, :kslots], non_blocking=True)
                torch.cuda.current_stream(dev).synchronize()
                towner, tcopy = pb.numpy(), False
            else:
                tcopy = not handed
            t.pop("dev")
            hidden_h = owner[:B * T * D * 4].view(np.float32).reshape(B, T, D) if want_h else None
            seg_h = towner[o_seg:o_seg + B * kslots * 2 * 8].view(np.int64).reshape(B, kslots, 2)
            feats_h = towner[o_feat:o_feat + B * kslots * D * 4].view(np.float32).reshape(B, kslots, D) if want_f else None
            outputs = []
            for i in range(B):
                n = int(nseg_h[i])
                segments = seg_h[i, :n].copy() if n > 0 else np.array([])
                o = {"segments": segments * 1.0 / FRAME_RATE if in_second else segments}
                if want_f:
                    o["segment_features"] = (feats_h[i, :n].copy() if tcopy else feats_h[i, :n]) if n > 0 else np.array([])
                if want_h:
                    o["hidden_states"] = hidden_h[i] if handed else hidden_h[i].copy()
                outputs.append(o)
            return outputs[0] if t["single"] else outputs

        try:
            first = next(it)
        except StopIteration:
            return
        # two batches are issued ahead of the one being handed out: the D2H of batch i - 1 takes most of batch i's forward (the
        # copies share the chip with 160-KiB-LDS GEMM workgroups), so waiting for it before ISSUING batch i + 1 left the GPU idle
        # now and then (tools/api_stream_timeline.py)
        # the three batches in flight hold a leased block each: they must not eat the consumer's `max_pinned_batches` budget, or every
        # third batch falls back to a pageable copy of its 49 MB of hidden states (a 5 ms host stall, seen as a 7 ms GPU gap)
        budget = self.out_pool.max_leased
        self.out_pool.max_leased = budget + NSET
        pending = []
        # (round 6) the loop makes a few hundred container objects per batch; the full garbage collection they trigger every few batches walks the
        # whole heap of the process (torch, numpy, ...) and stops this thread for 40-120 ms while the GPU drains.  gc.freeze() takes the long-lived
        # heap out of the collector's sight for the duration of the stream (profiles/r06_exchange.md)
        import gc as _gc
        frozen = bool(self.__dict__.get("_freeze_gc", True)) and _gc.isenabled()
        if frozen:
            _gc.freeze()                                                 # (no collect() first: a full collection of this heap is the 40-120 ms this avoids)
        try:
            nxt = issue_input(first)
            i = 0
            while nxt is not None:
                pending.append(issue_compute(nxt, i % NSET))
                try:
                    nxt = issue_input(next(it))           # padded + uploaded under the forward just issued
                except StopIteration:
                    nxt = None
                if len(pending) > NSET - 1:
                    yield finish(pending.pop(0))
                i += 1
            while pending:
                yield finish(pending.pop(0))
        finally:
            # an early exit (consumer stopped, or an exception): the batches still in flight own leased page-locked blocks that
            # the D2H stream may still be writing.  Their owners die with `pending`, the finalizers hand the blocks back, and the next
            # lease could write the same block from another stream -- so wait for those copies first
            for t in pending:
                ev = t.get("out_ev")
                if ev is not None:
                    ev.synchronize()
            pending.clear()
            if frozen:
                _gc.unfreeze()
            self.out_pool.max_leased = budget
            self.out_pool.trim()

    def _note_segments(self, k: int) -> None:
        """size the next leased block from the RECENT per-batch maximum (rounded up to 64, floor 128), not from an all-time one:
        one long-clip batch with many segments must not double every later block for good"""
        self._kcap_recent.append(int(k))
        self._kcap_seen = max(128, (max(self._kcap_recent) + 63) & ~63)

    def _nseg_pinned(self, B: int) -> torch.Tensor:
        buf = self.__dict__.get("_nseg_pin")
        if buf is None or buf.numel() < B:
            buf = torch.empty(max(B, 64), dtype=torch.int32, pin_memory=True)
            self._nseg_pin = buf
        return buf[:B]

    def _scratch_block(self, nbytes: int):
        """one private pinned block for the pageable-output mode (results are COPIED out of it, so it is reused)"""
        blk = self.__dict__.get("_scratch_pin")
        if blk is None or blk.numel() < nbytes:
            blk = torch.empty((nbytes + (1 << 20) - 1) >> 20 << 20, dtype=torch.uint8, pin_memory=True)
            self._scratch_pin = blk
        return blk.numpy(), blk
