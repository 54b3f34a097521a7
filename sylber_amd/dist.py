"""Utterance-sharded Segmenter over one node: one process per GPU, torch.distributed (backend
"nccl" = RCCL over xGMI on ROCm; "gloo" in the CPU tests).

The reference has no inference-time multi-device code (the only parallelism is
Lightning DDP for training, train.py:93).  Utterances are independent units on this path
(GroupNorm is per (b,c), attention per b, get_segment per utterance, sylber.py:126); the only
batch-global quantity is the padded length Lmax (sylber.py:93-97), so every shard is padded to the
GLOBAL max length and then reproduces the single-process result row for row.

Exchange: root scatters contiguous row blocks of the padded waveform batch (+ lengths), every rank
runs forward + segmentation on its block, root gathers hidden states, segment tables and pooled
features.  No collective sits inside the compute path.
"""
from __future__ import annotations

import os

from typing import List, Optional, Sequence

import numpy as np
import torch
import torch.distributed as dist

FRAME_RATE = 50


class ShardedSegmenter:
    """``engine`` is a per-rank object with ``device``, ``num_frames(n)``, ``forward(wav, lengths)`` and
    ``segment(hidden, norm_thr, merge_thr)`` (sylber_amd.HubertEncoderHIP on the GPU).  A LIST of engines (independent
    handles = independent workspaces) lets ``run_stream`` keep that many shards in flight on this rank, each on its
    own HIP stream: the memory-bound phases of one batch then run under the MFMA phases of the other."""

    def __init__(self, engine, norm_threshold: float = 2.6, merge_threshold: float = 0.8, group=None,
                 always_collective: bool = False, segment_on_side_stream: bool = True, streams=None):
        self.engines = list(engine) if isinstance(engine, (list, tuple)) else [engine]
        if len(self.engines) >= 2:                       # run_stream keeps one batch in flight per engine: they share the chip
            for e_ in self.engines:
                if hasattr(e_, "set_batches_in_flight"):
                    e_.set_batches_in_flight(len(self.engines))
        self.engine = self.engines[0]
        self.norm_threshold = norm_threshold
        self.merge_threshold = merge_threshold
        self.group = group
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.device = self.engine.device
        # a one-rank group normally short-cuts the exchange; `always_collective` sends it through the communicator
        # anyway (scatter / gather to self) so that the RCCL code path can be exercised on a one-GPU box
        self._coll = self.world > 1 or (always_collective and dist.is_initialized())
        self._cuda = torch.device(self.device).type == "cuda"
        # streams that share a hardware queue run one after the other (sylber_amd/streams.py): probe for independent ones
        # `streams`: 2 x engines streams the caller already holds (bench.py passes the ones its resident steps ran on).  A process that leases
        # MORE independent streams than it needs fills every hardware queue (GPU_MAX_HW_QUEUES) with its own streams, and RCCL's internal
        # stream then has to share a queue with one of them: its copies, which wait for a batch's boundary detection, sit in front of the
        # other engine's next forward and the two batches in flight run one after the other (6-8 ms per step instead of 5:
        # profiles/r06_exchange.md)
        pool = None
        if self._cuda:
            if streams is not None:
                pool = list(streams)
                assert len(pool) >= 2 * len(self.engines), "streams= must hold 2 x engines streams"
            else:
                from .streams import concurrent_streams
                pool = concurrent_streams(2 * len(self.engines), self.device)
        self._streams = pool[:len(self.engines)] if self._cuda else None
        # boundary detection (one workgroup per utterance, ~0.25 ms of latency on 32 CUs) and the gather that follows it run
        # on a second stream per engine: the engine's next forward then does not queue behind them
        # (segment_on_side_stream=False keeps both on the engine stream: A/B switch)
        self._sides = (pool[len(self.engines):] if segment_on_side_stream else list(self._streams)) if self._cuda else None
        # the input exchange (scatter, or the per-rank H2D) is issued under a stream of its own, E batches ahead (run_stream): under the
        # engine's stream it waited for that engine's previous forward, and with a look-ahead of one it sat behind the previous gather of
        # the SAME engine in the communicator's single in-order stream -- forward(i + 2) started one gather + one scatter late (the 5 %
        # of the one-rank self-test, profiles/r06_exchange.md)
        self._ingest = torch.cuda.Stream(device=self.device) if self._cuda else None
        # defaults of run_stream (A/B: bench.py --exchange-lookahead / --exchange-ingest-stream; profiles/r06_exchange.md: on a one-rank
        # communicator the step time is dominated by RCCL's copy kernels displacing the 160-KiB-LDS GEMM workgroups and varies 13-45 % run to
        # run; E batches ahead under the engine's own stream had the best mean of the four combinations)
        self.lookahead = len(self.engines)
        self.ingest_stream = False
        self.side_delay = 0                       # A/B: iterations by which a batch's side-stream work is issued late (run_stream; no measured effect)
        # run_stream freezes the garbage collector's view of the heap while it runs (gc.freeze): the loop makes a few hundred container objects per
        # step, which triggers a full collection every few steps, and a full collection of a process that holds torch, numpy and the weights'
        # Python side stops the host thread for 40-120 ms -- the GPU drains and idles.  THAT was the "5-65 % overhead" of the one-rank self-test
        # that round 6 chased through allocators, streams and hardware queues (profiles/r06_exchange.md); frozen, a collection only walks what the
        # loop itself created
        self.freeze_gc = True
        # Round 6, second half: ROOT'S OWN SHARE does not go through the collectives' copies.  Root computes on a view of its chunk of the batch (that view is also
        # entry 0 of the scatter list) and its engine writes straight into slot 0 of the gathered tensors (that view is also the gather's input): the collective's
        # own-rank step is `out.copy_(in)` with out IS in -- a no-op -- instead of a 20 MB + 68 MB device-to-device copy per step beside the GEMMs.  Same protocol for
        # every other rank (still one scatter and four gathers per step).  False: rounds 2-6a (A/B switch; bench.py --exchange-root-copies).
        self.inplace_root = True
        self.reset_stats()

    def consumer_stream(self):
        """A stream for the CONSUMER of ``run_stream`` that shares its hardware queue with none of the pipeline's streams.  Every batch handed over
        makes the consumer's current stream wait for that batch's gather; if that stream sits on the hardware queue of an engine's stream (the
        process's default stream usually does: streams.py), the wait lands IN FRONT of that engine's next forward and the batches in flight run
        one after the other -- from one process start to the next the one-rank self-test ran at 5.1 or at 6-9 ms per step for exactly this
        reason (profiles/r06_exchange.md).  ``with torch.cuda.stream(S.consumer_stream()): for out in S.run_stream(...)``."""
        if not self._cuda:
            return None
        cs = self.__dict__.get("_consumer")
        if cs is None:
            from .streams import concurrent_streams
            cs = self._consumer = concurrent_streams(1, self.device, avoid=list(self._streams) + list(self._sides))[0]
        return cs

    def reset_stats(self) -> None:
        """counters of ``run_stream`` (what the N > 1 bench line reports): host seconds this rank spent blocked in the
        gather hand-over, bytes it sent / received through the communicator, H2D bytes of the per-rank ingest"""
        self.stats = {"steps": 0, "wait_s": 0.0, "scatter_bytes": 0, "gather_bytes": 0, "h2d_bytes": 0,
                      # host seconds run_stream spent ISSUING each phase (nothing in it waits for the GPU: a step whose issue time exceeds its GPU
                      # time is host-bound -- found in round 6: the one-rank self-test was)
                      "host_s": {"input": 0.0, "forward": 0.0, "segment": 0.0, "gather": 0.0, "collect": 0.0, "consumer": 0.0, "loop": 0.0}}
        # what this rank was doing last (a watchdog-tripped bench line names it: "which collective hung")
        self.phase = "idle"

    def _hand_over(self, tensors):
        """results were produced (and allocated) under an engine's side stream; the caller consumes them on ITS current
        stream: tell the caching allocator, or a dropped result's block could be reused by compute(i + E) while the
        caller's queued work still reads it"""
        if self._cuda:
            cur = torch.cuda.current_stream(self.device)
            for t in tensors:
                if t is not None and t.is_cuda:
                    t.record_stream(cur)
        return tensors

    # ---- device-level step (what bench.py times) -------------------------------------------------
    def scatter(self, batch_root: Optional[torch.Tensor], lengths_root: Optional[Sequence[int]]):
        """batch_root: [Btot, Lmax] on root (None elsewhere).  Returns this rank's block [Bper, Lmax],
        its lengths, and (Btot, Bper)."""
        W = self.world
        meta = torch.zeros(2, dtype=torch.int64, device=self.device)
        if self.rank == 0:
            meta[0], meta[1] = batch_root.shape[0], batch_root.shape[1]
        if self._coll:
            dist.broadcast(meta, src=0, group=self.group)
        btot, lmax = int(meta[0]), int(meta[1])
        bper = (btot + W - 1) // W
        lens = torch.full((bper * W,), lmax, dtype=torch.int32, device=self.device)
        if self.rank == 0 and lengths_root is not None:
            lens[:btot] = torch.as_tensor(list(lengths_root), dtype=torch.int32)
        if not self._coll:
            my_wav, my_lens = batch_root, lens
        else:
            my_lens = torch.empty(bper, dtype=torch.int32, device=self.device)
            my_wav = torch.empty(bper, lmax, dtype=torch.float32, device=self.device)
            if self.rank == 0:
                pad = bper * W - btot
                full = batch_root if pad == 0 else torch.cat(
                    [batch_root, torch.zeros(pad, lmax, dtype=torch.float32, device=self.device)], 0)
                wav_chunks = list(full.contiguous().view(W, bper, lmax).unbind(0))
                len_chunks = list(lens.view(W, bper).unbind(0))
            else:
                wav_chunks = len_chunks = None
            dist.scatter(my_lens, len_chunks, src=0, group=self.group)
            dist.scatter(my_wav, wav_chunks, src=0, group=self.group)
        return my_wav, my_lens, btot, bper

    def compute(self, my_wav: torch.Tensor, my_lens, k: int = 0):
        lens = [int(x) for x in (my_lens.tolist() if torch.is_tensor(my_lens) else my_lens)]
        eng = self.engines[k]
        hidden = eng.forward(my_wav, lens)
        seg, nseg, feats = eng.segment(hidden, self.norm_threshold, self.merge_threshold)
        return hidden, seg, nseg, feats

    def gather(self, hidden, seg, nseg, feats, btot: int):
        """Root receives [Btot, ...] tensors; other ranks receive None."""
        W = self.world
        if not self._coll:
            return hidden[:btot], seg[:btot], nseg[:btot], feats[:btot]

        def g(t):
            outs = [torch.empty_like(t) for _ in range(W)] if self.rank == 0 else None
            dist.gather(t.contiguous(), outs, dst=0, group=self.group)
            return torch.cat(outs, 0)[:btot] if self.rank == 0 else None

        # features are mostly empty rows: trim to the global max segment count first (tiny all-reduce)
        nmax = nseg.max().to(torch.int64).clone()
        dist.all_reduce(nmax, op=dist.ReduceOp.MAX, group=self.group)
        k = max(int(nmax), 1)
        return g(hidden), g(seg[:, :k]), g(nseg), g(feats[:, :k])

    def step(self, batch_root, lengths_root=None):
        my_wav, my_lens, btot, _ = self.scatter(batch_root, lengths_root)
        hidden, seg, nseg, feats = self.compute(my_wav, my_lens)
        return self.gather(hidden, seg, nseg, feats, btot)

    # ---- overlapped stream of batches --------------------------------------------------------------
    def full_results(self, bper: int, T: int, max_segments: int, ring_set=None, rbuf=None, ring_results: bool = False):
        """root: the four gathered tensors of one step, allocated BEFORE its forward so that the engine can write root's own rows into slot 0
        (hidden states [W * bper, T, 768], tables [W * bper, k, 2], counts [W * bper], pooled features [W * bper, k, 768])"""
        W, k = self.world, max(1, min(int(max_segments), T))
        shapes = (((W * bper, T, 768), torch.float32), ((W * bper, k, 2), torch.int64), ((W * bper,), torch.int32), ((W * bper, k, 768), torch.float32))
        if ring_results and ring_set is not None and rbuf is not None:
            return [rbuf(ring_set, "full%d" % j, shp, dt) for j, (shp, dt) in enumerate(shapes)]
        return [torch.empty(shp, dtype=dt, device=self.device) for shp, dt in shapes]

    def gather_async(self, hidden, seg, nseg, feats, btot: int, max_segments: int, check: bool = True, ring_set=None, rbuf=None,
                     ring_results: bool = False, fulls_pre=None):
        """Like ``gather`` but with asynchronous collectives and WITHOUT the host round trip that trims the pooled
        features to the global max segment count: the first ``max_segments`` slots are exchanged instead.  Returns a
        zero-argument ``wait`` function.  An utterance with more segments than that is an error: with ``check`` the
        ``wait`` raises (one blocking D2H read on root); ``run_stream`` passes ``check=False`` and tests the largest
        count ONCE after its loop (``wait.nmax`` = device scalar), so that no step contains a host synchronisation."""
        W = self.world
        k = max(1, min(int(max_segments), seg.shape[1]))
        bper_ = hidden.shape[0]
        if fulls_pre is not None:
            # root, in place: `hidden` IS slot 0 of the gathered hidden states (the engine wrote it there); the packed tables / counts / features are written
            # into slot 0 of theirs; each gather's input is then the very tensor that is entry 0 of its output list
            assert self.rank == 0 and hidden.data_ptr() == fulls_pre[0].data_ptr()
            seg_k, nseg_v, feats_k = fulls_pre[1][:bper_], fulls_pre[2][:bper_], fulls_pre[3][:bper_]
            seg_k.copy_(seg[:, :k])
            if nseg.data_ptr() != nseg_v.data_ptr():
                nseg_v.copy_(nseg)
            feats_k.copy_(feats[:, :k])
            parts = [hidden, seg_k, nseg_v, feats_k]
        elif ring_set is not None and rbuf is not None:
            # the packed [:, :k] copies go into the set's own buffers (run_stream's ring: no allocation in steady state)
            seg_k = rbuf(ring_set, "seg_k", (seg.shape[0], k, 2), seg.dtype)
            seg_k.copy_(seg[:, :k])
            feats_k = rbuf(ring_set, "feat_k", (feats.shape[0], k, feats.shape[2]), feats.dtype)
            feats_k.copy_(feats[:, :k])
            parts = [hidden, seg_k, nseg, feats_k]
        else:
            parts = [hidden.contiguous(), seg[:, :k].contiguous(), nseg.contiguous(), feats[:, :k].contiguous()]

        def too_many(n):
            return RuntimeError("an utterance has %d segments, more than max_segments=%d" % (n, k))
        if not self._coll:
            # the reduction is enqueued on the producing (engine) stream BEFORE the event, so whoever waits for `done` also
            # sees nmax written; it is handed over like the results (record_stream), or its block could be reused early
            nmax = nseg[:btot].max()
            done = None
            if self._cuda:
                done = torch.cuda.Event()
                done.record(torch.cuda.current_stream(self.device))

            def wait1():
                if done is not None:
                    torch.cuda.current_stream(self.device).wait_event(done)
                self._hand_over((nmax,))
                if check and int(nmax) > k:
                    raise too_many(int(nmax))
                return self._hand_over(tuple(t[:btot] for t in parts))
            wait1.nmax = nmax
            return wait1
        fulls, works = [], []
        for j_, t in enumerate(parts):
            # root receives straight into the slices of ONE [W * Bper, ...] tensor: no concatenation copy afterwards
            if self.rank != 0:
                full = None
            elif fulls_pre is not None:
                full = fulls_pre[j_]
            elif ring_results and ring_set is not None and rbuf is not None:
                full = rbuf(ring_set, "full%d" % j_, (W * t.shape[0],) + tuple(t.shape[1:]), t.dtype)
            else:
                full = torch.empty((W * t.shape[0],) + tuple(t.shape[1:]), dtype=t.dtype, device=t.device)
            o = list(full.view((W, t.shape[0]) + tuple(t.shape[1:])).unbind(0)) if self.rank == 0 else None
            if fulls_pre is not None:
                o[0] = t                                         # the same tensor object: the collective's own-rank copy is out.copy_(out)
            works.append(dist.gather(t, o, dst=0, group=self.group, async_op=True))
            fulls.append(full)
            nb = t.numel() * t.element_size()
            self.stats["gather_bytes"] += nb * (W - 1) if self.rank == 0 else nb

        def wait():
            for wk in works:
                wk.wait()
            if self.rank != 0:
                return None
            res = tuple(f[:btot] for f in fulls)
            wait.nmax = res[2].max()
            if check and int(wait.nmax) > k:
                raise too_many(int(wait.nmax))
            return self._hand_over(res)
        wait.nmax = None
        _keep = parts                                      # the sources must outlive the collectives
        wait.keep = _keep
        wait.works = works                                 # (run_stream's ring: the next writer of these source buffers waits for them)
        return wait

    def keep_local(self, hidden, seg, nseg, feats, mine: int, max_segments: int):
        """``gather="none"``: this rank's results stay on this rank (a corpus job whose ranks write their own shards: nothing but the
        input scatter crosses xGMI).  Same hand-over protocol as ``gather_async``: a zero-argument ``wait`` with ``wait.nmax``."""
        k = max(1, min(int(max_segments), seg.shape[1]))
        parts = (hidden[:mine], seg[:mine, :k], nseg[:mine], feats[:mine, :k])
        nmax = nseg[:mine].max() if mine > 0 else None
        done = None
        if self._cuda:
            done = torch.cuda.Event()
            done.record(torch.cuda.current_stream(self.device))

        def wait():
            if done is not None:
                torch.cuda.current_stream(self.device).wait_event(done)
            if nmax is not None:
                self._hand_over((nmax,))
            return self._hand_over(parts)
        wait.nmax = nmax
        return wait

    def run_stream(self, batches_root, lengths_root=None, max_segments: int = 128, ingest: str = "scatter",
                   host_shards=None, gather: str = "root", lookahead: Optional[int] = None, ingest_stream: Optional[bool] = None,
                   reuse_results: bool = False):
        """Generator over a sequence of root batches (``[Btot, Lmax]`` tensors on root, ``None`` elsewhere; every rank
        must pass a sequence of the same length).  Software pipeline over ONE communicator, whose collectives
        execute in issue order: the scatter of batch i+1 is issued BEFORE the compute of batch i, and the gather of
        batch i after it, asynchronously — so gather(i) travels over xGMI while compute(i+1) runs, and scatter(i+2)
        queues behind gather(i) without anybody waiting for it yet.  All shapes and lengths are broadcast ONCE up
        front, so no step contains a host round trip.  Yields the gathered result of each batch in order (root:
        tensors, other ranks: None).

        ``ingest="per-rank"`` (inputs that originate on the host): no scatter; ``host_shards[i]`` is THIS
        rank's ``[Bper, Lmax]`` block of batch i in page-locked host memory and crosses this GPU's own PCIe link
        (asynchronous H2D on the consuming engine's stream); ``batches_root[i]`` then only supplies the shape on root.
        The gather is unchanged.  ``gather="none"``: no gather at all -- EVERY rank yields its own rows
        ``[rank * Bper, min((rank + 1) * Bper, Btot))`` of each batch (``hidden, seg[:, :max_segments], nseg, feats[:, :max_segments]``)
        and checks its own overflow; the deployment of a corpus job whose ranks write their own shards, and the upper bound of what
        the root gather can reach.  The overflow test of ``max_segments`` runs once, after the last step: a batch yielded
        ``reuse_results=True`` (a consumer that is done with a batch's tensors before it asks for the fourth batch after it -- a corpus
        loop that writes each result out, ``bench.py``): the gathered tensors root yields come from the same ring as the step's other buffers
        and are valid until ``2 x engines - 1`` further batches have been yielded (GPU-side: the gather that overwrites them waits for the
        work the consumer had queued on its stream when it asked for the next batch).  Default False: fresh tensors every step (the caching
        allocator then has to find ~68 MB per step that another stream just released, and calls hipMalloc -- a device synchronisation -- about
        once per step: profiles/r06_exchange.md).
        EARLIER may therefore carry a table truncated to ``max_segments`` rows (its ``nseg`` row still holds the true count,
        so ``nseg[i] > max_segments`` identifies it); the error is raised when the generator is exhausted; ``close()`` /
        garbage collection of a generator the consumer abandoned early issue a ``RuntimeWarning`` instead (``GeneratorExit`` path below)."""
        if ingest not in ("scatter", "per-rank"):
            raise ValueError("ingest must be 'scatter' or 'per-rank'")
        if gather not in ("root", "none"):
            raise ValueError("gather must be 'root' (results collected on rank 0) or 'none' (every rank yields its own block of rows)")
        batches = list(batches_root)
        n = len(batches)
        if n == 0:
            return
        W = self.world
        # ---- one-off: (Btot, Lmax) of every batch and all lengths, to every rank
        shapes = torch.zeros(n, 2, dtype=torch.int64, device=self.device)
        if self.rank == 0:
            for i, b in enumerate(batches):
                shapes[i, 0], shapes[i, 1] = b.shape[0], b.shape[1]
        if self._coll:
            dist.broadcast(shapes, src=0, group=self.group)
        shapes_h = shapes.tolist()
        bmax = max(int(s_[0]) for s_ in shapes_h)
        lens_all = torch.zeros(n, bmax, dtype=torch.int64, device=self.device)
        if self.rank == 0:
            for i, (bt, lm) in enumerate(shapes_h):
                src = lengths_root[i] if (lengths_root is not None and lengths_root[i] is not None) else [lm] * bt
                lens_all[i, :bt] = torch.as_tensor(list(src), dtype=torch.int64)
        if self._coll:
            dist.broadcast(lens_all, src=0, group=self.group)
        lens_h = lens_all.tolist()

        # Device buffers of the steps come from a RING of grow-only sets, not from the caching allocator (round 6): a step allocated ~190 MB
        # (input block, hidden states, tables, pooled features, their packed copies) under two or three streams and handed them to RCCL's
        # stream; blocks shared between streams come back late, so the allocator kept calling hipMalloc in steady state -- 2.4 times per
        # step, each one a device synchronisation: THAT was the 5-65 % "cost of the exchange" of the one-rank self-test
        # (profiles/r06_exchange.md).  Set (i // E) % RING_DEPTH of engine i % E serves batch i; before a set is written again the engine's
        # stream waits for the gather that last read it.  Only where results are COPIED out before they reach the caller (root gather over a
        # communicator, engines that take out=): with gather="none" or a world of one the caller receives the engine's own tensors.
        E_ = len(self.engines)
        RING_DEPTH = 2
        use_ring = (self._cuda and self._coll and gather == "root"
                    and all(getattr(e_, "supports_out", False) for e_ in self.engines))
        ring = self.__dict__.setdefault("_ring", {}) if use_ring else None

        def ring_set(i):
            return ring.setdefault((i % E_, (i // E_) % RING_DEPTH), {"works": None})
        # root's own share in place (self.inplace_root): the gathered tensors of a step exist before its forward
        inplace = bool(self.inplace_root) and self.rank == 0 and self._coll and gather == "root"
        pre_by_step = {}

        def rbuf(d, key, shape, dtype):
            numel = 1
            for x in shape:
                numel *= int(x)
            b = d.get(key)
            if b is None or b.numel() < numel or b.dtype != dtype:
                if b is not None:
                    torch.cuda.synchronize(self.device)          # (grow-only: a larger batch shape than any before)
                b = d[key] = torch.empty(int(numel * 1.1) + 64, dtype=dtype, device=self.device)
            return b[:numel].view(*shape)

        def scatter_known(i):
            btot, lmax = int(shapes_h[i][0]), int(shapes_h[i][1])
            bper = (btot + W - 1) // W
            mine = (lens_h[i][:btot] + [lmax] * (bper * W - btot))[self.rank * bper:(self.rank + 1) * bper]
            if ingest == "per-rank":
                src = host_shards[i]
                my_wav = rbuf(ring_set(i), "wav", (bper, lmax), torch.float32) if use_ring else torch.empty(bper, lmax, dtype=torch.float32, device=self.device)
                my_wav.copy_(src, non_blocking=True)
                self.stats["h2d_bytes"] += my_wav.numel() * 4
                return my_wav, mine, btot
            if not self._coll:
                return batches[i], mine, btot
            my_wav = rbuf(ring_set(i), "wav", (bper, lmax), torch.float32) if use_ring else torch.empty(bper, lmax, dtype=torch.float32, device=self.device)
            self.stats["scatter_bytes"] += my_wav.numel() * 4 * ((W - 1) if self.rank == 0 else 1)
            chunks = None
            if self.rank == 0:
                pad = bper * W - btot
                full = batches[i] if pad == 0 else torch.cat(
                    [batches[i], torch.zeros(pad, lmax, dtype=torch.float32, device=self.device)], 0)
                chunks = list(full.contiguous().view(W, bper, lmax).unbind(0))
            if self.rank == 0 and self.inplace_root:
                my_wav = chunks[0]                               # the scatter's own-rank copy becomes out.copy_(out); the engine reads the caller's rows
            self.phase = "scatter of batch %d (root -> ranks, %d bytes per rank)" % (i, my_wav.numel() * 4)
            dist.scatter(my_wav, chunks, src=0, group=self.group)
            return my_wav, mine, btot

        # step i runs on engine / stream i % E.  Collectives take their stream dependencies from the stream that is
        # current when they are issued, so scatter(i+1) is issued under the stream that will consume it and gather(i)
        # under the stream that produced its operands.
        import contextlib
        E = len(self.engines)

        def on(k):
            return torch.cuda.stream(self._streams[k]) if self._cuda else contextlib.nullcontext()

        def on_side(k):
            return torch.cuda.stream(self._sides[k]) if self._cuda else contextlib.nullcontext()

        # A/B switches (bench.py --exchange-lookahead / --exchange-ingest-stream): lookahead = how many batches ahead the input exchange is issued
        # (1 = round 5: scatter(i + 1) before compute(i)), ingest_stream = under its own stream instead of the consuming engine's
        LA = max(1, int(self.lookahead if lookahead is None else lookahead))
        own_stream = self.ingest_stream if ingest_stream is None else bool(ingest_stream)

        def on_ingest(i=0):
            if not self._cuda:
                return contextlib.nullcontext()
            return torch.cuda.stream(self._ingest if own_stream else self._streams[i % E])

        if self._cuda:
            cur = torch.cuda.current_stream(self.device)
            for st in self._streams + [self._ingest]:
                st.wait_stream(cur)                                      # the root batches were produced on `cur`
        import time as _time

        hs = self.stats["host_s"]

        def issue_input(i):
            """input exchange of batch i under the ingest stream; -> (my_wav, my_lens, btot, event the consuming engine stream waits for)"""
            t_h = _time.perf_counter()
            try:
                return _issue_input(i)
            finally:
                hs["input"] += _time.perf_counter() - t_h

        def _issue_input(i):
            with on_ingest(i):
                w_, l_, b_ = scatter_known(i)
                ev = None
                if self._cuda and own_stream:
                    ev = torch.cuda.Event()
                    ev.record(self._ingest)
            return w_, l_, b_, ev
        ahead = [issue_input(i) for i in range(min(LA, n))]                   # LA batches ahead (E: the scatter of a batch never queues behind
        pending = None                                                   # the gather of its own engine's previous batch
        worst = None                                                     # device scalar: largest segment count seen (root)

        def collect(wait_fn):
            nonlocal worst
            t0 = _time.perf_counter()
            self.phase = "wait for the gather of batch %d (ranks -> root: hidden states, segment tables, counts, features)" % self.stats["steps"]
            out = wait_fn()
            self.stats["wait_s"] += _time.perf_counter() - t0
            self.stats["steps"] += 1
            if wait_fn.nmax is not None:
                worst = wait_fn.nmax if worst is None else torch.maximum(worst, wait_fn.nmax)
            return out
        def overflow_check():
            # the ONE host synchronisation of the stream of batches: did any utterance overflow the exchanged slots?
            k = max(1, min(int(max_segments), 1 << 30))
            if worst is not None and int(worst) > k:
                raise RuntimeError("an utterance has %d segments, more than max_segments=%d" % (int(worst), k))
        # (A/B switch, no measured effect: the side-stream work of batch i issued DELAY iterations late)
        DELAY = max(0, int(self.side_delay)) if self._cuda else 0
        deferred = []

        def issue_side(i, k, hidden, ready, rs, btot):
            eng = self.engines[k]
            t_h = _time.perf_counter()
            with on_side(k):
                if ready is not None:
                    self._sides[k].wait_event(ready)
                    hidden.record_stream(self._sides[k])             # allocated under the engine stream, read here
                self.phase = "segmentation of batch %d (engine %d, side stream)" % (i, k)
                if rs is not None:
                    B_, T_ = hidden.shape[0], hidden.shape[1]
                    seg, nseg, feats = eng.segment(hidden, self.norm_threshold, self.merge_threshold,
                                                   out=(rbuf(rs, "seg", (B_, T_, 2), torch.int64), rbuf(rs, "nseg", (B_,), torch.int32),
                                                        rbuf(rs, "feat", (B_, T_, 768), torch.float32)))
                else:
                    seg, nseg, feats = eng.segment(hidden, self.norm_threshold, self.merge_threshold)
                self.phase = "issue of the asynchronous gather of batch %d" % i
                hs["segment"] += _time.perf_counter() - t_h
                t_h = _time.perf_counter()
                if gather == "none":
                    bper_ = hidden.shape[0]
                    wait = self.keep_local(hidden, seg, nseg, feats, max(0, min(bper_, btot - self.rank * bper_)), max_segments)
                else:
                    if rs is not None and reuse_results and rs.get("consumed") is not None:
                        self._sides[k].wait_event(rs["consumed"])   # the caller's queued work on the results this gather overwrites
                    wait = self.gather_async(hidden, seg, nseg, feats, btot, max_segments, check=False, ring_set=rs, rbuf=rbuf if rs is not None else None,
                                             ring_results=reuse_results, fulls_pre=pre_by_step.pop(i, None))
                    if rs is not None:
                        rs["works"] = getattr(wait, "works", None)
                        wait.ring_set = rs
            hs["gather"] += _time.perf_counter() - t_h
            return wait

        import gc as _gc
        frozen = False
        if self.freeze_gc and _gc.isenabled():
            _gc.freeze()          # (no collect() first: a full collection of this heap IS the 40-120 ms this avoids); the long-lived heap leaves the collector's sight until the stream ends
            frozen = True
        t_loop = _time.perf_counter()
        try:
            for step in range(n + DELAY):
                hs["loop"] = _time.perf_counter() - t_loop               # (host time of the steps issued so far, the consumer's share included)
                if step < n:
                    i = step
                    k = i % E
                    my_wav, my_lens, btot, arrived = ahead.pop(0)
                    if i + LA < n:
                        ahead.append(issue_input(i + LA))                     # prefetch: issued before compute(i), i.e. before gather(i)
                    t_h = _time.perf_counter()
                    with on(k):
                        if arrived is not None:
                            self._streams[k].wait_event(arrived)
                            if my_wav.is_cuda:
                                my_wav.record_stream(self._streams[k])       # allocated under the ingest stream, read here
                        self.phase = "compute of batch %d (forward, engine %d)" % (i, k)
                        eng = self.engines[k]
                        rs = None
                        if use_ring:
                            rs = ring_set(i)
                            if rs["works"] is not None:                      # the gather that last read this set's buffers (RING_DEPTH x E batches ago)
                                for wk in rs["works"]:
                                    wk.wait()                                # (stream-side wait of the engine's stream; long done in steady state)
                                rs["works"] = None
                            T_ = eng.num_frames(my_wav.shape[1])
                            fp_ = None
                            if inplace:
                                if reuse_results and rs.get("consumed") is not None:
                                    self._streams[k].wait_event(rs["consumed"])     # the caller's queued work on the results this forward overwrites
                                fp_ = self.full_results(my_wav.shape[0], T_, max_segments, rs, rbuf, reuse_results)
                                hidden = eng.forward(my_wav, [int(x) for x in my_lens], out=fp_[0][:my_wav.shape[0]])
                            else:
                                hidden = eng.forward(my_wav, [int(x) for x in my_lens], out=rbuf(rs, "hid", (my_wav.shape[0], T_, 768), torch.float32))
                            pre_by_step[i] = fp_
                        else:
                            hidden = eng.forward(my_wav, [int(x) for x in my_lens])
                            if inplace:                                      # an engine without out= (the CPU engines of the gloo tests): one copy into slot 0
                                fp_ = self.full_results(hidden.shape[0], hidden.shape[1], max_segments)
                                fp_[0][:hidden.shape[0]].copy_(hidden)
                                hidden = fp_[0][:hidden.shape[0]]
                                pre_by_step[i] = fp_
                        ready = None
                        if self._cuda:
                            ready = torch.cuda.Event()
                            ready.record(self._streams[k])
                    hs["forward"] += _time.perf_counter() - t_h
                    deferred.append((i, k, hidden, ready, rs, btot))
                if step < DELAY:
                    continue
                wait = issue_side(*deferred.pop(0))
                if pending is not None:
                    yielded = pending
                    t_h = _time.perf_counter()
                    res_ = collect(pending)
                    hs["collect"] += _time.perf_counter() - t_h
                    t_h = _time.perf_counter()
                    yield res_
                    hs["consumer"] += _time.perf_counter() - t_h           # (what the caller did between two batches)
                    # (resumed: whatever the consumer does with that batch is queued on its stream by now)
                    if use_ring and reuse_results and getattr(yielded, "ring_set", None) is not None:
                        ev_c = torch.cuda.Event()
                        ev_c.record(torch.cuda.current_stream(self.device))
                        yielded.ring_set["consumed"] = ev_c
                pending = wait
            last = collect(pending)
        except GeneratorExit:
            if frozen:
                _gc.unfreeze()
                frozen = False
            # the consumer stopped early (close(), garbage collection, interpreter shutdown): re-join the engine and side streams to
            # the caller's stream, and check the batches it already holds -- but only WARN: an exception raised from here would
            # replace GeneratorExit on close() and be printed as "Exception ignored" everywhere else.  A truncated
            # table is still identifiable afterwards: its `nseg` row holds the true count (> max_segments)
            if self._cuda:
                for st in self._streams + self._sides + [self._ingest]:
                    torch.cuda.current_stream(self.device).wait_stream(st)
            try:
                overflow_check()
            except RuntimeError as e:
                import warnings
                warnings.warn("ShardedSegmenter.run_stream abandoned early: %s (tables yielded so far may be truncated to max_segments rows)" % e,
                              RuntimeWarning, stacklevel=2)
            raise
        if frozen:
            _gc.unfreeze()
        if self._cuda:
            for st in self._streams + self._sides + [self._ingest]:
                torch.cuda.current_stream(self.device).wait_stream(st)
        overflow_check()
        yield last

    # ---- reference-shaped API on root ------------------------------------------------------------
    def __call__(self, wav: Optional[List[torch.Tensor]] = None, in_second: bool = True):
        """``wav``: list of [1, N] tensors on root (ignored elsewhere).  Root returns the same list of
        dicts the single-process Segmenter returns (sylber.py:128-138); other ranks return None."""
        batch = lengths = None
        if self.rank == 0:
            lengths = [int(w.shape[1]) for w in wav]
            batch = torch.zeros(len(wav), max(lengths), dtype=torch.float32, device=self.device)
            for i, w in enumerate(wav):
                batch[i, : lengths[i]] = w[0].to(self.device)
        out = self.step(batch, lengths)
        if self.rank != 0:
            return None
        hidden, seg, nseg, feats = (t.cpu().numpy() for t in out)
        res = []
        for i in range(hidden.shape[0]):
            n = int(nseg[i])
            segments = seg[i, :n].copy() if n > 0 else np.array([])
            res.append({"segments": segments * 1.0 / FRAME_RATE if in_second else segments,
                        "segment_features": feats[i, :n].copy() if n > 0 else np.array([]),
                        "hidden_states": hidden[i]})
        return res
