"""CPU tier, world_size 2 over gloo: the utterance-sharding logic of sylber_amd/dist.py (scatter
to the global max length, per-rank compute, gather) reproduces the single-process result row for
row.  The per-rank engine here is the CPU oracle (tests may use it); on the GPU the engine is
HubertEncoderHIP and the backend is nccl (RCCL)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle import hubert_ref, segment_oracle
from oracle.segmenter_ref import SegmenterRef
from sylber_amd.synth import syllable_wave
from sylber_amd.weights import synthetic_state_dict


class OracleEngine:
    device = torch.device("cpu")

    def __init__(self, sd):
        self.sd = sd

    def num_frames(self, n):
        return hubert_ref.num_frames(n)

    def forward(self, wav, lengths):
        with torch.no_grad():
            return hubert_ref.forward(self.sd, wav, lengths, num_layers=2)["hidden"].contiguous()

    def segment(self, hidden, nt, mt):
        B, T, D = hidden.shape
        seg = torch.zeros(B, T, 2, dtype=torch.int64)
        nseg = torch.zeros(B, dtype=torch.int32)
        feats = torch.zeros(B, T, D)
        for b in range(B):
            s = segment_oracle.get_segment(hidden[b].numpy(), nt, mt).reshape(-1, 2)
            nseg[b] = len(s)
            if len(s):
                seg[b, : len(s)] = torch.from_numpy(s)
                feats[b, : len(s)] = torch.from_numpy(segment_oracle.mean_pool(hidden[b].numpy(), s))
        return seg, nseg, feats


LENS = [12000, 9000, 16000, 7000, 11000]   # odd count: exercises the padded tail block


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(2)
    from sylber_amd.dist import ShardedSegmenter
    sd = synthetic_state_dict(0, num_layers=2)
    S = ShardedSegmenter(OracleEngine(sd), norm_threshold=2.6, merge_threshold=0.8)
    wavs = [syllable_wave(n, 60 + i) for i, n in enumerate(LENS)] if rank == 0 else None
    out = S(wavs, in_second=False)
    # the overlapped stream of batches returns, batch by batch, exactly what the synchronous step returns
    batches = lens = [None, None, None]
    if rank == 0:
        sets = [LENS, LENS[1:4], LENS[::-1]]
        lens = [list(ls) for ls in sets]
        batches = []
        for j, ls in enumerate(sets):
            b = torch.zeros(len(ls), max(ls))
            for i, n in enumerate(ls):
                b[i, :n] = syllable_wave(n, 80 + 10 * j + i)[0]
            batches.append(b)
    sync = [S.step(b, l) for b, l in zip(batches, lens)]
    streamed = list(S.run_stream(batches, lens, max_segments=64))
    # a LIST of engines (what bench.py passes: one handle per batch in flight) takes the steps round-robin: same results
    S2 = ShardedSegmenter([S.engine, OracleEngine(sd)], norm_threshold=2.6, merge_threshold=0.8)
    streamed2 = list(S2.run_stream(batches, lens, max_segments=64))
    ok = True
    # counters of the stream of batches (what the N > 1 bench line reports): three steps, both directions counted
    st = S.stats
    ok &= st["steps"] >= 3 and st["wait_s"] >= 0.0 and st["gather_bytes"] > 0 and st["scatter_bytes"] > 0
    # per-rank ingest: every rank feeds its own block of each batch from host memory, no scatter; same results on root
    sets_all = [LENS, LENS[1:4], LENS[::-1]]
    shards = []
    for j, ls in enumerate(sets_all):
        bper = (len(ls) + world - 1) // world
        blk = torch.zeros(bper, max(ls))
        for i, n in enumerate(ls):
            if rank * bper <= i < (rank + 1) * bper:
                blk[i - rank * bper, :n] = syllable_wave(n, 80 + 10 * j + i)[0]
        shards.append(blk)
    S.reset_stats()
    streamed3 = list(S.run_stream(batches, lens, max_segments=64, ingest="per-rank", host_shards=shards))
    ok &= S.stats["scatter_bytes"] == 0 and S.stats["steps"] == 3
    if rank == 0:
        for b, c in zip(streamed, streamed3):
            ok &= all(torch.equal(x, y) for x, y in zip(b, c))
    # an utterance with more segments than the exchanged slots: ONE error, after the loop (no per-step host check)
    raised = False
    try:
        list(S.run_stream(batches, lens, max_segments=1))
    except RuntimeError as e:
        raised = "max_segments=1" in str(e)
    ok &= raised if rank == 0 else True
    # a consumer that stops early (every rank after the same step) still gets the overflow error, from close()
    raised = False
    g = S.run_stream(batches, lens, max_segments=1)
    next(g)
    try:
        g.close()
    except RuntimeError as e:
        raised = "max_segments=1" in str(e)
    ok &= raised if rank == 0 else True
    dist.barrier()
    if rank == 0:
        for b, c in zip(streamed, streamed2):
            ok &= all(torch.equal(x, y) for x, y in zip(b, c))
    else:
        assert all(x is None for x in streamed2)
    if rank == 0:
        for a, b in zip(sync, streamed):
            k = b[1].shape[1]
            ok &= torch.equal(a[0], b[0]) and torch.equal(a[2], b[2])
            n = int(a[2].max())
            ok &= torch.equal(a[1][:, :n], b[1][:, :n]) and torch.equal(a[3][:, :n], b[3][:, :n]) and n <= k
        q.put([(o["segments"], o["segment_features"], o["hidden_states"]) for o in out])
        q.put(bool(ok) and len(streamed) == 3)
    else:
        assert all(x is None for x in streamed)
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_sharding_matches_single_process():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = q.get(timeout=300)
    stream_ok = q.get(timeout=300)
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    assert stream_ok, "run_stream (prefetched scatter + asynchronous gather) differs from the synchronous step"
    sd = synthetic_state_dict(0, num_layers=2)
    ref = SegmenterRef(sd, encoding_layer=2)([syllable_wave(n, 60 + i) for i, n in enumerate(LENS)], in_second=False)
    assert len(got) == len(ref) == len(LENS)
    for (seg, feats, hid), r in zip(got, ref):
        assert hid.shape == r["hidden_states"].shape
        assert np.abs(hid - r["hidden_states"]).max() < 2e-5      # same fp32 CPU ops; batch split changes BLAS blocking
        assert seg.shape == r["segments"].shape and np.array_equal(seg, r["segments"])
        if len(seg):
            assert np.abs(feats - r["segment_features"]).max() < 2e-5


def test_single_process_degenerate_path():
    from sylber_amd.dist import ShardedSegmenter
    sd = synthetic_state_dict(0, num_layers=2)
    S = ShardedSegmenter(OracleEngine(sd))
    wavs = [syllable_wave(n, 60 + i) for i, n in enumerate(LENS[:2])]
    out = S(wavs, in_second=True)
    ref = SegmenterRef(sd, encoding_layer=2)(wavs, in_second=True)
    for o, r in zip(out, ref):
        assert np.array_equal(o["segments"], r["segments"])
        assert np.abs(o["hidden_states"] - r["hidden_states"]).max() < 2e-5
