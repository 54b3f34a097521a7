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
    # root's own share in place (round 6: root computes on a view of its chunk, its results are written straight into slot 0 of the gathered tensors, the
    # collectives' own-rank copies become out.copy_(out)) against the copying form of rounds 2-6a: same results, and the caller's batches are left untouched
    keep = [None if b is None else b.clone() for b in batches]
    S.inplace_root = False
    legacy = list(S.run_stream(batches, lens, max_segments=64))
    S.inplace_root = True
    if rank == 0:
        for a, b in zip(streamed, legacy):
            ok &= all(torch.equal(x, y) for x, y in zip(a, b))
        ok &= all(torch.equal(x, y) for x, y in zip(batches, keep))
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
    # a consumer that stops early (every rank after the same step) is told too -- by a RuntimeWarning: close() must end with
    # GeneratorExit, not with another exception (and one raised during garbage collection would only be printed)
    import warnings
    g = S.run_stream(batches, lens, max_segments=1)
    next(g)
    with warnings.catch_warnings(record=True) as caught:
        warnings.simplefilter("always")
        g.close()
    warned = any("max_segments=1" in str(w.message) and issubclass(w.category, RuntimeWarning) for w in caught)
    ok &= warned if rank == 0 else True
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


# ---- world size 4: odd ragged batches (a rank with an EMPTY real block, a padded tail), run_stream with two
# engines, gather="none" (results stay sharded), an early close() on every rank, then the communicator is used again
LENS4 = [[12000, 9000, 16000, 7000, 11000],          # 5 utterances over 4 ranks: Bper = 2, rank 2 holds one real row, rank 3 none
         [8000, 15000, 6000],                        # 3 over 4 ranks: Bper = 1, rank 3 holds only padding
         [10000, 10000, 9000, 12000, 7000, 14000, 9500]]


def _worker4(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(1)
    import warnings
    from sylber_amd.dist import ShardedSegmenter
    sd = synthetic_state_dict(0, num_layers=1)
    S = ShardedSegmenter([OracleEngine(sd), OracleEngine(sd)], norm_threshold=2.6, merge_threshold=0.8)
    for e in S.engines:                                  # (one encoder layer keeps the four single-thread ranks quick)
        e.forward = (lambda eng: (lambda wav, lengths: hubert_ref.forward(eng.sd, wav, lengths, num_layers=1)["hidden"].contiguous()))(e)
    batches = lens = [None] * len(LENS4)
    if rank == 0:
        lens = [list(ls) for ls in LENS4]
        batches = []
        for j, ls in enumerate(LENS4):
            b = torch.zeros(len(ls), max(ls))
            for i, n in enumerate(ls):
                b[i, :n] = syllable_wave(n, 200 + 10 * j + i)[0]
            batches.append(b)
    with torch.no_grad():
        rooted = list(S.run_stream(batches, lens, max_segments=64))
        local = list(S.run_stream(batches, lens, max_segments=64, gather="none"))
        # early close on every rank after the first batch, then the same communicator serves a synchronous step
        g = S.run_stream(batches, lens, max_segments=64)
        next(g)
        with warnings.catch_warnings():
            warnings.simplefilter("error")                 # no overflow here: close() must be silent
            g.close()
        again = S.step(batches[1], lens[1])
    # every rank ships its local rows to the parent, which stitches them and compares with root's gather
    npy = lambda out: tuple(t.numpy().copy() for t in out)       # (numpy: a torch tensor in the queue is a file descriptor of a process that exits)
    q.put((rank, [npy(out) for out in local], [npy(out) for out in rooted] if rank == 0 else None, npy(again) if rank == 0 else None))
    dist.barrier()
    dist.destroy_process_group()


def test_four_rank_sharding_ragged_odd_batches_and_local_results():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    W = 4
    procs = [ctx.Process(target=_worker4, args=(r, W, port, q)) for r in range(W)]
    for p in procs:
        p.start()
    got = {}
    for _ in range(W):
        rank, mine, rooted, again = q.get(timeout=600)
        got[rank] = (mine, rooted, again)
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    rooted, again = got[0][1], got[0][2]
    sd = synthetic_state_dict(0, num_layers=1)
    for j, ls in enumerate(LENS4):
        btot = len(ls)
        bper = (btot + W - 1) // W
        # gather="none": rank r yields rows [r * bper, min((r + 1) * bper, btot)) -- possibly none at all
        rows = [got[r][0][j] for r in range(W)]
        for r in range(W):
            assert rows[r][0].shape[0] == max(0, min(bper, btot - r * bper)), (j, r)
        for part in range(4):
            stitched = np.concatenate([rows[r][part] for r in range(W)], 0)
            assert np.array_equal(stitched, rooted[j][part]), (j, part)
        # and root's gather is the single-process result (same Lmax padding: sylber.py:93-97)
        ref = SegmenterRef(sd, encoding_layer=1)([syllable_wave(n, 200 + 10 * j + i) for i, n in enumerate(ls)], in_second=False)
        hid, seg, nseg, _ = rooted[j]
        for i, r_ in enumerate(ref):
            t = r_["hidden_states"].shape[0]
            assert np.abs(hid[i, :t] - r_["hidden_states"]).max() < 2e-5
            n = int(nseg[i])
            assert n == len(r_["segments"]) and (n == 0 or np.array_equal(seg[i, :n], r_["segments"]))
    assert np.array_equal(again[0], rooted[1][0]) and np.array_equal(again[2], rooted[1][2])
