"""GPU tier: ``ShardedSegmenter`` driving the HIP engine (world_size 1 on the one-GPU box) — the N > 1 control flow of
bench.py / sylber_amd/dist.py with the real kernels: ``step``, the pipelined ``run_stream`` (two handles in flight) and the
reference-shaped ``__call__`` agree BITWISE with the single-process ``Segmenter``; the same again with every exchange
forced through a one-rank RCCL communicator (scatter / gather / broadcast to self), which is the only RCCL
coverage a one-GPU box allows; and ``bench.py --gpus 2`` refuses to run on one GPU instead of reporting n_gpus: 1."""
import os
import socket
import subprocess
import sys

import numpy as np
import pytest
import torch

from sylber_amd.synth import syllable_wave
from sylber_amd.weights import synthetic_state_dict

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LENS = [24000, 16000, 31000, 9000, 20000]


@pytest.fixture(scope="module")
def sd():
    return synthetic_state_dict(0)


def _batches(lens_sets):
    out = []
    for j, ls in enumerate(lens_sets):
        b = torch.zeros(len(ls), max(ls))
        for i, n in enumerate(ls):
            b[i, :n] = syllable_wave(n, 40 + 10 * j + i)[0]
        out.append(b.cuda())
    return out


def _check_against_segmenter(S, sd):
    from sylber_amd import Segmenter
    wavs = [syllable_wave(n, 60 + i) for i, n in enumerate(LENS)]
    ref = Segmenter(model_ckpt=sd)(wav=wavs, in_second=False)
    got = S(wavs, in_second=False)
    assert len(got) == len(ref)
    for g, r in zip(got, ref):
        assert np.array_equal(g["hidden_states"], r["hidden_states"])
        assert g["segments"].shape == r["segments"].shape and np.array_equal(g["segments"], r["segments"])
        assert np.array_equal(g["segment_features"], r["segment_features"], equal_nan=True)
    # step() and the pipelined stream (two handles in flight) return the same tensors batch by batch
    sets = [LENS, LENS[1:4], LENS[::-1], LENS[:2]]
    batches = _batches(sets)
    sync = [S.step(b, ls) for b, ls in zip(batches, sets)]
    streamed = list(S.run_stream(batches, sets, max_segments=96))
    torch.cuda.synchronize()
    assert len(streamed) == len(sync)
    for a, b in zip(sync, streamed):
        assert torch.equal(a[0], b[0]) and torch.equal(a[2], b[2])
        for i, n in enumerate(a[2].tolist()):           # rows beyond an utterance's own count are uninitialised
            assert torch.equal(a[1][i, :n], b[1][i, :n]) and torch.equal(a[3][i, :n], b[3][i, :n])


def test_sharded_hip_engine_world1(sd):
    from sylber_amd import HubertEncoderHIP
    from sylber_amd.dist import ShardedSegmenter
    S = ShardedSegmenter([HubertEncoderHIP(sd), HubertEncoderHIP(sd)])
    _check_against_segmenter(S, sd)


def test_sharded_hip_engine_through_rccl_one_rank(sd):
    import torch.distributed as dist
    from sylber_amd import HubertEncoderHIP
    from sylber_amd.dist import ShardedSegmenter
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        S = ShardedSegmenter([HubertEncoderHIP(sd), HubertEncoderHIP(sd)], always_collective=True)
        assert S._coll
        _check_against_segmenter(S, sd)
        # round 6: run_stream's buffer ring (no allocation in steady state) with reuse_results=True -- the tensors root yields are valid
        # until 2 x engines - 1 further batches have been yielded, so a consumer that copies each batch out at once sees exactly what the
        # synchronous step returns, over more batches than the ring has sets, with batch shapes that grow and shrink
        sets = [LENS, LENS[1:4], LENS[::-1], LENS[:2], LENS, LENS[2:], LENS[::-1], LENS[:3], LENS, LENS[1:]]
        batches = _batches(sets)
        sync = [S.step(b, ls) for b, ls in zip(batches, sets)]
        got = []
        for out in S.run_stream(batches, sets, max_segments=96, reuse_results=True):
            got.append(tuple(t.clone() for t in out))
        torch.cuda.synchronize()
        assert len(got) == len(sync)
        for a, b in zip(sync, got):
            assert torch.equal(a[0], b[0]) and torch.equal(a[2], b[2])
            for i, n in enumerate(a[2].tolist()):
                assert torch.equal(a[1][i, :n], b[1][i, :n]) and torch.equal(a[3][i, :n], b[3][i, :n])
        m0 = torch.cuda.memory_stats()["num_device_alloc"]
        for out in S.run_stream(batches, sets, max_segments=96, reuse_results=True):
            pass
        torch.cuda.synchronize()
        assert torch.cuda.memory_stats()["num_device_alloc"] - m0 <= 2, "the stream of batches still allocates device memory in steady state"
    finally:
        dist.destroy_process_group()


def test_bench_refuses_more_ranks_than_gpus():
    if torch.cuda.device_count() >= 2:
        pytest.skip("needs a one-GPU box")
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                       capture_output=True, text=True, env=env, timeout=300)
    assert r.returncode != 0 and "GPU" in r.stderr and '"n_gpus"' not in r.stdout


def _two_rank_bench(extra):
    import json
    import socket
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ)
    env["SYLBER_DIST_BACKEND"] = "gloo"
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE"):
        env.pop(k, None)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1",
           "--batch", "4", "--clip-seconds", "2", "--no-cpu-baseline"] + list(extra)
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env, cwd=root)
    assert r.returncode == 0, r.stderr[-2000:]
    return json.loads(r.stdout.strip().splitlines()[-1])


def test_bench_two_rank_control_flow_on_one_gpu():
    """bench.py's N > 1 path end to end (launcher env, rank-0-only buffers, exchange as the default step with the
    resident-shard figure next to it, watchdog, ONE JSON line) with two ranks SHARING this box's GPU over gloo
    (SYLBER_DIST_BACKEND=gloo, a development aid: RCCL refuses two ranks on one device).  The numbers mean nothing; the
    control flow and the collectives' argument shapes on ranks 0 and 1 are what a one-GPU box cannot otherwise reach."""
    line = _two_rank_bench([])
    assert line["n_gpus"] == 2 and line["config"]["global_batch"] == 8
    assert "exchange_error" not in line
    assert "root scatter + gather" in line["config"]["parallelism"]
    assert line["resident_shards"]["value"] > 0 and line["value"] > 0
    assert "not a BASELINE.json configuration" in line["config"]["workload"]         # 4 x 2 s is no BASELINE config
    assert "2 s clips" in line["metric"]
    # what a first real multi-GPU run needs in order to explain itself
    d = line["exchange_detail"]
    assert d["ingest"] == "scatter"
    assert len(d["per_rank_ms_per_step"]) == 2 and all(x > 0 for x in d["per_rank_ms_per_step"])
    assert len(d["wait_ms_per_step_by_rank"]) == 2 and d["root_wait_ms_per_step"] >= 0
    assert d["scatter_bytes_per_step_root"] == 4 * 32000 * 4                         # one peer's 4 x 2 s block
    hid = 4 * 99 * 768 * 4
    assert d["gather_bytes_per_step_root"] >= hid                                    # at least the peer's hidden states
    assert len(line["resident_per_rank_ms_per_step"]) == 2


def test_bench_two_rank_per_rank_ingest():
    """--ingest per-rank: no scatter, every rank H2Ds its own page-locked shard; the gather is unchanged"""
    line = _two_rank_bench(["--ingest", "per-rank"])
    assert "exchange_error" not in line
    assert "per-rank H2D ingest" in line["config"]["parallelism"]
    d = line["exchange_detail"]
    assert d["ingest"] == "per-rank" and d["scatter_bytes_per_step_root"] == 0
    assert d["h2d_bytes_per_step_rank0"] == 4 * 32000 * 4
    assert d["gather_bytes_per_step_root"] > 0 and line["value"] > 0


def test_watchdog_line_is_complete_and_names_the_phase():
    """a hung exchange must still yield ONE contract-complete JSON line whose `exchange_error` says which step of the exchange
    rank 0 was in: the one-rank RCCL self-test with a zero-second watchdog trips it on purpose"""
    import json
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "NCCL_DEBUG")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--exchange-selftest", "--exchange-timeout", "0", "--steps", "2",
                        "--warmup", "1", "--batch", "4", "--clip-seconds", "2", "--no-cpu-baseline", "--no-api"],
                       capture_output=True, text=True, env=env, timeout=600, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    line = json.loads(r.stdout.strip().splitlines()[-1])
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
                "dtype", "data", "config"):
        assert key in line, key
    assert line["value"] > 0 and "workload" in line["config"]
    err = line["exchange_error"]
    assert "watchdog" in err and "rank 0 was in:" in err
    assert any(w in err for w in ("scatter", "gather", "compute", "idle")), err


def test_pipeline_streams_are_on_different_hardware_queues():
    """sylber_amd/streams.py: the streams the two-batches-in-flight pipelines use must run kernels concurrently (HIP streams that
    share a hardware queue run one after the other, whatever the program's dependencies say)"""
    import torch
    from sylber_amd.streams import concurrent_streams, serialised
    st = concurrent_streams(4, "cuda:0")
    assert len(st) == 4 and len({s.cuda_stream for s in st}) == 4
    assert not serialised(st[0], st[1])                      # the two compute streams
    assert sum(serialised(st[i], st[j]) for i in range(4) for j in range(i + 1, 4)) <= 1   # (4 hardware queues by default)
    assert serialised(st[0], st[0])                          # the probe itself: one stream against itself is serial


def test_concurrent_streams_are_leased_not_shared():
    """the probed set of independent streams is shared by the whole process, so two owners used to be handed the SAME streams and
    serialised against each other.  Streams are leased: a second caller gets streams nobody holds while there are any, a released stream goes
    out again first, and an index-less 'cuda' device means the current device."""
    import torch
    from sylber_amd import streams as S
    S._LEASES.clear()                                        # (earlier tests of this process hold leases they never return)
    a = S.concurrent_streams(2, "cuda")
    b = S.concurrent_streams(2, "cuda:0")
    ids = lambda st: {int(x.cuda_stream) for x in st}
    assert len(ids(a)) == 2 and len(ids(b)) == 2
    assert not (ids(a) & ids(b)), "two owners were handed the same streams while unused independent ones existed"
    S.release_streams(a)
    c = S.concurrent_streams(2, "cuda:0")
    assert ids(c) == ids(a), "released streams go out again before anything is shared"
    S.release_streams(b)
    S.release_streams(c)
