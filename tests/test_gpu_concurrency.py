"""GPU tier: two handles in flight on two HIP streams (what bench.py and ShardedSegmenter.run_stream do) must return,
bit for bit, what each handle returns alone.

Round 1 never checked this and was wrong under concurrency twice over: (1) the frame counts travelled through a pageable
hipMemcpyAsync from a stack buffer; (2) packed-fp32 VALU chains (v_pk_fma_f32) in conv0 + GroupNorm + GELU returned wrong
values in lanes 48-63 whenever MFMA waves of the OTHER handle's 128x128 GEMM shared the SIMD
(profiles/r02_packed_f32_hazard.md).  Small batches matter: their kernels leave CUs free, so kernels of the two
streams really are co-resident."""
import ctypes
import threading

import pytest
import torch

from sylber_amd.synth import noise_batch, syllable_wave
from sylber_amd.weights import synthetic_state_dict

pytestmark = pytest.mark.gpu
LENS = [24000, 16000, 31000, 9000, 20000]


def _batch(ls, seed):
    b = torch.zeros(len(ls), max(ls))
    for i, n in enumerate(ls):
        b[i, :n] = syllable_wave(n, seed + i)[0]
    return b.cuda()


@pytest.fixture(scope="module")
def engines():
    from sylber_amd import HubertEncoderHIP
    sd = synthetic_state_dict(0)
    return [HubertEncoderHIP(sd), HubertEncoderHIP(sd)]


@pytest.mark.parametrize("stage", [0, 1, 2, 3, 7])
def test_two_handles_in_flight_are_bitwise_sequential(engines, stage):
    S = [torch.cuda.Stream(), torch.cuda.Stream()]
    cases = [(_batch(LENS, 40), LENS), (_batch(LENS[1:4], 50), LENS[1:4]), (noise_batch(32, 160000, 3).cuda(), None)]
    for ci, (x, ls) in enumerate(cases):
        ref = [e.forward(x, ls, stop_stage=stage).clone() for e in engines]
        torch.cuda.synchronize()
        assert torch.equal(ref[0], ref[1])
        for it in range(10 if ci < 2 else 4):
            outs = []
            for k in ((0, 1) if it % 2 == 0 else (1, 0)):
                with torch.cuda.stream(S[k]):
                    outs.append(engines[k].forward(x, ls, stop_stage=stage))
            torch.cuda.synchronize()
            for o in outs:
                assert torch.equal(o, ref[0]), (stage, ci, it, float((o - ref[0]).abs().max()))


def test_conv_frontend_beside_the_128x128_gemm(engines):
    """the reproducer of the packed-fp32 fault: the conv frontend of one handle while a second stream runs the
    4-wave 128x128 GEMM back to back (its 128-register waves are the ones that fit beside conv0's on a SIMD)"""
    from sylber_amd import _lib
    lib = _lib.load()
    P = lambda t: ctypes.c_void_p(t.data_ptr())   # noqa: E731
    A = engines[0]
    x = _batch(LENS[:3], 50)
    M, N, K = 9216, 512, 1536
    a = torch.randn(M, K, device="cuda"); w = torch.randn(N, K, device="cuda"); c = torch.empty(M, N, device="cuda")
    S0, S1 = torch.cuda.Stream(), torch.cuda.Stream()
    ref = A.forward(x, LENS[:3], stop_stage=1).clone()
    torch.cuda.synchronize()
    stop = []

    def aggressor():
        with torch.cuda.stream(S1):
            while not stop:
                lib.sylber_op_linear(P(a), P(w), None, P(c), M, N, K, 1, 0, 3, ctypes.c_void_p(S1.cuda_stream))

    th = threading.Thread(target=aggressor)
    th.start()
    try:
        bad = 0
        for _ in range(300):
            with torch.cuda.stream(S0):
                o = A.forward(x, LENS[:3], stop_stage=1)
                S0.synchronize()
            bad += int(not torch.equal(o, ref))
    finally:
        stop.append(1)
        th.join()
    assert bad == 0, "%d of 300 forwards differ from the sequential result" % bad
