#!/usr/bin/env python3
"""A small interpreter for the IR of tools/gen_attn_asm.py: executes the generated attention key loop for the four waves of ONE
workgroup on numpy vectors (64 lanes per VGPR), with the memory-ordering rules that matter modelled pessimistically:

* an LDS-DMA piece (buffer_load ... lds) reaches LDS either at once (mode "early") or only when its wave executes the s_waitcnt vmcnt
  that covers it (mode "late") -- a missing wait or barrier shows up as poison read from LDS in one of the two modes;
* the registers of a ds_read are written only when the covering s_waitcnt lgkmcnt executes (data sampled at issue);
* waves run one after the other between barriers, in a configurable order.

What it cannot see are issue-timing hazards (MFMA result -> VALU etc.): gen_attn_asm.check_hazards covers those statically.
Test infrastructure (tests/test_attn_asm_gen.py); nothing under sylber_amd/ imports it."""
import numpy as np

import gen_attn_asm as G

NLANE = 64
POISON = 0x7FC07FC0          # bf16 NaN pair: anything computed from unwritten LDS becomes NaN


def bf16_round(x32):
    """float32 array -> bf16 bits (uint32 in the low 16), round to nearest even"""
    b = x32.view(np.uint32).astype(np.uint64)
    r = (b + 0x7FFF + ((b >> 16) & 1)) >> 16
    return (r & 0xFFFF).astype(np.uint32)


def bf16_to_f32(h):
    return (h.astype(np.uint32) << 16).view(np.float32)


class Wave:
    def __init__(self, wid, operands):
        self.id = wid
        self.v = np.zeros((512, NLANE), np.uint32)
        self.v[:] = 0x7FC00000                              # NaN: a read of a never-written register poisons the result
        self.ops = dict(operands)                            # name -> np.uint32[64] | np.uint32[n][64] | int | ("mem", array)
        self.vcc = np.zeros(NLANE, bool)
        self.scc = 0
        self.m0 = 0
        self.pc = 0
        self.vmq = []                                        # pending LDS-DMA pieces: (lds address, bytes)
        self.ldsq = []                                       # pending ds_read results: (first register, data [4][64])
        self.done = False
        self.at_barrier = False
        self.executed = 0


class Emu:
    def __init__(self, ins, lds_bytes, mode="late", fmt="bf16"):
        self.ins = [d for d in ins if d["kind"] != "comment"]
        self.labels = {d["name"]: i for i, d in enumerate(self.ins) if d["kind"] == "label"}
        self.lds = np.zeros(lds_bytes, np.uint8)
        self.lds.view(np.uint32)[:] = POISON
        self.mode = mode
        self.fmt = fmt

    # ---- operand access
    def rd(self, w, x, n=1):
        """-> uint32 [n][64] (n registers) or a broadcast scalar"""
        if isinstance(x, str):
            assert x == "m0"
            return np.uint32(w.m0)
        k = x[0]
        if k == "v":
            return w.v[x[1]]
        if k == "vr":
            return w.v[x[1]:x[1] + x[2]]
        if k in ("op", "opr"):
            val = w.ops[x[1]]
            return val
        if k == "lit":
            return np.uint32(x[1])
        if k == "imm":
            return np.uint32(x[1] & 0xFFFFFFFF)
        raise ValueError(x)

    def rdf(self, w, x):
        val = self.rd(w, x)
        if isinstance(val, np.ndarray):
            return val.view(np.float32)
        if x[0] == "imm":                                    # inline constant in a float instruction: small integers are NOT floats,
            assert x[1] == 0                                 # only 0 is used that way here
            return np.float32(0.0)
        return np.array([val], np.uint32).view(np.float32)[0]

    def wr(self, w, x, val):
        if isinstance(x, str):
            assert x == "m0"
            w.m0 = int(val) & 0xFFFFFFFF
            return
        k = x[0]
        if k == "v":
            w.v[x[1]] = np.broadcast_to(np.asarray(val, np.uint32), (NLANE,))
        elif k == "vr":
            w.v[x[1]:x[1] + x[2]] = val
        elif k == "op":
            w.ops[x[1]] = val if isinstance(val, np.ndarray) else int(val) & 0xFFFFFFFF
        else:
            raise ValueError(x)

    def sval(self, w, x):
        val = self.rd(w, x)
        assert not isinstance(val, np.ndarray) or val.ndim == 0, ("scalar operand expected", x)
        return int(val)

    # ---- instruction semantics
    def unpack16(self, regs):
        """uint32 [4][64] -> float32 [64 lanes][8 elements]"""
        lo = (regs & 0xFFFF).astype(np.uint32)
        hi = (regs >> 16).astype(np.uint32)
        inter = np.stack([lo, hi], 1).reshape(8, NLANE)       # element 2 r + {0, 1}
        if self.fmt == "bf16":
            return bf16_to_f32(inter).T
        return inter.astype(np.uint16).view(np.float16).astype(np.float32).T

    def mfma(self, w, d):
        a = self.unpack16(self.rd(w, d["a"]))                # [lane][8]: row lane & 31, k = 8 (lane >> 5) + e
        b = self.unpack16(self.rd(w, d["b"]))
        A = np.zeros((32, 16), np.float32)
        B = np.zeros((16, 32), np.float32)
        for lane in range(NLANE):
            r, h = lane & 31, lane >> 5
            A[r, 8 * h:8 * h + 8] = a[lane]
            B[8 * h:8 * h + 8, r] = b[lane]
        with np.errstate(invalid="ignore", over="ignore"):
            D = A.astype(np.float64) @ B.astype(np.float64)
        c = d["c"]
        if c[0] == "imm":
            Cm = np.zeros((16, NLANE), np.float32)
        else:
            Cm = self.rd(w, c).view(np.float32)
        out = np.zeros((16, NLANE), np.float32)
        for lane in range(NLANE):
            col, h = lane & 31, lane >> 5
            for r16 in range(16):
                row = 8 * (r16 >> 2) + 4 * h + (r16 & 3)
                out[r16, lane] = np.float32(D[row, col] + np.float64(Cm[r16, lane]))
        self.wr(w, d["dst"], out.view(np.uint32))

    def valu(self, w, d):
        o, dst, src = d["op"], d["dst"], d["src"]
        f = lambda i: np.broadcast_to(np.asarray(self.rdf(w, src[i]), np.float32), (NLANE,))
        u = lambda i: self.rd(w, src[i])
        with np.errstate(invalid="ignore", over="ignore", under="ignore", divide="ignore"):
            if o == "v_mov_b32":
                res = np.broadcast_to(np.asarray(u(0), np.uint32), (NLANE,)).copy()
            elif o == "v_add_u32":
                res = (np.asarray(u(0), np.uint64) + np.asarray(u(1), np.uint64)).astype(np.uint64) & 0xFFFFFFFF
                res = np.broadcast_to(res.astype(np.uint32), (NLANE,)).copy()
            elif o in ("v_and_b32", "v_xor_b32"):
                a0 = np.broadcast_to(np.asarray(u(0), np.uint32), (NLANE,))
                a1 = np.broadcast_to(np.asarray(u(1), np.uint32), (NLANE,))
                res = (a0 & a1) if o == "v_and_b32" else (a0 ^ a1)
            elif o == "v_lshlrev_b32":                       # D = S1 << S0
                res = (np.broadcast_to(np.asarray(u(1), np.uint32), (NLANE,)).astype(np.uint64) << int(u(0))).astype(np.uint64) & 0xFFFFFFFF
                res = res.astype(np.uint32)
            elif o == "v_cvt_f32_f16":
                res = (np.asarray(u(0), np.uint32) & 0xFFFF).astype(np.uint16).view(np.float16).astype(np.float32).view(np.uint32)
            elif o == "v_subrev_u32":
                res = ((np.asarray(u(1), np.int64) - np.asarray(u(0), np.int64)) & 0xFFFFFFFF).astype(np.uint32)
                res = np.broadcast_to(res, (NLANE,)).copy()
            elif o == "v_max3_f32":
                res = np.maximum(np.maximum(f(0), f(1)), f(2)).astype(np.float32).view(np.uint32)
            elif o == "v_max_f32":
                res = np.maximum(f(0), f(1)).astype(np.float32).view(np.uint32)
            elif o == "v_add_f32":
                res = np.broadcast_to((np.float32(1) * f(0) + f(1)).astype(np.float32), (NLANE,)).copy().view(np.uint32)
            elif o == "v_sub_f32":
                res = np.broadcast_to((f(0) - f(1)).astype(np.float32), (NLANE,)).copy().view(np.uint32)
            elif o == "v_mul_f32":
                res = np.broadcast_to((f(0) * f(1)).astype(np.float32), (NLANE,)).copy().view(np.uint32)
            elif o == "v_fmamk_f32":                         # D = S0 * K + S1, one rounding
                res = (np.asarray(f(0), np.float64) * np.asarray(f(1), np.float64) + np.asarray(f(2), np.float64)).astype(np.float32).view(np.uint32)
            elif o == "v_exp_f32":
                res = np.exp2(f(0).astype(np.float64)).astype(np.float32).view(np.uint32)
            elif o == "v_cvt_pk_bf16_f32":
                res = bf16_round(np.ascontiguousarray(f(0))) | (bf16_round(np.ascontiguousarray(f(1))) << 16)
            elif o == "v_cvt_pk_f16_f32":
                lo = f(0).astype(np.float16).view(np.uint16).astype(np.uint32)
                hi = f(1).astype(np.float16).view(np.uint16).astype(np.uint32)
                res = lo | (hi << 16)
            elif o == "v_cndmask_b32":                       # D = vcc ? src1 : src0
                a0 = np.broadcast_to(np.asarray(u(0), np.uint32), (NLANE,))
                a1 = np.broadcast_to(np.asarray(u(1), np.uint32), (NLANE,))
                res = np.where(w.vcc, a1, a0).astype(np.uint32)
            elif o == "v_permlane32_swap_b32":               # swaps dst[32:64] with src[0:32]
                a0, a1 = self.rd(w, dst).copy(), u(0).copy()
                n0, n1 = a0.copy(), a1.copy()
                n0[32:] = a1[:32]
                n1[:32] = a0[32:]
                self.wr(w, dst, n0)
                self.wr(w, src[0], n1)
                return
            else:
                raise ValueError(o)
        self.wr(w, dst, np.asarray(res, np.uint32))

    def step(self, w):
        """execute one instruction of wave w; returns False when the wave parks at a barrier or ends"""
        if w.pc >= len(self.ins):
            self.flush(w)
            w.done = True
            return False
        d = self.ins[w.pc]
        w.pc += 1
        w.executed += 1
        k = d["kind"]
        if k == "label":
            return True
        if k == "mfma":
            self.mfma(w, d)
        elif k == "valu":
            self.valu(w, d)
        elif k == "vcmp":
            if d["op"] in ("v_cmp_gt_f32", "v_cmp_lt_f32"):
                with np.errstate(invalid="ignore"):
                    a_, b_ = self.rdf(w, d["a"]), self.rdf(w, d["b"])
                    w.vcc = np.broadcast_to((a_ > b_) if d["op"] == "v_cmp_gt_f32" else (a_ < b_), (NLANE,)).copy()
            elif d["op"] == "v_cmp_lt_i32":
                a = np.asarray(self.rd(w, d["a"]), np.uint32).astype(np.int64)
                b = np.asarray(self.rd(w, d["b"]), np.uint32).view(np.int32).astype(np.int64)
                a = np.where(a >= 2 ** 31, a - 2 ** 32, a)
                w.vcc = np.broadcast_to(a < b, (NLANE,)).copy()
            else:
                raise ValueError(d["op"])
        elif k == "ds_read":
            addr = self.rd(w, d["addr"]).astype(np.int64) + d["off"]
            data = np.zeros((4, NLANE), np.uint32)
            for lane in range(NLANE):
                a = int(addr[lane])
                assert a % 16 == 0 and 0 <= a <= len(self.lds) - 16, ("LDS read out of range", a)
                data[:, lane] = self.lds[a:a + 16].view(np.uint32)
            w.ldsq.append((d["dst"], data))
        elif k == "dma":
            voff = self.rd(w, d["voff"]).astype(np.int64)
            mem = w.ops[d["rsrc"][1]][1]
            soff = self.sval(w, d["soff"])
            buf = np.zeros(1024, np.uint8)
            for lane in range(NLANE):
                a = int(voff[lane]) + soff
                assert 0 <= a <= len(mem) - 16, ("global read out of range", a, len(mem))
                buf[16 * lane:16 * lane + 16] = mem[a:a + 16]
            assert 0 <= w.m0 <= len(self.lds) - 1024 and w.m0 % 16 == 0, ("LDS-DMA destination", w.m0)
            if self.mode == "early":
                self.lds[w.m0:w.m0 + 1024] = buf
                w.vmq.append(None)
            else:
                w.vmq.append((w.m0, buf))
        elif k == "salu":
            o, src = d["op"], d["src"]
            if o == "s_mov_b32":
                self.wr(w, d["dst"], self.sval(w, src[0]))
            elif o == "s_add_u32":
                t = self.sval(w, src[0]) + self.sval(w, src[1])
                w.scc = 1 if t >= 2 ** 32 else 0
                self.wr(w, d["dst"], t & 0xFFFFFFFF)
            elif o == "s_sub_u32":
                a, b = self.sval(w, src[0]), self.sval(w, src[1])
                w.scc = 1 if b > a else 0
                self.wr(w, d["dst"], (a - b) & 0xFFFFFFFF)
            elif o == "s_cselect_b32":
                self.wr(w, d["dst"], self.sval(w, src[0]) if w.scc else self.sval(w, src[1]))
            else:
                raise ValueError(o)
        elif k == "scmp":
            a, b = self.sval(w, d["a"]), self.sval(w, d["b"])
            w.scc = int({"s_cmp_eq_u32": a == b, "s_cmp_lg_u32": a != b, "s_cmp_ge_u32": a >= b, "s_cmp_lt_u32": a < b}[d["op"]])
        elif k == "branch":
            o = d["op"]
            take = {"s_branch": True, "s_cbranch_scc1": w.scc == 1, "s_cbranch_scc0": w.scc == 0,
                    "s_cbranch_vccnz": bool(w.vcc.any()), "s_cbranch_vccz": not w.vcc.any()}[o]
            if take:
                w.pc = self.labels[d["target"]]
        elif k == "nop":
            pass
        elif k == "waitcnt":
            if d["vm"] is not None:
                while len(w.vmq) > d["vm"]:
                    piece = w.vmq.pop(0)
                    if piece is not None:
                        self.lds[piece[0]:piece[0] + 1024] = piece[1]
            if d["lgkm"] is not None:
                while len(w.ldsq) > d["lgkm"]:
                    dst, data = w.ldsq.pop(0)
                    self.wr(w, dst, data)
        elif k == "barrier":
            w.at_barrier = True
            return False
        else:
            raise ValueError(k)
        return True

    def flush(self, w):
        for piece in w.vmq:
            if piece is not None:
                self.lds[piece[0]:piece[0] + 1024] = piece[1]
        w.vmq = []

    def run(self, waves, order=None, limit=2_000_000):
        order = list(range(len(waves))) if order is None else order
        n = 0
        while not all(w.done for w in waves):
            for i in order:
                w = waves[i]
                if w.done or w.at_barrier:
                    continue
                while self.step(w):
                    n += 1
                    assert n < limit, "instruction limit (endless loop?)"
            live = [w for w in waves if not w.done]
            if live and all(w.at_barrier for w in live):
                assert len(live) == len(waves), "a wave ended while others wait at a barrier"
                for w in live:
                    w.at_barrier = False
        return waves


# ====================================================================================================================================
# harness: one workgroup = 128 queries of one (utterance, head)
# ====================================================================================================================================
def swap23(n):
    """V^T key position of key n: bits 2 and 3 swapped (written by the q/k/v GEMM epilogue, read by the attention kernels)"""
    return (n & ~0xC) | ((n & 4) << 1) | ((n & 8) >> 1)


def to16(x, fmt):
    if fmt == "bf16":
        return bf16_round(np.ascontiguousarray(x, np.float32).reshape(-1)).astype(np.uint16).reshape(x.shape)
    return x.astype(np.float16).view(np.uint16)


def from16(h, fmt):
    if fmt == "bf16":
        return bf16_to_f32(h.astype(np.uint32))
    return h.view(np.float16).astype(np.float32)


def run_workgroup(q, k, vmat, nvalid, qblock=0, fmt="bf16", mode="late", order=None, ring_base=0, prog=None):
    """q, k, vmat: float32 [T][64] of one (utterance, head), q already scaled by log2(e) / 8 (scores in log2 units).  Returns (ctx [128][64] float32 for queries
    qblock*128 ..., the generator's instruction count per wave)."""
    T = q.shape[0]
    Tp = (T + 31) & ~31
    Tpv = (Tp + 63) & ~63
    q16 = np.zeros((Tp, 64), np.uint16); q16[:T] = to16(q, fmt)
    k16 = np.zeros((Tp + 64, 64), np.uint16); k16[:T] = to16(k, fmt)
    k16[Tp:] = 0x7FC0                                            # rows behind the utterance: NaN (their scores must be masked, not used)
    vt16 = np.zeros((64, Tpv), np.uint16)
    v16 = to16(vmat, fmt)
    for n in range(T):
        vt16[:, (n & ~15) | swap23(n & 15)] = v16[n]
    kmem = np.ascontiguousarray(k16).view(np.uint8).reshape(-1)
    vmem = np.ascontiguousarray(vt16).view(np.uint8).reshape(-1)
    if prog is None:
        prog = G.Gen(fmt).build()
        G.check_hazards(prog.ins)
    nt = (nvalid + 63) // 64
    emu = Emu(prog.ins, ring_base + G.NSLOT * G.SLOT, mode=mode, fmt=fmt)
    waves = []
    lane = np.arange(NLANE)
    ql, h = lane & 31, lane >> 5
    swz = (ql >> 1) & 7
    for w in range(4):
        q0 = qblock * 128 + w * 32
        qr = np.minimum(q0 + ql, Tp - 1)
        ops = {}
        for ks in range(4):
            frag = np.zeros((4, NLANE), np.uint32)
            for ln in range(NLANE):
                e = q16[qr[ln], ks * 16 + h[ln] * 8: ks * 16 + h[ln] * 8 + 8].astype(np.uint32)
                frag[:, ln] = e[0::2] | (e[1::2] << 16)
            ops["q%d" % ks] = frag
            ops["off%d" % ks] = (ring_base + ql * 128 + (((2 * ks + h) ^ swz) << 4)).astype(np.uint32)
        srow, spos = lane >> 3, lane & 7
        for i in range(2):
            r = w * 16 + i * 8 + srow
            c = spos ^ ((r >> 1) & 7)
            ops["kvoff%d" % i] = ((r * 64 + c * 8) * 2).astype(np.uint32)
            ops["vvoff%d" % i] = ((r * Tpv + c * 8) * 2).astype(np.uint32)
        ops["limbase"] = ((nvalid - 4 * h) & 0xFFFFFFFF).astype(np.uint32)
        ops["rsk"] = ("mem", kmem)
        ops["rsv"] = ("mem", vmem)
        ops["ldsw"] = ring_base + w * 2048
        ops["nt"] = nt
        ops["kvl0"] = 64 * (nt - 1)
        ops["kvl1"] = 64 * (nt - 1) + 32
        for nm in ("koff", "voff", "resc", "snext", "dslot", "tdma", "tleft"):
            ops[nm] = 0xDEAD0000                                 # outputs: written before they are read
        waves.append(Wave(w, ops))
    emu.run(waves, order)
    ctx = np.zeros((128, 64), np.float32)
    for w, wv in enumerate(waves):
        o = wv.v[G.R["O0"]:G.R["O0"] + 32].view(np.float32)      # O0, O1
        l = wv.v[G.R["l"]].view(np.float32)
        ltot = l + l[lane ^ 32]
        for ln in range(NLANE):
            qq, hh = ln & 31, ln >> 5
            for ds in range(2):
                for r16 in range(16):
                    dd = 32 * ds + 8 * (r16 >> 2) + 4 * hh + (r16 & 3)
                    ctx[w * 32 + qq, dd] = o[16 * ds + r16, ln] / ltot[ln]
    return ctx, [wv.executed for wv in waves]


def reference(q, k, vmat, nvalid, fmt="bf16"):
    """softmax(q k^T) v over the valid keys, on the 16-bit-rounded operands, float64"""
    qq, kk, vv = (from16(to16(x, fmt), fmt).astype(np.float64) for x in (q, k, vmat))
    s = qq @ kk[:nvalid].T                                   # log2 units (q carries log2(e) / 8)
    s -= s.max(1, keepdims=True)
    p = np.exp2(s)
    return (p / p.sum(1, keepdims=True)) @ vv[:nvalid]


if __name__ == "__main__":
    import sys
    rng = np.random.default_rng(0)
    T = int(sys.argv[1]) if len(sys.argv) > 1 else 200
    nvalid = int(sys.argv[2]) if len(sys.argv) > 2 else T
    q = rng.standard_normal((T, 64)).astype(np.float32) * 0.125 * 3
    k = rng.standard_normal((T, 64)).astype(np.float32)
    vm = rng.standard_normal((T, 64)).astype(np.float32)
    for mode in ("late", "early"):
        ctx, n = run_workgroup(q, k, vm, nvalid, mode=mode)
        ref = reference(q, k, vm, nvalid)
        nq = min(128, T)
        err = np.abs(ctx[:nq] - ref[:nq]).max()
        print(mode, "T", T, "nvalid", nvalid, "max abs err", err, "instructions per wave", n)
