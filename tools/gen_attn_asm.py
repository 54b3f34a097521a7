#!/usr/bin/env python3
"""Generate the hand-scheduled key loop of attention_asm_kernel (csrc/attention.hip) -> attn_asm_{bf16,f16}.inc.

The whole key loop of one wave (32 queries x all keys of one (utterance, head)) is ONE inline-asm statement, software-pipelined
over 32-key HALF-TILES so that the matrix pipe never waits for the softmax of the same wave (semantics:
transformers eager_attention_forward / sdpa TP:234-259 reached from sylber/model/sylber.py:122 -- softmax(q k^T / 8 + key mask) v):

    iteration i:   MFMA   S(i+1) = K(half i+1) . Q^T - m     4 x v_mfma_f32_32x32x16, a chain on one accumulator block whose first link
                                                             takes the SEED BLOCK (16 registers holding -m) as its C operand
                   MFMA   O^T   += V^T(half i-1) . P(i-1)^T  4 x v_mfma (two accumulator blocks x two 16-key groups)
                   VALU   P(i)   = exp2(S(i)), row sums, 16-bit pack   (49 instructions: q is pre-scaled by log2(e) / 8 by its producer,
                                                             so the scores leave the matrix pipe relative to m and in log2 units)
    issue order P.V, P.V, S, P.V, S, P.V, S, S: the S chain starts behind this iteration's lazy-maximum decision (which may rewrite the
    seed block), and behind a tile barrier the P.V fragments are the ones already in registers; the VALU work is dealt into the gaps.

Everything stays "query = lane & 31" as in the compiler-scheduled attention_bf16_kernel (kept as the reference): S^T[key][q]
leaves the matrix pipe with lane (q, h) holding keys 8 g + 4 h + e of the half-tile in register 4 g + e, which packed to 16 bits in
register order IS the B operand of the two P.V MFMAs (V^T's key axis is stored with bits 2 and 3 swapped by the q/k/v GEMM epilogue).
P has ONE register buffer: a pack is gated behind the issue of the P.V MFMAs that still read the registers it overwrites.

Registers are FIXED physical VGPRs (an asm operand cannot be addressed by sub-register, and the softmax works on single registers
of the MFMA accumulator blocks): see R below (163 registers: three waves per SIMD).  Inputs arrive in compiler-allocated operands.

LDS: a ring of three 16-KiB slots per workgroup (K tile of 64 keys x 128 B at +0, V^T tile of 64 features x 128 B at +8192); three
tiles are live at any time (V of tile t-1, K and V of tile t, K of tile t+1).  One barrier per 64-key tile, at the head of every odd
iteration: behind it every wave's LDS-DMA pieces of tile t+1 have landed (own pieces: s_waitcnt vmcnt(0) before the barrier) and
every wave has finished reading tile t-1, whose slot the pieces of tile t+2 are then requested into.

Online softmax: the running maximum is LAZY (as in attention_bf16_kernel): scores are exponentiated against a stale reference until
a relative score exceeds 8 (p <= 2^8); the test is per lane against the lane's own 16 scores (no cross-lane exchange on the fast
path -- both halves of a query hold the same m, and the slow path, entered by the whole wave when ANY lane trips, exchanges the maxima
with v_permlane32_swap, decides per query, rebases the scores in flight and rewrites the seed block).  The rescale of the accumulator
by alpha is deferred to the end of the iteration (the P.V MFMAs of half-tile i-1, issued during iteration i, belong to the old scale).

The instruction list is built as a small IR that is (a) printed as the asm text, (b) checked statically for the hazards the
compiler would otherwise handle (MFMA result -> VALU: 12 wait states on gfx950; VALU result -> MFMA operand: 2; transcendental
result -> next VALU: 1; M0 write -> LDS-DMA: 1; along both arms of every branch), and (c) EXECUTED by tools/attn_asm_emu.py against a
numpy softmax (tests/test_attn_asm_gen.py, CPU tier).

History (profiles/r05_attention.md): v1 multiplied and subtracted per score (65 VALU); v2 fed -m through the matrix pipe as a ninth
MFMA (constant-one feature x -m) -- time-neutral under the chip's power cap; v3 (this one) seeds the chain's accumulator instead: -2 %.
"""
import os
import struct
import sys

OUTDIR = os.environ.get("GEN_GEMM_ASM_OUT") or os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "sylber_amd", "build", "gen")

SLOT = 16384            # bytes of one ring slot (K tile + V^T tile)
VT = 8192               # offset of the V^T tile inside a slot
NSLOT = 3

# ---- fixed physical VGPRs -------------------------------------------------------------------------------------------------------
R = {
    "O0": 40, "O1": 56,                 # O^T accumulators (features 0-31 / 32-63), 16 registers each
    "SA": 72, "SB": 88,                 # score blocks (double buffer by half-tile parity)
    "P": 104,                           # packed probabilities, 8 registers (ONE buffer: a pack is held back until the P.V MFMAs that read
                                        # its registers have issued)
    "F": 112,                           # fragment ring: 4 x 4 registers
    "SD": 128,                          # seed block: 16 registers holding -m, the C operand of the first MFMA of every S chain
    "rk": 144, "rv": 148,               # LDS read addresses (4 each): slot of the K tile being read / of the V^T tile being read
    "m": 152, "l": 153, "mx": 154, "ps": 155,   # ps: 155..158
    "alpha": 159, "ta": 160, "tb": 161, "ninf": 162,
}
FIXED_LO, FIXED_HI = 40, 162

LOG2E = 1.44269504088896341
def f32bits(x):
    return struct.unpack("<I", struct.pack("<f", x))[0]
LIT_THR = f32bits(8.0)                      # lazy-maximum slack: p <= 2^8 (scores arrive in log2 units: q is pre-scaled by log2(e) / 8)
LIT_NINF = 0xff800000


def v(n):
    return ("v", n)

def vr(n, c):
    return ("vr", n, c)

def op(name):               # compiler-allocated operand, single register (VGPR or SGPR: the printer does not care)
    return ("op", name)

def opr(name, c):           # compiler-allocated VGPR tuple operand
    return ("opr", name, c)

def lit(bits):
    return ("lit", bits)

def imm(x):
    return ("imm", x)


class Prog:
    def __init__(self):
        self.ins = []

    def add(self, kind, **kw):
        d = dict(kind=kind, **kw)
        self.ins.append(d)
        return d

    # ---- instruction constructors (each records reads / writes as sets of fixed VGPR numbers for the hazard checker)
    def comment(self, text):
        self.add("comment", text=text)

    def label(self, name):
        self.add("label", name=name)

    def mfma(self, dst, a, b, c):
        self.add("mfma", dst=dst, a=a, b=b, c=c)

    def ds_read(self, dst, addr, off):
        self.add("ds_read", dst=dst, addr=addr, off=off)

    def valu(self, opname, dst, *src, trans=False):
        self.add("valu", op=opname, dst=dst, src=list(src), trans=trans)

    def vcmp(self, opname, a, b):
        self.add("vcmp", op=opname, a=a, b=b)

    def salu(self, opname, dst, *src):
        self.add("salu", op=opname, dst=dst, src=list(src))

    def scmp(self, opname, a, b):
        self.add("scmp", op=opname, a=a, b=b)

    def branch(self, opname, target):
        self.add("branch", op=opname, target=target)

    def nop(self, n):
        self.add("nop", n=n)

    def waitcnt(self, vm=None, lgkm=None):
        self.add("waitcnt", vm=vm, lgkm=lgkm)

    def barrier(self):
        self.add("barrier")

    def dma(self, voff, rsrc, soff):
        self.add("dma", voff=voff, rsrc=rsrc, soff=soff)


def regs_of(x):
    """fixed VGPR numbers an operand touches (operands allocated by the compiler are outside the fixed block by construction)"""
    if x is None:
        return set()
    if x[0] == "v":
        return {x[1]}
    if x[0] == "vr":
        return set(range(x[1], x[1] + x[2]))
    return set()


# ====================================================================================================================================
# the schedule
# ====================================================================================================================================
# knock-out variants for timing experiments (results wrong by construction; experiments build only)
VARIANTS = {
    1: {"noexp"},                      # v_exp_f32 -> v_mov_b32
    2: {"nosoftmax"},                  # no softmax instructions at all
    3: {"nomfma"},                     # no MFMAs
    4: {"nolds"},                      # no fragment reads
    5: {"nodma"},                      # no LDS-DMA, no tile barrier
    6: {"nomax"},                      # no row maximum / lazy-maximum test
    7: {"nolds", "nodma"},             # MFMA + softmax only
    8: {"nolds", "nodma", "nosoftmax"},   # MFMA only
    9: {"nosoftmax", "nomfma"},        # data movement only
}
TUNE_FLAGS = set()


class Gen:
    def __init__(self, fmt="bf16", knock=()):
        self.p = Prog()
        self.knock = set(knock)
        self.fmt = fmt
        self.cvt = "v_cvt_pk_bf16_f32" if fmt == "bf16" else "v_cvt_pk_f16_f32"
        self.lds_order = []          # frag-ring slots with a read in flight, in LDS issue order: for exact lgkmcnt values
        self.landed = set()          # ring slots whose fragment has landed and not been consumed yet
        self.uid = 0

    def new_label(self, stem):
        self.uid += 1
        return "L_attn_%s_%d_%%=" % (stem, self.uid)

    # ---- fragment reads -----------------------------------------------------------------------------------------------------------
    def F(self, slot):
        return vr(R["F"] + 4 * slot, 4)

    def read(self, slot, addr_reg, off):
        assert slot not in self.lds_order and slot not in self.landed, ("ring slot %d still holds an unconsumed fragment" % slot)
        self.p.ds_read(self.F(slot), v(addr_reg), off)
        self.lds_order.append(slot)

    def wait_frag(self, slot):
        """wait until the outstanding read into ring slot `slot` has landed (LDS operations return in order)"""
        if slot in self.landed:                                  # drained by an earlier, wider wait
            self.landed.discard(slot)
            return
        assert slot in self.lds_order, ("fragment never requested", slot)
        k = self.lds_order.index(slot)
        younger = len(self.lds_order) - 1 - k
        self.p.waitcnt(lgkm=younger)
        self.landed |= set(self.lds_order[:k])
        self.lds_order = self.lds_order[k + 1:]

    def drain(self):
        self.landed |= set(self.lds_order)
        self.lds_order = []

    def frag_state(self):
        return (tuple(self.lds_order), tuple(sorted(self.landed)))

    # ---- softmax of one half-tile as a list of (closure, gate) pairs (each closure emits ONE instruction) ---------------------------
    def softmax_ops(self, S, masked, init, kv_half, resume_label, slow_label, has_pv):
        """The scores arrive RELATIVE to the query's reference m and in log2 units: q is pre-scaled by log2(e) / 8 by its producer and the
        first MFMA of every S chain takes the seed block (-m in all 16 registers) as its C operand, so the softmax is exp2 + row sum +
        pack and nothing else.  Only the very first half-tile (init: no reference yet) subtracts explicitly.
        gate: None, or the set of MFMA positions that must have ISSUED before the instruction may be emitted (the packs overwrite the
        single P buffer the P.V MFMAs of this iteration still read)."""
        p, ops = self.p, []
        s = [v(S + r) for r in range(16)]
        pr = [v(R["P"] + r) for r in range(8)]
        ps = [v(R["ps"] + k) for k in range(4)]
        mx, m, l, ta, tb = (v(R[k]) for k in ("mx", "m", "l", "ta", "tb"))
        E = lambda f, gate=None: ops.append((f, gate))
        if masked:
            # key of register r: kv + (r & 3) + 8 (r >> 2) + 4 h ; valid iff  c_r < nvalid - 4 h - kv =: lim
            E(lambda: p.valu("v_subrev_u32", ta, op(kv_half), op("limbase")))          # lim = limbase - kv
            for r in range(16):
                c = (r & 3) + 8 * (r >> 2)
                E(lambda c=c: p.vcmp("v_cmp_lt_i32", imm(c), ta))                        # c < lim: the key is valid
                E(lambda r=r: p.valu("v_cndmask_b32", s[r], v(R["ninf"]), s[r]))           # vcc ? s : -inf (a literal AND vcc: two constant-bus reads)
        # row maximum of the lane's 16 scores: a tree of v_max3 (the tree keeps the dependent chains short)
        E(lambda: p.valu("v_max3_f32", ps[0], s[0], s[1], s[2]))
        E(lambda: p.valu("v_max3_f32", ps[1], s[3], s[4], s[5]))
        E(lambda: p.valu("v_max3_f32", ps[2], s[6], s[7], s[8]))
        E(lambda: p.valu("v_max3_f32", ps[3], s[9], s[10], s[11]))
        E(lambda: p.valu("v_max3_f32", mx, s[12], s[13], s[14]))
        E(lambda: p.valu("v_max3_f32", ps[0], ps[0], ps[1], ps[2]))
        E(lambda: p.valu("v_max3_f32", mx, mx, ps[3], s[15]))
        E(lambda: p.valu("v_max_f32", mx, mx, ps[0]))
        if init:
            # first half-tile: raw scores.  m = the query's maximum over both lane halves (key 0 is always valid, so it is finite)
            E(lambda: p.valu("v_mov_b32", ta, mx))
            E(lambda: p.valu("v_mov_b32", tb, mx))
            E(lambda: p.nop(1))
            E(lambda: p.valu("v_permlane32_swap_b32", ta, tb))
            E(lambda: p.nop(1))
            E(lambda: p.valu("v_max_f32", m, ta, tb))
            for r in range(16):
                E(lambda r=r: p.valu("v_sub_f32", v(R["SD"] + r), imm(0), m))
            for r in range(16):
                E(lambda r=r: p.valu("v_sub_f32", s[r], s[r], m))
            E(lambda: self.mark_ref_ready())                     # (behind the rebase: a VALU result needs two wait states before an MFMA reads it)
        else:
            E(lambda: p.vcmp("v_cmp_lt_f32", lit(LIT_THR), mx))
            E(lambda: p.branch("s_cbranch_vccnz", slow_label))
            E(lambda: (p.label(resume_label), self.mark_ref_ready()))
        # p = exp2(s'), in place; row sum in four partial sums; packed pairs in register order
        g0 = {1, 3} if has_pv else None          # P[0..3] is the B operand of the P.V MFMAs at positions 1 and 3, P[4..7] of 5 and 7
        g1 = {5, 7} if has_pv else None
        ex = lambda r: E(lambda r=r: p.valu("v_exp_f32", s[r], s[r], trans=True))
        add0 = lambda k: E(lambda k=k: p.valu("v_add_f32", ps[k], s[k], s[4 + k]))
        addn = lambda k, b: E(lambda k=k, b=b: p.valu("v_add_f32", ps[k], ps[k], s[b + k]))
        cv = lambda i: E(lambda i=i: p.valu(self.cvt, pr[i], s[2 * i], s[2 * i + 1]), g0 if i < 4 else g1)
        for r in range(8):
            ex(r)
        for k in range(4):
            add0(k)
        for r in range(8, 12):
            ex(r)
        cv(0); cv(1)
        for r in range(12, 16):
            ex(r)
        cv(2); cv(3)
        for k in range(4):
            addn(k, 8)
        for k in range(4):
            addn(k, 12)
        E(lambda: p.valu("v_add_f32", ps[0], ps[0], ps[1]))
        E(lambda: p.valu("v_add_f32", ps[2], ps[2], ps[3]))
        E(lambda: p.valu("v_add_f32", ps[0], ps[0], ps[2]))
        E(lambda: p.valu("v_add_f32", l, l, ps[0]))
        cv(4); cv(5); cv(6); cv(7)
        return ops

    def mark_ref_ready(self):
        self.ref_ready = True

    def slow_path(self, slow_label, resume_label, S):
        """some lane's relative maximum left the lazy window (> 8): new reference per QUERY (both lane halves decide alike),
        m' = m + max, the CURRENT scores are rebased by delta = m' - m, alpha = exp2(-delta) (= 1 exactly where the query keeps its
        reference); l is rescaled here, O at the end of the iteration; the seed block is rewritten, so the S chain in flight (whose
        first MFMA is issued only behind this point of the iteration) is relative to the new reference"""
        p = self.p
        mx, m, l, ta, tb, al = (v(R[k]) for k in ("mx", "m", "l", "ta", "tb", "alpha"))
        p.label(slow_label)
        p.valu("v_mov_b32", ta, mx)
        p.valu("v_mov_b32", tb, mx)
        p.nop(1)
        p.valu("v_permlane32_swap_b32", ta, tb)
        p.nop(1)
        p.valu("v_max_f32", mx, ta, tb)
        p.valu("v_add_f32", ta, mx, m)
        p.vcmp("v_cmp_lt_f32", lit(LIT_THR), mx)
        p.valu("v_cndmask_b32", ta, m, ta)                       # m' = need ? m + max : m
        p.valu("v_sub_f32", tb, ta, m)                           # delta >= 0 (exactly 0 where the reference stays)
        p.valu("v_mov_b32", m, ta)
        for r in range(16):
            p.valu("v_sub_f32", v(R["SD"] + r), imm(0), m)
        for r in range(16):
            p.valu("v_sub_f32", v(S + r), v(S + r), tb)
        p.valu("v_sub_f32", al, imm(0), tb)
        p.valu("v_exp_f32", al, al, trans=True)
        p.nop(0)
        p.valu("v_mul_f32", l, l, al)
        p.salu("s_mov_b32", op("resc"), imm(1))
        p.branch("s_branch", resume_label)

    def rescale_block(self):
        p = self.p
        skip = self.new_label("noresc")
        p.scmp("s_cmp_eq_u32", op("resc"), imm(0))
        p.branch("s_cbranch_scc1", skip)
        p.nop(15)                                                # the last P.V MFMAs wrote O: 12 wait states before a VALU may touch it
        for ds in ("O0", "O1"):
            for r in range(16):
                p.valu("v_mul_f32", v(R[ds] + r), v(R["alpha"]), v(R[ds] + r))
        p.salu("s_mov_b32", op("resc"), imm(0))
        p.nop(3)                                                 # VALU write -> MFMA SrcC
        p.label(skip)

    # ---- one iteration ------------------------------------------------------------------------------------------------------------
    def iteration(self, name, parity, s_half, pv_half, masked=False, init=False, pv_first=False, nxt=None, kv_half=None,
                  barrier_dma=False, addr_update=None):
        """parity: 0 = even half-tile (softmax on SA, S MFMAs -> SB), 1 = odd.  P has ONE buffer.
        s_half: None or the K half (0 / 1) of S(i+1) read through rk; pv_half: None or the V half of P.V(i-1) read through rv.
        nxt: (s_half, pv_half, k_after_barrier) of the NEXT iteration, for the prefetch of its first four fragments (None: no S / PV).
        barrier_dma: this iteration opens a new tile: wait for the own pieces of tile t+1, barrier, request tile t+2, THEN read K.
        addr_update: "full" (rv <- rk, rk <- slot of the next tile) / "rv" (rv <- rk only) / None.
        MFMA positions: even n = S chain link n / 2 (K fragment ks = n / 2), odd n = P.V (j = n / 4, ds = (n / 2) % 2); position n uses ring
        slot n % 4.  ISSUE order: P.V, P.V, S, P.V, S, P.V, S, S -- the S chain starts behind this iteration's lazy-maximum decision (its
        first link reads the seed block), and behind a tile barrier the P.V fragments are the ones already in registers."""
        p = self.p
        p.comment("==== %s" % name)
        Ssm = R["SA"] if parity == 0 else R["SB"]
        Smm = R["SB"] if parity == 0 else R["SA"]
        resume, slow = self.new_label("resume"), self.new_label("slow")
        has_s, has_pv = s_half is not None, pv_half is not None
        fill = self.softmax_ops(Ssm, masked, init, kv_half, resume, slow, has_pv)
        mf = []
        for n in range(8):
            if n % 2 == 0:
                mf.append(("S", n // 2) if has_s else None)
            else:
                mf.append(("PV", (n // 2) // 2, (n // 2) % 2) if has_pv else None)
        if barrier_dma:
            # the K fragments of positions 0 and 2 could not be requested before the barrier (they are read from tile t+1)
            p.waitcnt(vm=0, lgkm=0)
            self.drain()
            p.barrier()
            self.dma_block()
            self.read(0, R["rk"] + 0, s_half * 4096)
            self.read(2, R["rk"] + 1, s_half * 4096)
        order = [n for n in (1, 3, 0, 5, 2, 7, 4, 6) if mf[n] is not None]
        nf, gaps = len(fill), len(order)
        done_f = 0
        issued = set()
        self.ref_ready = False
        state = {"rv_moved": False, "rk_moved": False}

        def emit_fillers(want, force_ref=False):
            nonlocal done_f
            while done_f < nf and (done_f < want or (force_ref and not self.ref_ready)):
                f, gate = fill[done_f]
                if gate is not None and not gate <= issued:
                    assert not force_ref, "a gated pack in front of the lazy-maximum decision"
                    break
                f()
                done_f += 1

        def addr_steps():
            # rv <- rk once every read of this iteration through rv (V of tile t-1: the refills behind positions 1 and 3) has been issued
            # -- from then on "through rv" means V of tile t; rk <- slot of the next tile once rv has its copy and every read through
            # rk (K of tile t: the refills behind positions 0 and 2) has been issued
            if not addr_update:
                return
            if not state["rv_moved"] and {n for n in (1, 3) if mf[n] is not None} <= issued:
                for k in range(4):
                    p.valu("v_mov_b32", v(R["rv"] + k), v(R["rk"] + k))
                state["rv_moved"] = True
            if addr_update == "full" and state["rv_moved"] and not state["rk_moved"] and {n for n in (0, 2) if mf[n] is not None} <= issued:
                for k in range(4):
                    p.valu("v_add_u32", v(R["rk"] + k), op("snext"), op("off%d" % k))
                p.salu("s_add_u32", op("snext"), op("snext"), imm(SLOT))
                p.scmp("s_cmp_eq_u32", op("snext"), imm(NSLOT * SLOT))
                p.salu("s_cselect_b32", op("snext"), imm(0), op("snext"))
                state["rk_moved"] = True

        for gi, n in enumerate(order):
            m = mf[n]
            slot = n % 4
            if m[0] == "S" and not self.ref_ready:
                emit_fillers(0, force_ref=True)                  # the S chain's first link reads the seed block: decision first
            self.wait_frag(slot)
            if m[0] == "S":
                ks = m[1]
                assert self.ref_ready
                p.mfma(vr(Smm, 16), self.F(slot), opr("q%d" % ks, 4), vr(R["SD"], 16) if ks == 0 else vr(Smm, 16))
            else:
                _, j, ds = m
                O = R["O0"] if ds == 0 else R["O1"]
                p.mfma(vr(O, 16), self.F(slot), vr(R["P"] + 4 * j, 4), imm(0) if (pv_first and j == 0) else vr(O, 16))
            issued.add(n)
            # the read that refills this ring slot: positions 0-3 fetch the fragment of position n + 4 of THIS iteration, 4-7 the
            # fragment of position n - 4 of the NEXT one
            if n < 4:
                m4 = mf[n + 4]
                if m4 is not None:
                    if m4[0] == "S":
                        assert not state["rk_moved"]
                        self.read(slot, R["rk"] + m4[1], s_half * 4096)
                    else:
                        assert not state["rv_moved"]
                        self.read(slot, R["rv"] + 2 * pv_half + m4[1], VT + m4[2] * 4096)
            addr_steps()
            if n >= 4 and nxt is not None:
                ns, npv, k_late = nxt
                pos = n - 4
                if pos % 2 == 0 and ns is not None and not k_late:
                    self.read(slot, R["rk"] + pos // 2, ns * 4096)
                if pos % 2 == 1 and npv is not None:
                    j, ds = (pos // 2) // 2, (pos // 2) % 2
                    assert state["rv_moved"] or not addr_update, "the next iteration's V fragments are read through the moved rv"
                    self.read(slot, R["rv"] + 2 * npv + j, VT + ds * 4096)
            emit_fillers((nf * (gi + 1) + gaps - 1) // gaps if gi + 1 < gaps else nf)
        # positions that do not exist in this iteration still have successors to prefetch (first tile: no P.V, last odd: no S)
        for n in (5, 7, 4, 6):
            if mf[n] is None and nxt is not None:
                ns, npv, k_late = nxt
                pos, slot = n - 4, n % 4
                issued.add(n - 4); issued.add(n)
                addr_steps()
                if pos % 2 == 0 and ns is not None and not k_late:
                    self.read(slot, R["rk"] + pos // 2, ns * 4096)
                if pos % 2 == 1 and npv is not None:
                    j, ds = (pos // 2) // 2, (pos // 2) % 2
                    self.read(slot, R["rv"] + 2 * npv + j, VT + ds * 4096)
        issued |= set(range(8))
        addr_steps()
        emit_fillers(nf)
        assert done_f == nf
        assert not addr_update or (state["rv_moved"] and (addr_update != "full" or state["rk_moved"]))
        if has_pv and not init:
            self.rescale_block()
        self.pending_slow = getattr(self, "pending_slow", [])
        if not init:
            self.pending_slow.append((slow, resume, Ssm))

    def dma_block(self):
        """request tile t+2 into the slot tile t-1 occupied (four 1-KiB pieces per wave: two of K, two of V^T), if there is one"""
        p = self.p
        skip = self.new_label("nodma")
        p.scmp("s_cmp_ge_u32", op("tdma"), op("nt"))
        p.branch("s_cbranch_scc1", skip)
        p.salu("s_add_u32", "m0", op("ldsw"), op("dslot"))
        p.salu("s_add_u32", op("koff"), op("koff"), imm(8192))
        p.dma(op("kvoff0"), opr("rsk", 4), op("koff"))
        p.salu("s_add_u32", "m0", "m0", imm(1024))
        p.salu("s_add_u32", op("voff"), op("voff"), imm(128))
        p.dma(op("kvoff1"), opr("rsk", 4), op("koff"))
        p.salu("s_add_u32", "m0", "m0", imm(VT - 1024))
        p.salu("s_add_u32", op("dslot"), op("dslot"), imm(SLOT))
        p.dma(op("vvoff0"), opr("rsv", 4), op("voff"))
        p.salu("s_add_u32", "m0", "m0", imm(1024))
        p.scmp("s_cmp_eq_u32", op("dslot"), imm(NSLOT * SLOT))
        p.dma(op("vvoff1"), opr("rsv", 4), op("voff"))
        p.salu("s_cselect_b32", op("dslot"), imm(0), op("dslot"))
        p.label(skip)
        p.salu("s_add_u32", op("tdma"), op("tdma"), imm(1))

    def flush_slow(self):
        for slow, resume, S in getattr(self, "pending_slow", []):
            self.slow_path(slow, resume, S)
        self.pending_slow = []

    # ---- the whole statement ------------------------------------------------------------------------------------------------------
    def build(self):
        p = self.p
        L_single, L_loop, L_last, L_end = (self.new_label(s_) for s_ in ("single", "loop", "last", "end"))
        # ---------------- prologue: tiles 0 and 1 requested, S(0) computed
        p.comment("prologue: DMA of tile 0 (slot 0) and, if it exists, tile 1 (slot 1)")
        p.salu("s_mov_b32", op("koff"), imm(0))
        p.salu("s_mov_b32", op("voff"), imm(0))
        p.salu("s_mov_b32", op("resc"), imm(0))
        p.salu("s_add_u32", "m0", op("ldsw"), imm(0))
        p.valu("v_mov_b32", v(R["l"]), imm(0))
        p.valu("v_mov_b32", v(R["ninf"]), lit(LIT_NINF))
        p.dma(op("kvoff0"), opr("rsk", 4), op("koff"))
        p.salu("s_add_u32", "m0", "m0", imm(1024))
        for k in range(4):
            p.valu("v_mov_b32", v(R["rk"] + k), op("off%d" % k))
        p.dma(op("kvoff1"), opr("rsk", 4), op("koff"))
        p.salu("s_add_u32", "m0", "m0", imm(VT - 1024))
        p.salu("s_mov_b32", op("snext"), imm(SLOT))
        p.dma(op("vvoff0"), opr("rsv", 4), op("voff"))
        p.salu("s_add_u32", "m0", "m0", imm(1024))
        p.salu("s_mov_b32", op("dslot"), imm(SLOT))
        p.dma(op("vvoff1"), opr("rsv", 4), op("voff"))
        p.salu("s_mov_b32", op("tdma"), imm(1))
        self.dma_block()                                         # tile 1 (tdma = 1 < nt), leaves tdma = 2, dslot = 2 slots
        # wait for tile 0 only: with a second tile in flight four younger pieces may stay outstanding
        L_one = self.new_label("one")
        L_go = self.new_label("go")
        p.scmp("s_cmp_eq_u32", op("nt"), imm(1))
        p.branch("s_cbranch_scc1", L_one)
        p.waitcnt(vm=4)
        p.branch("s_branch", L_go)
        p.label(L_one)
        p.waitcnt(vm=0)
        p.label(L_go)
        p.barrier()
        p.comment("S(0) = K(tile 0, half 0) . Q^T, and the first fragments of iteration 0")
        for ks in range(4):
            self.read(ks, R["rk"] + ks, 0)
        for ks in range(4):
            self.wait_frag(ks)
            p.mfma(vr(R["SA"], 16), self.F(ks), opr("q%d" % ks, 4), imm(0) if ks == 0 else vr(R["SA"], 16))
            if ks % 2 == 0:                                      # iteration 0 has no P.V: only its S fragments (ring slots 0, 2)
                self.read(ks, R["rk"] + ks // 2, 4096)
        p.nop(7)                                                 # S(0) is read by VALU at the head of iteration 0: the MFMA chain must have
        p.nop(7)                                                 # retired (12 wait states behind the last MFMA; its issue itself waits for the pipe)
        p.salu("s_sub_u32", op("tleft"), op("nt"), imm(1))
        p.scmp("s_cmp_eq_u32", op("tleft"), imm(0))
        p.branch("s_cbranch_scc1", L_single)

        # ---------------- first tile (of several): no P.V in its even iteration, unmasked
        lo0 = self.frag_state()
        self.iteration("tile 0, even: S(1), softmax(0) [init]", 0, 1, None, init=True, nxt=(0, 0, True), addr_update="full")
        self.iteration("tile 0, odd: S(2) from tile 1, P.V(0), softmax(1)", 1, 0, 0, pv_first=True, nxt=(1, 1, False), barrier_dma=True)
        p.salu("s_sub_u32", op("tleft"), op("tleft"), imm(1))
        p.scmp("s_cmp_eq_u32", op("tleft"), imm(0))
        p.branch("s_cbranch_scc1", L_last)
        # ---------------- steady state: one trip = one 64-key tile that has a successor
        lo_loop = self.frag_state()
        p.label(L_loop)
        self.iteration("tile t, even: S(2t+1), P.V(2t-1), softmax(2t)", 0, 1, 1, nxt=(0, 0, True), addr_update="full")
        self.iteration("tile t, odd: S(2t+2) from tile t+1, P.V(2t), softmax(2t+1)", 1, 0, 0, nxt=(1, 1, False), barrier_dma=True)
        assert self.frag_state() == lo_loop, (self.frag_state(), lo_loop)
        p.salu("s_sub_u32", op("tleft"), op("tleft"), imm(1))
        p.scmp("s_cmp_lg_u32", op("tleft"), imm(0))
        p.branch("s_cbranch_scc1", L_loop)
        # ---------------- last tile (after at least one other): masked softmax, no successor
        p.label(L_last)
        self.iteration("last tile, even: S(n-1), P.V(n-3), softmax(n-2) [masked]", 0, 1, 1, masked=True, kv_half="kvl0", nxt=(None, 0, False),
                       addr_update="rv")
        self.iteration("last tile, odd: P.V(n-2), softmax(n-1) [masked]", 1, None, 0, masked=True, kv_half="kvl1", nxt=(None, 1, False))
        self.final_pv(0)
        p.branch("s_branch", L_end)
        # ---------------- the only tile
        p.label(L_single)
        self.lds_order, self.landed = list(lo0[0]), set(lo0[1])
        self.iteration("only tile, even: S(1), softmax(0) [init, masked]", 0, 1, None, masked=True, init=True, kv_half="kvl0", nxt=(None, 0, False),
                       addr_update="rv")
        self.iteration("only tile, odd: P.V(0), softmax(1) [masked]", 1, None, 0, masked=True, pv_first=True, kv_half="kvl1", nxt=(None, 1, False))
        self.final_pv(0)
        p.branch("s_branch", L_end)
        self.flush_slow()
        p.label(L_end)
        p.nop(15)                                                # the caller reads O (VALU) right behind the statement
        if self.knock - TUNE_FLAGS:
            p.ins = self.apply_knock(p.ins)
        return p

    def apply_knock(self, ins):
        kn, out = self.knock, []
        soft_ops = {"v_max3_f32", "v_max_f32", "v_fmamk_f32", "v_exp_f32", "v_add_f32", self.cvt, "v_cndmask_b32", "v_subrev_u32", "v_mul_f32",
                    "v_sub_f32", "v_permlane32_swap_b32"}
        for d in ins:
            k = d["kind"]
            if "noexp" in kn and k == "valu" and d["op"] == "v_exp_f32":
                d = dict(d, op="v_mov_b32", trans=False)
            if "nosoftmax" in kn and ((k == "valu" and d["op"] in soft_ops) or k == "vcmp" or (k == "branch" and d["op"] == "s_cbranch_vccnz")):
                continue
            if "nomax" in kn and ((k == "valu" and d["op"] in ("v_max3_f32", "v_max_f32")) or (k == "vcmp" and d["op"] == "v_cmp_gt_f32")
                                  or (k == "branch" and d["op"] == "s_cbranch_vccnz")):
                continue
            if "nomfma" in kn and k == "mfma":
                continue
            if "nolds" in kn and (k == "ds_read" or (k == "waitcnt" and d["vm"] is None)):
                continue
            if "nodma" in kn and (k in ("dma", "barrier") or (k == "waitcnt" and d["vm"] is not None)):
                continue
            out.append(d)
        return out

    def final_pv(self, parity):
        """P.V of the last half-tile; its first two fragments were prefetched into ring slots 1, 3"""
        p = self.p
        p.comment("==== final P.V(n-1)")
        Prd = R["P"]
        # j = 1 fragments
        self.wait_frag(1)
        p.mfma(vr(R["O0"], 16), self.F(1), vr(Prd, 4), vr(R["O0"], 16))
        self.read(1, R["rv"] + 2 * 1 + 1, VT)
        self.wait_frag(3)
        p.mfma(vr(R["O1"], 16), self.F(3), vr(Prd, 4), vr(R["O1"], 16))
        self.read(3, R["rv"] + 2 * 1 + 1, VT + 4096)
        self.wait_frag(1)
        p.mfma(vr(R["O0"], 16), self.F(1), vr(Prd + 4, 4), vr(R["O0"], 16))
        self.wait_frag(3)
        p.mfma(vr(R["O1"], 16), self.F(3), vr(Prd + 4, 4), vr(R["O1"], 16))
        assert not self.lds_order and not self.landed


# ====================================================================================================================================
# printing
# ====================================================================================================================================
def fmt_operand(x):
    if isinstance(x, str):
        return x
    k = x[0]
    if k == "v":
        return "v%d" % x[1]
    if k == "vr":
        return "v[%d:%d]" % (x[1], x[1] + x[2] - 1)
    if k in ("op", "opr"):
        return "%%[%s]" % x[1]
    if k == "lit":
        return "0x%08x" % x[1]
    if k == "imm":
        return str(x[1])
    raise ValueError(x)


def to_asm(ins, mf="MF", cvt=None):
    """-> list of C string-literal lines"""
    out = []
    q = lambda s: '"%s\\n"' % s
    for d in ins:
        k = d["kind"]
        if k == "comment":
            out.append(q("; " + d["text"]))
        elif k == "label":
            out.append(q(d["name"] + ":"))
        elif k == "mfma":
            out.append('%s " %s, %s, %s, %s\\n"' % (mf, fmt_operand(d["dst"]), fmt_operand(d["a"]), fmt_operand(d["b"]), fmt_operand(d["c"])))
        elif k == "ds_read":
            out.append(q("ds_read_b128 %s, %s offset:%d" % (fmt_operand(d["dst"]), fmt_operand(d["addr"]), d["off"])))
        elif k == "valu":
            o, dst, src = d["op"], d["dst"], d["src"]
            if o == "v_cndmask_b32":
                out.append(q("v_cndmask_b32 %s, %s, %s, vcc" % (fmt_operand(dst), fmt_operand(src[0]), fmt_operand(src[1]))))
            elif o == "v_permlane32_swap_b32":
                out.append(q("v_permlane32_swap_b32 %s, %s" % (fmt_operand(dst), fmt_operand(src[0]))))
            else:
                out.append(q("%s %s, %s" % (o, fmt_operand(dst), ", ".join(fmt_operand(s_) for s_ in src))))
        elif k == "vcmp":
            out.append(q("%s vcc, %s, %s" % (d["op"], fmt_operand(d["a"]), fmt_operand(d["b"]))))
        elif k == "salu":
            out.append(q("%s %s, %s" % (d["op"], fmt_operand(d["dst"]), ", ".join(fmt_operand(s_) for s_ in d["src"]))))
        elif k == "scmp":
            out.append(q("%s %s, %s" % (d["op"], fmt_operand(d["a"]), fmt_operand(d["b"]))))
        elif k == "branch":
            out.append(q("%s %s" % (d["op"], d["target"])))
        elif k == "nop":
            out.append(q("s_nop %d" % d["n"]))
        elif k == "waitcnt":
            parts = []
            if d["vm"] is not None:
                parts.append("vmcnt(%d)" % d["vm"])
            if d["lgkm"] is not None:
                parts.append("lgkmcnt(%d)" % d["lgkm"])
            out.append(q("s_waitcnt " + " ".join(parts)))
        elif k == "barrier":
            out.append(q("s_barrier"))
        elif k == "dma":
            out.append(q("buffer_load_dwordx4 %s, %s, %s offen lds" % (fmt_operand(d["voff"]), fmt_operand(d["rsrc"]), fmt_operand(d["soff"]))))
        else:
            raise ValueError(k)
    return out


# ====================================================================================================================================
# static hazard checks (what the compiler's hazard recogniser would do for code it can see)
# ====================================================================================================================================
def check_hazards(ins):
    real = [d for d in ins if d["kind"] not in ("comment",)]
    n = len(real)

    def states(d):
        return d["n"] + 1 if d["kind"] == "nop" else (0 if d["kind"] == "label" else 1)

    def reads_writes(d):
        k = d["kind"]
        if k == "valu":
            rd = set().union(*[regs_of(s_) for s_ in d["src"]]) if d["src"] else set()
            wr = regs_of(d["dst"])
            if d["op"] == "v_permlane32_swap_b32":
                rd |= wr | regs_of(d["src"][0]); wr |= regs_of(d["src"][0])
            return rd, wr
        if k == "vcmp":
            return regs_of(d["a"]) | regs_of(d["b"]), set()
        if k == "ds_read":
            return regs_of(d["addr"]), regs_of(d["dst"])
        return set(), set()

    labels = {d["name"]: j for j, d in enumerate(real) if d["kind"] == "label"}

    def follow(j, ws, limit, visit, seen=None):
        """every instruction reachable from position j within `limit` wait states, along BOTH arms of conditional branches"""
        seen = set() if seen is None else seen
        while j < n and ws < limit:
            if (j, ws) in seen:
                return
            seen.add((j, ws))
            e = real[j]
            visit(e, ws)
            if e["kind"] == "branch":
                tgt = labels[e["target"]]
                if e["op"] == "s_branch":
                    j, ws = tgt, ws + 1
                    continue
                follow(tgt, ws + 1, limit, visit, seen)
            ws += states(e)
            j += 1

    for i, d in enumerate(real):
        if d["kind"] == "mfma":
            dst = regs_of(d["dst"])
            # (a) MFMA result -> VALU / LDS read or write of it: 12 wait states (gfx950, 8-pass XDL)

            def chk_a(e, ws, d=d, dst=dst):
                if e["kind"] in ("valu", "vcmp", "ds_read"):
                    rd, wr = reads_writes(e)
                    assert not ((rd | wr) & dst), ("MFMA result touched after %d wait states" % ws, d, e)
            follow(i + 1, 0, 12, chk_a)
        if d["kind"] == "valu":
            rd, wr = reads_writes(d)
            # (b) VALU write -> MFMA reads it as A / B / C: 2 wait states

            def chk_b(e, ws, d=d, wr=wr):
                if e["kind"] == "mfma":
                    used = regs_of(e["a"]) | regs_of(e["b"]) | regs_of(e["c"]) | regs_of(e["dst"])
                    assert not (used & wr), ("VALU result read by an MFMA too early", d, e)
            follow(i + 1, 0, 2, chk_b)
            # (c) transcendental result -> the next VALU must not read it
            if d.get("trans"):
                def chk_c(e, ws, d=d, wr=wr):
                    if e["kind"] in ("valu", "vcmp") and not e.get("trans"):
                        rd2, _ = reads_writes(e)
                        assert not (rd2 & wr), ("trans result read by the next VALU", d, e)
                follow(i + 1, 0, 1, chk_c)
        if d["kind"] == "salu" and d["dst"] == "m0" and i + 1 < n:
            assert real[i + 1]["kind"] != "dma", ("LDS-DMA right behind its M0 write", d)
        if d["kind"] == "barrier":
            # every barrier that publishes LDS-DMA data is preceded by the wait for the wave's own pieces
            prev = real[i - 1]
            assert prev["kind"] in ("waitcnt", "label"), ("barrier without a wait in front", prev)
    return True


OPERANDS_OUT = ['[o0] "=&{v[40:55]}"(o0)', '[o1] "=&{v[56:71]}"(o1)', '[lsum] "=&{v%d}"(lsum)' % R["l"],
                '[koff] "=&s"(koff)', '[voff] "=&s"(voff)', '[resc] "=&s"(resc)', '[snext] "=&s"(snext)', '[dslot] "=&s"(dslot)',
                '[tdma] "=&s"(tdma)', '[tleft] "=&s"(tleft)']
OPERANDS_IN = ['[q0] "v"(qf[0])', '[q1] "v"(qf[1])', '[q2] "v"(qf[2])', '[q3] "v"(qf[3])',
               '[off0] "v"(off[0])', '[off1] "v"(off[1])', '[off2] "v"(off[2])', '[off3] "v"(off[3])',
               '[kvoff0] "v"(kvoff[0])', '[kvoff1] "v"(kvoff[1])', '[vvoff0] "v"(vvoff[0])', '[vvoff1] "v"(vvoff[1])',
               '[limbase] "v"(limbase)', '[rsk] "s"(rsk)', '[rsv] "s"(rsv)', '[ldsw] "s"(ldsw)', '[nt] "s"(nt)',
               '[kvl0] "s"(kvl0)', '[kvl1] "s"(kvl1)']


def emit(fmt, var=0):
    g = Gen(fmt, VARIANTS[var] if var else ())
    prog = g.build()
    if not (g.knock - TUNE_FLAGS):
        check_hazards(prog.ins)
    lines = to_asm(prog.ins)
    clob = ", ".join('"v%d"' % r for r in range(FIXED_LO, FIXED_HI + 1) if not (40 <= r <= 71 or r == R["l"]))
    os.makedirs(OUTDIR, exist_ok=True)
    dst = os.path.join(OUTDIR, "attn_asm_%s%s.inc" % (fmt, "_v%d" % var if var else ""))
    with open(dst, "w") as f:
        f.write("// GENERATED by tools/gen_attn_asm.py -- do not edit; the schedule is documented there.\n")
        f.write("// Expects MF (MFMA mnemonic string literal) and the operands named below in scope.\n")
        # M0 is re-pointed for every LDS-DMA block; it is a RESERVED register (hipcc refuses it as a clobber), so the statement saves it in
        # a scalar of its own and restores it (+ the wait state an M0 write needs before the compiler's next LDS-DMA / movrel)
        f.write("{ int m0_keep_;\n")
        f.write("asm volatile(\n")
        f.write('    "s_mov_b32 %[m0k], m0\\n"\n')
        for l in lines:
            f.write("    " + l + "\n")
        f.write('    "s_mov_b32 m0, %[m0k]\\n"\n')
        f.write('    "s_nop 0\\n"\n')
        f.write("    : " + ",\n      ".join(OPERANDS_OUT + ['[m0k] "=&s"(m0_keep_)']) + "\n")
        f.write("    : " + ",\n      ".join(OPERANDS_IN) + "\n")
        f.write('    : "scc", "vcc", "memory", ' + clob + "); }\n")
    print("wrote", os.path.normpath(dst), len(lines), "lines")
    return prog


def emit_product():
    emit("bf16")
    emit("f16")


if __name__ == "__main__":
    what = sys.argv[1] if len(sys.argv) > 1 else "product"
    if what == "product":
        emit_product()
    elif what == "experiments":
        os.makedirs(OUTDIR, exist_ok=True)
        for var in sorted(VARIANTS):
            emit("bf16", var)
    elif what == "hashes":
        import hashlib, json, tempfile
        with tempfile.TemporaryDirectory() as tmp:
            OUTDIR = tmp
            emit_product()
            h = {f: hashlib.sha256(open(os.path.join(tmp, f), "rb").read()).hexdigest() for f in sorted(os.listdir(tmp))}
        dst = os.path.join(os.path.dirname(os.path.abspath(__file__)), "attn_asm_hashes.json")
        json.dump(h, open(dst, "w"), indent=1, sort_keys=True)
        print("wrote", dst, len(h), "entries")
    else:
        raise SystemExit("usage: gen_attn_asm.py [product|hashes]   (GEN_GEMM_ASM_OUT overrides the directory)")
