"""Builds libsylber_hip.so (hand-written HIP for gfx950) in-tree with hipcc.

hipcc cross-compiles without a GPU; the .so is git-ignored but travels to the GPU box with the
gpurun snapshot.  segment.hip is compiled with -ffp-contract=off: the segmenter must reproduce
numpy's unfused float32 arithmetic bit for bit.
"""
from __future__ import annotations

import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
# SYLBER_EXPERIMENTS=1: the timing-only kernels (knock-out loops, phase stamps; some store nothing) are compiled in and the
# result goes to ITS OWN file, so an experiments build can never be mistaken for / left behind as the product library.
EXPERIMENTS = bool(os.environ.get("SYLBER_EXPERIMENTS"))
# SYLBER_BUILD_VARIANT=name + SYLBER_EXTRA_CFLAGS="-D...": an A/B build with extra compiler flags into ITS OWN library
# (libsylber_hip_<name>.so, objects under build/<name>/): load it with SYLBER_HIP_LIB for a same-box comparison
VARIANT = os.environ.get("SYLBER_BUILD_VARIANT", "")
EXTRA_CFLAGS = os.environ.get("SYLBER_EXTRA_CFLAGS", "").split() if VARIANT else []
LIB = os.path.join(HERE, "libsylber_hip_%s.so" % VARIANT if VARIANT else ("libsylber_hip_exp.so" if EXPERIMENTS else "libsylber_hip.so"))
GEN_DIR = os.path.join(HERE, "build", "gen")           # generated inline-asm loops (never committed: tools/gen_gemm_asm.py writes them here)
GEN_EXP_DIR = os.path.join(HERE, "build", "gen_exp")   # the knock-out / timing variants of the experiments build: their own directory, so that
                                                       # they neither ship with the product snapshot nor age the product's objects
GENERATORS = [os.path.join(os.path.dirname(HERE), "tools", g) for g in ("gen_gemm_asm.py", "gen_attn_asm.py")]
SOURCES = ["api.hip", "gemm_bf16.hip", "frontend.hip", "attention.hip", "posconv.hip", "segment.hip", "fp32_path.hip", "ingest.hip", "gemm_mxfp8.hip", "downstream.hip", "gemm_rowln.hip", "gemm_asm.hip", "gemm_asm_f8.hip", "gemm_asm16.hip", "flac_host.hip"]
EXTRA = {"segment.hip": ["-ffp-contract=off"]}
# -fno-slp-vectorize: NO packed-fp32 VALU (v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32) anywhere in the library.  Measured on
# MI355X / ROCm 7.2 (profiles/r02_packed_f32_hazard.md): a wave running dependent packed-fp32 chains returns wrong values
# in lanes 48-63 of the LOW half of a result when MFMA waves of another kernel share its SIMD (two streams in flight).
# tests/test_abi.py checks the built objects for such instructions.
COMMON = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wno-unused-result", "-fno-slp-vectorize", "-fno-vectorize"]


def _hipcc() -> str:
    for cand in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if cand and (os.path.isabs(cand) and os.path.exists(cand) or not os.path.isabs(cand)):
            return cand
    return "hipcc"


def _newest_source() -> float:
    t = 0.0
    for root in (CSRC, os.path.join(os.path.dirname(HERE), "include")):
        for f in os.listdir(root):
            t = max(t, os.path.getmtime(os.path.join(root, f)))
    for g in GENERATORS:
        if not os.path.exists(g):
            raise FileNotFoundError("%s is missing: the inline-asm K loops are generated at build time (tools/ must ship with the package)" % g)
        t = max(t, os.path.getmtime(g))
    return max(t, os.path.getmtime(os.path.abspath(__file__)))


def generate(what=("product",), outdir: str = GEN_DIR) -> None:
    """run the loop generators; a file is only rewritten when its text changes (so that unchanged loops do not force recompiles)"""
    import tempfile
    os.makedirs(outdir, exist_ok=True)
    with tempfile.TemporaryDirectory() as tmp:
        for g in GENERATORS:
            for w in what:
                r = subprocess.run([sys.executable, g, w], capture_output=True, text=True, env=dict(os.environ, GEN_GEMM_ASM_OUT=tmp))
                if r.returncode != 0:            # e.g. a hazard-check assert of the generator: say which one and why
                    raise RuntimeError("%s %s failed (exit %d):\n%s" % (os.path.basename(g), w, r.returncode, (r.stderr or r.stdout)[-3000:]))
        for f in sorted(os.listdir(tmp)):
            new = open(os.path.join(tmp, f), "rb").read()
            dst = os.path.join(outdir, f)
            if not os.path.exists(dst) or open(dst, "rb").read() != new:
                with open(dst, "wb") as fh:
                    fh.write(new)


def build(force: bool = False, verbose: bool = False) -> str:
    if not force and os.path.exists(LIB) and os.path.getmtime(LIB) >= _newest_source():
        return LIB
    objdir = os.path.join(HERE, "build", VARIANT if VARIANT else ("exp" if EXPERIMENTS else "obj"))
    os.makedirs(objdir, exist_ok=True)
    hipcc = _hipcc()
    # the inline-asm K loops are generated at build time (33 k lines that used to be committed beside their generator)
    generate(("product",))
    exp_flags = ["-I", GEN_DIR]
    gen_dirs = [GEN_DIR]
    if EXPERIMENTS:
        generate(("experiments",), GEN_EXP_DIR)
        exp_flags += ["-I", GEN_EXP_DIR, "-DSYLBER_GEMM_ASM_EXPERIMENTS"]
        gen_dirs.append(GEN_EXP_DIR)

    # a source is recompiled when it, any header / generated loop beside it, the public headers or this file is newer
    shared = [os.path.getmtime(os.path.abspath(__file__))]
    for root in [CSRC, os.path.join(os.path.dirname(HERE), "include")] + gen_dirs:
        shared += [os.path.getmtime(os.path.join(root, f)) for f in os.listdir(root) if not f.endswith(".hip")]
    newest_shared = max(shared)

    def compile_one(src):
        obj = os.path.join(objdir, src.replace(".hip", ".o"))
        if not force and os.path.exists(obj) and os.path.getmtime(obj) >= max(newest_shared, os.path.getmtime(os.path.join(CSRC, src))):
            return obj
        extra = EXTRA.get(src, []) + exp_flags + EXTRA_CFLAGS
        cmd = [hipcc] + COMMON + extra + ["-c", os.path.join(CSRC, src), "-o", obj]
        if verbose:
            print(" ".join(cmd), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            errs = [ln for ln in r.stderr.splitlines() if "error" in ln]
            raise RuntimeError("hipcc failed for %s:\n%s\n...\n%s" % (src, "\n".join(errs[:20]), r.stderr[-1500:]))
        return obj

    with ThreadPoolExecutor(max_workers=min(6, os.cpu_count() or 1)) as ex:
        objs = list(ex.map(compile_one, SOURCES))
    cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError("link failed:\n" + r.stderr[-4000:])
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
