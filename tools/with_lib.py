"""Development only: run a script against ANOTHER build of libsylber_hip.so (same-box A/B of two builds, or the experiments build with its
timing kernels).  The product loader (sylber_amd/_lib.py) honours no environment variable; this is the one way to swap the library.

    python tools/with_lib.py ref bench.py --no-api ...          -> sylber_amd/libsylber_hip_ref.so
    python tools/with_lib.py exp tools/gemm_asm_trace.py        -> sylber_amd/libsylber_hip_exp.so (SYLBER_EXPERIMENTS=1 python sylber_amd/build.py)
    python tools/with_lib.py /path/to/lib.so script.py ...
"""
import os
import runpy
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

if __name__ == "__main__":
    if len(sys.argv) < 3:
        raise SystemExit(__doc__)
    which, script = sys.argv[1], sys.argv[2]
    path = which if os.path.sep in which or which.endswith(".so") else os.path.join(ROOT, "sylber_amd", "libsylber_hip_%s.so" % which)
    if not os.path.exists(path):
        raise SystemExit("with_lib.py: %s does not exist" % path)
    from sylber_amd import _lib
    _lib.use_library(path)
    os.environ["SYLBER_DEV_LIB"] = path                     # informational (tools that label their output); nothing loads from it
    sys.argv = [script] + sys.argv[3:]
    sys.path.insert(0, os.path.dirname(os.path.abspath(script)))
    runpy.run_path(script, run_name="__main__")
