"""Disassembles the gfx950 code objects inside libsylber_hip.so and reports instructions the library bans.

Packed-fp32 VALU arithmetic (v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32) is banned: measured on MI355X / ROCm 7.2, a wave
running dependent packed-fp32 chains returned wrong values in lanes 48-63 of the low half of a result whenever MFMA
waves of ANOTHER kernel shared its SIMD (two batches in flight); see profiles/r02_packed_f32_hazard.md."""
import os
import re
import shutil
import subprocess
import sys
import tempfile

LLVM = "/opt/rocm/lib/llvm/bin"
BANNED = re.compile(r"\bv_pk_(fma|mul|add|min|max)_f32\b")


def device_disassembly(lib_path: str) -> str:
    tmp = tempfile.mkdtemp(prefix="sylisa_")
    try:
        dst = os.path.join(tmp, "lib.so")
        shutil.copy(lib_path, dst)
        subprocess.run([os.path.join(LLVM, "llvm-objdump"), "--offloading", dst], capture_output=True, text=True, check=True)
        out = []
        for f in sorted(os.listdir(tmp)):
            if "amdgcn" in f:
                r = subprocess.run([os.path.join(LLVM, "llvm-objdump"), "-d", os.path.join(tmp, f)], capture_output=True, text=True,
                                   check=True)
                out.append(r.stdout)
        text = "\n".join(out)
        # an empty disassembly (changed tool output naming, missing tool, other ROCm layout) must not read as "clean"
        if len(re.findall(r"\bv_mfma_f32_32x32x16_bf16\b", text)) < 100 or "s_endpgm" not in text:
            raise RuntimeError("check_isa: the gfx950 code objects of %s were not disassembled (no MFMA / s_endpgm found)" % lib_path)
        return text
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def banned_instructions(lib_path: str):
    counts = {}
    for m in BANNED.finditer(device_disassembly(lib_path)):
        counts[m.group(0)] = counts.get(m.group(0), 0) + 1
    return counts


if __name__ == "__main__":
    here = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    c = banned_instructions(sys.argv[1] if len(sys.argv) > 1 else os.path.join(here, "sylber_amd", "libsylber_hip.so"))
    print(c or "no banned instructions")
    sys.exit(1 if c else 0)
