#!/usr/bin/env python3
"""Static VALU mix of the extraction kernels by measured issue class, as JSON.

On gfx950 a wave64 VALU instruction issues in 2 cycles only for a small set of opcodes (32-bit
add / sub / and / or / xor / mov / not / right shifts, f32 add / sub / mul, the 16-bit VOP2 forms)
and only without a scalar-register operand, SDWA or DPP; everything else (three-operand, packed,
compares, multiplies, left shifts, min / max, conversions ...) takes 4 (profiles/*valu_ops_stdout.txt,
tools/valu_ops.hip).  bench.py prices a kernel's measured SQ_INSTS_VALU with the mix printed here.

Usage: tools/isa_price.py > profiles/isa_mix.json   (compiles slam_amd/csrc/extract_kernels.hip to assembly)"""
import collections
import json
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FAST = {"v_xor_b32", "v_and_b32", "v_or_b32", "v_add_u32", "v_sub_u32", "v_subrev_u32", "v_mov_b32", "v_not_b32",
        "v_ashrrev_i32", "v_lshrrev_b32", "v_add_f32", "v_sub_f32", "v_mul_f32", "v_add_u16", "v_sub_u16",
        "v_max_u16", "v_min_u16", "v_max_i16", "v_min_i16", "v_mul_lo_u16", "v_lshlrev_b16", "v_lshrrev_b16",
        "v_ashrrev_i16", "v_subrev_u16"}


def mix(asm, kernel):
    out = {}
    for m in re.finditer(r"^(\w+):.*?\.end_amdhsa_kernel", asm, re.S | re.M):
        if kernel not in m.group(1):
            continue
        c = collections.Counter()
        for line in m.group(0).split("\n"):
            t = line.strip().split()
            if not line.startswith("\t") or not t or t[0][0] in ".;" or not t[0].startswith("v_"):
                continue
            base = re.sub(r"_(e32|e64|sdwa|dpp)$", "", t[0])
            ops = " ".join(t[1:]).split(";")[0]
            scalar = bool(re.search(r"\bs\d+|\bs\[|vcc|exec", ops))
            fast = base in FAST and not scalar and not t[0].endswith(("sdwa", "dpp"))
            c["fast" if fast else "slow"] += 1
        n = c["fast"] + c["slow"]
        out = {"valu_static": n, "fast_2_cycles": c["fast"], "slow_4_cycles": c["slow"],
               "cycles_per_inst": round((2 * c["fast"] + 4 * c["slow"]) / max(n, 1), 3)}
    return out


def main():
    with tempfile.TemporaryDirectory() as tmp:
        s = os.path.join(tmp, "k.s")
        subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-mllvm", "-amdgpu-mfma-vgpr-form", "-S",
                        "--cuda-device-only", "-o", s, os.path.join(ROOT, "slam_amd/csrc/extract_kernels.hip")],
                       check=True, stderr=subprocess.DEVNULL)
        asm = open(s).read()
    rec = {"source": "static instruction mix of slam_amd/csrc/extract_kernels.hip (hipcc -O3, gfx950), issue classes "
                     "from tools/valu_ops.hip",
           "kernels": {k: mix(asm, k + "_kernel") for k in ("fast_nms", "describe", "select_compact")}}
    json.dump(rec, sys.stdout, indent=1)
    print()


if __name__ == "__main__":
    main()
