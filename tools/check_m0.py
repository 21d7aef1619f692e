#!/usr/bin/env python3
"""Checks the built kernels for uses of M0 that the hand-written LDS-DMA code does not know of.

wn_dma.h changes M0 in inline assembly that the compiler cannot be told about (M0 is a reserved
register: it is refused in clobber lists).  That is safe as long as the compiler itself never
uses M0 in those kernels.  This script disassembles the device code of the object files and lists
every instruction that names m0; anything but `s_mov_b32 m0, <sgpr>` / `s_mov_b32 <sgpr>, m0`
(ours) is an error.  The DMA loads (`global_load_lds_*`) read M0 implicitly and do not name it.

usage: tools/check_m0.py [object files]     (default: wavenet_amd/csrc/wn_detect.o wn_sync.o)
"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LLVM = "/opt/rocm/lib/llvm/bin"
# implicit users of M0 on gfx9: indexed moves, lane access through M0, GWS/ordered-count and
# interpolation instructions, s_sendmsg; none of them has a place in these kernels
IMPLICIT = re.compile(r"\b(v_movrel|s_movrel|v_interp|ds_gws|ds_ordered_count|s_sendmsg|v_readlane_b32 \S+ \S+ m0|v_writelane_b32 \S+ \S+ m0)")
OURS = re.compile(r"^s_mov_b32 (m0, s\d+|s\d+, m0)$")


def device_asm(obj):
    with tempfile.TemporaryDirectory() as tmp:
        fat = os.path.join(tmp, "fat.bin")
        subprocess.run([os.path.join(LLVM, "llvm-objcopy"), "-O", "binary", "--only-section=.hip_fatbin", obj, fat],
                       check=True)
        subprocess.run([os.path.join(LLVM, "clang-offload-bundler"), "--unbundle", "--type=o",
                        "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", "--input=" + fat,
                        "--output=" + os.path.join(tmp, "dev.o")], check=True)
        return subprocess.run([os.path.join(LLVM, "llvm-objdump"), "-d", "--no-show-raw-insn", os.path.join(tmp, "dev.o")],
                              check=True, capture_output=True, text=True).stdout


def check(obj):
    bad, kernel, ours = [], "?", 0
    for line in device_asm(obj).splitlines():
        m = re.match(r"^[0-9a-f]+ <(.*)>:$", line)
        if m:
            kernel = m.group(1)
            continue
        ins = line.split("//")[0].strip()
        if not ins:
            continue
        if re.search(r"\bm0\b", ins):
            if OURS.match(ins):
                ours += 1
            else:
                bad.append((kernel, ins))
        elif IMPLICIT.search(ins):
            bad.append((kernel, ins))
    return ours, bad


def main(argv):
    objs = argv or [os.path.join(ROOT, "wavenet_amd", "csrc", f) for f in ("wn_detect.o", "wn_sync.o")]
    rc = 0
    for obj in objs:
        ours, bad = check(obj)
        print("%s: %d moves to or from m0 by the DMA code, %d other uses" % (os.path.basename(obj), ours, len(bad)))
        for kernel, ins in bad[:20]:
            print("   %s: %s" % (kernel[:60], ins))
        rc |= bool(bad)
    return rc


if __name__ == "__main__":
    sys.exit(main(sys.argv[1:]))
