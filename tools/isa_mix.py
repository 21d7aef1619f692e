"""Instruction mix per basic block of one kernel in a hipcc --save-temps .s file.

usage: python tools/isa_mix.py file.s kernel_substring [min_block_len]
"""
import collections
import re
import sys


def main():
    path, key = sys.argv[1], sys.argv[2]
    minlen = int(sys.argv[3]) if len(sys.argv) > 3 else 40
    lines = open(path).read().split("\n")
    start = [i for i, l in enumerate(lines) if l.startswith("_Z") and key in l.split(":")[0]][0]
    end = [i for i, l in enumerate(lines) if i > start and "s_endpgm" in l][0]
    blocks = [["entry", []]]
    for l in lines[start + 1:end]:
        m = re.match(r"^(\.LBB\d+_\d+):", l)
        if m:
            blocks.append([m.group(1), []])
            continue
        s = l.strip()
        if not s or s[0] in ";.":
            continue
        blocks[-1][1].append(s.split()[0])
    for name, ins in blocks:
        c = collections.Counter()
        for i in ins:
            if i.startswith("v_mfma"):
                c["mfma"] += 1
            elif i.startswith("v_accvgpr"):
                c["acc"] += 1
            elif i.startswith("v_"):
                c["valu"] += 1
            elif i.startswith("ds_"):
                c["ds"] += 1
            elif i.startswith("s_nop"):
                c["nop"] += 1
            elif i.startswith("s_waitcnt"):
                c["wait"] += 1
            elif i.startswith("s_"):
                c["salu"] += 1
            else:
                c["mem"] += 1
        if len(ins) >= minlen:
            print(name, len(ins), dict(c))


main()
