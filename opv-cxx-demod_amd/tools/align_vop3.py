#!/usr/bin/env python3
"""align_vop3.py in.s out.s — keep 8-byte instructions of a gfx950 kernel on 8-byte addresses.

Why: a LONE wave on a SIMD (the MSK front-end's regime: one wave per stream) pays about one cycle more for every
8-byte instruction (VOP3, DPP, SDWA, DS, FLAT ...) that starts at an address of 4 mod 8
(scripts/microbench/loop_align.hip: 4.5 vs 5.5 cycles per dependent v_fma_f64, period 8 bytes). hipcc shrinks every
VOP1 / VOP2 / VOPC instruction it can to its 4-byte `_e32` encoding, so the parity flips all over a loop body: in the
front-end's symbol loop 46 % of the 8-byte instructions sat on the wrong parity.

What: device assembly in (`hipcc --cuda-device-only -S`), device assembly out. The input is assembled once to learn every
instruction's address and size (llvm-objdump). Then, function by function, runs of 4-byte instructions get ONE of their
shrinkable members re-encoded as `_e64` (same operation, same operands, 8 bytes) where that lowers the number of 8-byte
instructions on the wrong parity (a dynamic programme over the function: runs without a shrinkable member - s_nop,
s_waitcnt, branches only - pass their parity on). No instruction is added, none is moved. The output is assembled again as a check; an `_e64` form the assembler rejects is
blacklisted and the pass repeated.
"""
import re
import subprocess
import sys
import tempfile
from pathlib import Path

import os

# tool chain and target come from the Makefile (LLVMBIN / ARCH in the environment), not from this file
LLVM = Path(os.environ.get("LLVMBIN", "/opt/rocm/lib/llvm/bin"))
TARGET = ["-target", "amdgcn-amd-amdhsa", "-mcpu=" + os.environ.get("ARCH", "gfx950")]


def assemble(src, obj):
    return subprocess.run([str(LLVM / "clang"), "-x", "assembler", *TARGET, "-c", str(src), "-o", str(obj)],
                          capture_output=True, text=True)


def disassemble(obj):
    """{function: [(address, size, mnemonic)]} in address order"""
    out = subprocess.run([str(LLVM / "llvm-objdump"), "-d", str(obj)], capture_output=True, text=True, check=True).stdout
    funcs, cur = {}, None
    for line in out.split("\n"):
        m = re.match(r"^[0-9a-f]+ <([^>]+)>:", line)
        if m:
            cur = funcs.setdefault(m.group(1), [])
            continue
        m = re.match(r"^\s+(\S+)(.*?)//\s*([0-9A-F]+):", line)
        if m and cur is not None:
            cur.append([int(m.group(3), 16), 0, m.group(1), " ".join(m.group(2).split())])
    for ins in funcs.values():
        for a, b in zip(ins, ins[1:]):
            a[1] = b[0] - a[0]
        if ins:
            ins[-1][1] = 4                      # s_endpgm / trailing s_nop
    return funcs


def instruction_lines(lines):
    """{function: [line index]}: the lines of each function that assemble to one instruction"""
    funcs, cur = {}, None
    for i, raw in enumerate(lines):
        s = raw.strip()
        m = re.match(r"^([A-Za-z_][\w$.]*):", s)
        if m and not s.startswith(".L"):
            cur = funcs.setdefault(m.group(1), [])
            continue
        if s.startswith(".Lfunc_end"):
            cur = None
            continue
        if cur is None or not s or s[0] in ";.#" or s.endswith(":"):
            continue
        if re.match(r"^\.?L[\w$.]*:", s):
            continue
        cur.append(i)
    return funcs


def mnemonic(line):
    return line.strip().split()[0]


def widen(line):
    """the same instruction in its 8-byte VOP3 encoding"""
    m = re.match(r"^(\s*)(\S+)(.*)$", line)
    mn = m.group(2)
    mn = mn[:-4] + "_e64" if mn.endswith("_e32") else mn + "_e64"
    return m.group(1) + mn + m.group(3)


def flexible(src_mn, dis_mn, size):
    if size != 4 or not dis_mn.endswith("_e32"):
        return False
    if not dis_mn.startswith("v_"):
        return False
    if src_mn.endswith("_e64") or "_dpp" in src_mn or "_sdwa" in src_mn:
        return False
    return True


def base(mn):
    return re.sub(r"_e(32|64)$", "", mn)


def same_instructions(ins, idx, lines):
    return all(base(mnemonic(lines[i])) == base(x[2]) for x, i in zip(ins, idx))


def plan(ins, lines_idx, lines, banned):
    """indices (into ins) of the instructions to widen: the function as alternating runs of 4-byte instructions and
    blocks of 8-byte ones; a run with a shrinkable member may or may not add 4 bytes; dynamic programme over the parity
    (0 / 4) at which each block starts, cost = number of 8-byte instructions that start at 4 mod 8"""
    segs = []                                   # [first 8-byte index or None, n8, flexible member or None, bytes of the run]
    run_bytes, flex = 0, None
    parity0 = ins[0][0] % 8 if ins else 0
    k = 0
    while k < len(ins):
        size = ins[k][1]
        if size % 8:                            # 4-byte (or 12-byte) instruction: part of a run
            run_bytes += size
            if size == 4 and k not in banned and flexible(mnemonic(lines[lines_idx[k]]), ins[k][2], size):
                flex = k
            k += 1
            continue
        n8 = 0
        while k < len(ins) and ins[k][1] % 8 == 0:
            n8 += ins[k][1] // 8
            k += 1
        segs.append((n8, flex, run_bytes))
        run_bytes, flex = 0, None
    INF = 1 << 30
    cost = {parity0: 0}
    back = []
    for n8, flex, rb in segs:
        nxt, choice = {}, {}
        for p, c in cost.items():
            for add in ((0, 4) if flex is not None else (0,)):
                q = (p + rb + add) % 8
                cc = c + (n8 if q == 4 else 0)
                if cc < nxt.get(q, INF):
                    nxt[q] = cc
                    choice[q] = (p, add)
        cost = nxt
        back.append(choice)
    if not cost:
        return []
    q = min(cost, key=cost.get)
    flips = []
    for (n8, flex, rb), choice in zip(reversed(segs), reversed(back)):
        p, add = choice[q]
        if add:
            flips.append(flex)
        q = p
    return flips


def signature(x):
    """what must survive the pass: the operation and its operands (branch targets are addresses: they move)"""
    mn = base(x[2])
    return (mn, "" if mn.startswith(("s_branch", "s_cbranch", "s_call")) else x[3])


def verify(before, after, flipped):
    """exit non-zero unless every function the pass touched still holds the same instruction and operand sequence"""
    for f in flipped:
        a, b = [signature(x) for x in before[f]], [signature(x) for x in after.get(f, [])]
        while a and a[-1][0] == "s_nop":       # padding behind s_endpgm up to the next function is not code
            a.pop()
        while b and b[-1][0] == "s_nop":
            b.pop()
        if a != b:
            k = next((i for i, (x, y) in enumerate(zip(a, b)) if x != y), min(len(a), len(b)))
            sys.exit(f"align_vop3: {f}: re-assembled code differs from the input at instruction {k}: "
                     f"{a[k] if k < len(a) else None} vs {b[k] if k < len(b) else None}")


def stats(ins):
    eight = [x for x in ins if x[1] == 8]
    return len(eight), sum(1 for x in eight if x[0] % 8 == 4)


def main():
    src, dst = Path(sys.argv[1]), Path(sys.argv[2])
    lines = src.read_text().split("\n")
    with tempfile.TemporaryDirectory() as td:
        obj = Path(td) / "a.o"
        r = assemble(src, obj)
        if r.returncode:
            sys.exit("align_vop3: the input does not assemble:\n" + r.stderr)
        before = disassemble(obj)
        where = instruction_lines(lines)
        banned = {f: set() for f in before}
        for attempt in range(40):
            out = list(lines)
            flipped = {}
            for f, ins in before.items():
                idx = where.get(f)
                if idx is None or len(idx) > len(ins) or not same_instructions(ins, idx, lines):
                    if attempt == 0 and len(ins) > 200:
                        print(f"align_vop3: WARNING: {f} ({len(ins)} instructions) does not match its text 1:1 and is left unaligned", file=sys.stderr)
                    continue                    # not a function of this file's text (or no 1:1 match): left untouched
                ins = ins[:len(idx)]            # (behind s_endpgm the object carries padding up to the next function)
                fl = plan(ins, idx, lines, banned[f])
                for k in fl:
                    out[idx[k]] = widen(lines[idx[k]])
                flipped[f] = fl
            tmp = Path(td) / "b.s"
            tmp.write_text("\n".join(out))
            r = assemble(tmp, obj)
            if r.returncode == 0:
                break
            bad = {int(m.group(1)) - 1 for m in re.finditer(r"b\.s:(\d+):\d+: error", r.stderr)}
            if not bad:
                sys.exit("align_vop3: the output does not assemble:\n" + r.stderr)
            for f, fl in flipped.items():
                for k in fl:
                    if where[f][k] in bad:
                        banned[f].add(k)
        else:
            sys.exit("align_vop3: no assembling output after 40 attempts")
        after = disassemble(obj)
        verify(before, after, flipped)
        dst.write_text("\n".join(out))
        for f in before:
            if f in flipped:
                n0, m0 = stats(before[f])
                n1, m1 = stats(after[f])
                if n0 > 200:
                    print(f"align_vop3: {f}: {len(flipped[f])} instructions widened, misaligned 8-byte instructions {m0} -> {m1} of {n1}")


if __name__ == "__main__":
    main()
