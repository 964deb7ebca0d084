#!/usr/bin/env python3
"""Lint for the kernels that request scalar loads from inline asm and wait for them in a later asm statement.

The compiler does not know that the registers such a load names are written when the data arrives rather than when the
instruction issues: to it the destination holds its value from the asm statement on, and is free for other use from
the last statement that mentions it.  On a path where the destination is never mentioned again before the wait, the
register allocator may hand the registers to something else while the load is still in flight, and the data then lands
on top of that (a mask, a pointer): wrong sums at best, a memory fault at worst.  The same goes for a copy or a spill
of the destination made before the wait.

This walks the control-flow graph of every kernel in the device assembly of a source file and reports any instruction
that mentions (reads or writes) a scalar register that may still be the target of an asm-issued s_load on some path
reaching it, i.e. with no `s_waitcnt ... lgkmcnt(0)` in between.

    python tools/check_smem_inflight.py raytracer_amd/csrc/pb_render.hip [more.hip ...]

A `.s` file is read as it is (device assembly, as `hipcc -S --cuda-device-only` writes it).  Exit status 1 if anything
is reported.  raytracer_amd/csrc/Makefile runs it on pb_render.hip's assembly before it makes the object.
"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FLAGS = ["-O3", "-std=c++17", "--offload-arch=gfx950", "-fPIC", "-ffp-contract=off", "-fno-slp-vectorize",
         "-I" + os.path.join(ROOT, "include"), "-S", "--cuda-device-only"]

REG_RANGE = re.compile(r"\bs\[(\d+):(\d+)\]")
REG_ONE = re.compile(r"\bs(\d+)\b")
LABEL = re.compile(r"^(\.LBB\d+_\d+):")


def sregs(text):
    out = set()
    for a, b in REG_RANGE.findall(text):
        out.update(range(int(a), int(b) + 1))
    for a in REG_ONE.findall(REG_RANGE.sub(" ", text)):
        out.add(int(a))
    return out


def kernels(asm):
    """(name, lines) of every function body in the assembly text"""
    lines = asm.split("\n")
    i = 0
    while i < len(lines):
        m = re.match(r"^(_Z\w+):\s*(;.*)?$", lines[i])
        if m:
            j = i + 1
            while j < len(lines) and not lines[j].startswith(".Lfunc_end"):
                j += 1
            yield m.group(1), lines[i + 1:j]
            i = j
        i += 1


def check(name, body):
    # basic blocks: a label starts one, a branch ends one
    blocks = []          # [label or None, [(lineno, text, in_asm)]]
    cur = [None, []]
    blocks.append(cur)
    in_asm = False
    for n, raw in enumerate(body, 1):
        text = raw.split(";")[0].strip() if ";;#" not in raw else raw.strip()
        if ";;#ASMSTART" in raw:
            in_asm = True
            continue
        if ";;#ASMEND" in raw:
            in_asm = False
            continue
        m = LABEL.match(raw)
        if m:
            cur = [m.group(1), []]
            blocks.append(cur)
            continue
        if not text or text.startswith("."):
            continue
        cur[1].append((n, text, in_asm))
        op = text.split()[0]
        if op.startswith("s_cbranch") or op == "s_branch" or op == "s_endpgm" or op.startswith("s_setpc"):
            cur = [None, []]
            blocks.append(cur)
    index = {b[0]: k for k, b in enumerate(blocks) if b[0]}
    succ = []
    for k, (label, ins) in enumerate(blocks):
        s = []
        last = ins[-1][1] if ins else ""
        op = last.split()[0] if last else ""
        if op == "s_branch":
            s.append(index[last.split()[1]])
        elif op.startswith("s_cbranch"):
            s.append(index[last.split()[1]])
            if k + 1 < len(blocks):
                s.append(k + 1)
        elif op == "s_endpgm" or op.startswith("s_setpc"):
            pass
        elif k + 1 < len(blocks):
            s.append(k + 1)
        succ.append(s)
    entry = [frozenset() for _ in blocks]
    reports = {}

    def run(k, state, collect):
        state = set(state)
        for n, text, asm in blocks[k][1]:
            op = text.split()[0]
            if op == "s_waitcnt":
                if "lgkmcnt(0)" in text:
                    state.clear()
                continue
            if asm and op.startswith("s_load"):
                dst = text.split(None, 1)[1].split(",")[0]
                hit = state & (sregs(text) - sregs(dst))
                if hit and collect:
                    reports[n] = (text, sorted(hit))
                state |= sregs(dst)
                continue
            hit = state & sregs(text)
            if hit and collect:
                reports[n] = (text, sorted(hit))
        return frozenset(state)

    work = list(range(len(blocks)))
    while work:
        k = work.pop()
        out = run(k, entry[k], False)
        for s in succ[k]:
            merged = entry[s] | out
            if merged != entry[s]:
                entry[s] = merged
                work.append(s)
    for k in range(len(blocks)):
        run(k, entry[k], True)
    return reports


def main(argv):
    bad = 0
    for src in argv:
        if src.endswith(".s"):
            asm = open(src).read()
        else:
            with tempfile.TemporaryDirectory() as tmp:
                out = os.path.join(tmp, "dev.s")
                subprocess.run(["/opt/rocm/bin/hipcc"] + FLAGS + ["-o", out, src], check=True, stderr=subprocess.DEVNULL)
                asm = open(out).read()
        n_kernels = 0
        for name, body in kernels(asm):
            n_kernels += 1
            for line, (text, regs) in sorted(check(name, body).items()):
                bad += 1
                print(f"{src}: {name}: line {line} of the kernel: `{text}` touches s{regs} while an asm-issued load may be in flight")
        print(f"{src}: {n_kernels} kernels checked")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main(sys.argv[1:]))
