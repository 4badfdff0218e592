#!/usr/bin/env python3
"""Static check of the gfx950 ISA hipcc emits for hand-issued asynchronous LDS reads.

gfx950 has no VGPR interlock for an LDS read that is still in flight: an instruction that touches the destination of a
`ds_read*` before an `s_waitcnt lgkmcnt(0)` sees stale bits.  For reads the COMPILER issues it places the wait itself; for the
reads our kernels issue from inline asm (rzcc.hip: the scan kernel's `ds_read2_b64` tile fetch, the encoder's filter wave) the
compiler does not know the asm is asynchronous, so the source must tie the destinations to the wait (see `landed()` there) -- and
this script checks the result: per kernel, a forward may-analysis over the control-flow graph of the emitted assembly; state = the
set of VGPRs written by an inline-asm LDS read (the `;;#ASMSTART` ... `;;#ASMEND` regions) that has not been followed by an
`s_waitcnt` with `lgkmcnt(0)` on SOME path; any instruction that reads or writes such a register is a finding.

    python tools/check_isa_hazards.py build_dev/isa/rzcc.s [kernel-name regex]     exit 1 on findings
    python tools/check_isa_hazards.py --emit rzcc.hip                              (re)generate build_dev/isa/rzcc.s first

ADVICE r5 (medium): rzcc_scan_kernel copied `ds_read2_b64` destinations with `v_mov_b64` ahead of the wait (50 sites)."""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REG = re.compile(r"\bv(\d+)\b|\bv\[(\d+):(\d+)\]")
LABEL = re.compile(r"^([.\w$]+):")
BRANCH = re.compile(r"^\s*(s_branch|s_cbranch_\w+)\s+([.\w$]+)")


def regs(text):
    out = set()
    for m in REG.finditer(text):
        if m.group(1) is not None:
            out.add(int(m.group(1)))
        else:
            out.update(range(int(m.group(2)), int(m.group(3)) + 1))
    return out


def emit(src):
    out = os.path.join(ROOT, "build_dev", "isa")
    os.makedirs(out, exist_ok=True)
    dst = os.path.join(out, os.path.splitext(src)[0] + ".s")
    path = os.path.join(ROOT, "haghighatshoarmuir2024_amd", "csrc", src)
    deps = [path, os.path.join(os.path.dirname(path), "micloc_internal.h")]
    if os.path.exists(dst) and os.path.getmtime(dst) >= max(os.path.getmtime(d) for d in deps):
        return dst
    extra = ["-mllvm", "-amdgpu-mfma-vgpr-form"] if src == "xylo.hip" else []
    subprocess.check_call(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-ffp-contract=off", "-fno-fast-math",
                           "--cuda-device-only", "-S"] + extra + ["-o", dst, path], stderr=subprocess.DEVNULL)
    return dst


def kernels(text):
    """(name, [lines]) per function of the assembly: from its label to the `.end` / size directive behind it."""
    cur, body = None, []
    for ln in text.split("\n"):
        m = re.match(r"^(_Z\w+|\w+):\s*(;.*)?$", ln)
        if m and not ln.startswith(".L"):
            if cur:
                yield cur, body
            cur, body = m.group(1), []
            continue
        if cur is not None:
            if ln.startswith(".Lfunc_end") or ln.strip().startswith(".section"):
                yield cur, body
                cur, body = None, []
            else:
                body.append(ln)
    if cur:
        yield cur, body


def check_kernel(name, lines):
    """Findings [(line index, text, registers)] of one function."""
    # instructions with their block structure
    blocks, order, cur = {}, [], "<entry>"
    blocks[cur] = []
    order.append(cur)
    in_asm = False
    for i, raw in enumerate(lines):
        ln = raw.split(";")[0].rstrip() if not raw.lstrip().startswith(";;#ASM") else raw.strip()
        if ln.startswith(";;#ASMSTART"):
            in_asm = True
            continue
        if ln.startswith(";;#ASMEND"):
            in_asm = False
            continue
        m = LABEL.match(ln)
        if m:
            cur = m.group(1)
            if cur not in blocks:
                blocks[cur] = []
                order.append(cur)
            continue
        t = ln.strip()
        if not t or t.startswith("."):
            continue
        blocks[cur].append((i, t, in_asm))
        if BRANCH.match(t) or t.startswith("s_endpgm") or t.startswith("s_setpc"):
            cur = f"<after {i}>"  # a branch ends its basic block: the fall-through path starts a new one
            blocks[cur] = []
            order.append(cur)
    succ = {}
    for bi, b in enumerate(order):
        s = set()
        ins = blocks[b]
        falls = True
        for (_, t, _) in ins:
            m = BRANCH.match(t)
            if m:
                s.add(m.group(2))
                if m.group(1) == "s_branch":
                    falls = False
            if t.startswith("s_endpgm") or t.startswith("s_setpc"):
                falls = False
        if falls and bi + 1 < len(order):
            s.add(order[bi + 1])
        succ[b] = [x for x in s if x in blocks]

    def transfer(b, state, report=None):
        st = set(state)
        for (i, t, in_asm) in blocks[b]:
            op = t.split()[0]
            if op == "s_waitcnt":
                if "lgkmcnt(0)" in t or re.fullmatch(r"s_waitcnt\s+0(x0+)?", t):
                    st = set()
                continue
            if in_asm and op.startswith("ds_read"):
                operands = t[len(op):].split(",")
                st |= regs(operands[0])  # the destination becomes pending (its address operand is read at issue: fine)
                continue
            if st:
                hit = regs(t[len(op):]) & st
                if hit and report is not None:
                    report.append((i, t, sorted(hit)))
        return st

    inn = {b: set() for b in order}
    work = list(order)
    while work:
        b = work.pop(0)
        out = transfer(b, inn[b])
        for s in succ[b]:
            if not out <= inn[s]:
                inn[s] |= out
                if s not in work:
                    work.append(s)
    findings = []
    for b in order:
        transfer(b, inn[b], findings)
    return findings


def check_file(path, pattern="."):
    pat = re.compile(pattern)
    text = open(path).read()
    total, seen, with_asm = [], 0, 0
    for name, body in kernels(text):
        if not pat.search(name):
            continue
        seen += 1
        if not any("ds_read" in ln for ln in body):
            continue
        if any(ln.strip().startswith(";;#ASMSTART") for ln in body):
            with_asm += 1
        for (i, t, r) in check_kernel(name, body):
            total.append((name, i, t, r))
    return total, seen, with_asm


def main(argv):
    if argv and argv[0] == "--emit":
        path = emit(argv[1])
        argv = [path] + argv[2:]
    path = argv[0]
    findings, seen, with_asm = check_file(path, argv[1] if len(argv) > 1 else ".")
    for (name, i, t, r) in findings[:40]:
        print(f"{name}: line +{i}: `{t}` touches v{r} while an inline-asm LDS read of it may be in flight")
    print(f"{os.path.relpath(path, ROOT)}: {seen} functions, {with_asm} with hand-issued LDS reads, {len(findings)} findings")
    return 1 if findings else 0


if __name__ == "__main__":
    sys.exit(main(sys.argv[1:]))
