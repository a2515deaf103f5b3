#!/usr/bin/env python3
"""Static check of the one-wave-per-window kernels' inline-assembly MFMAs (posterior_wave_impl.h).

hipcc's hazard recogniser does not look inside inline assembly, so it never inserts the wait states a
v_mfma_f64_16x16x4_f64 result needs before anything but the SrcC of the next MFMA on the same registers may
touch it (19 wait states; the kernel's wave_settle supplies them at the end of every row loop).  What the source
cannot rule out is register-allocator code - a spill, a copy - that lands between the asm MFMAs and touches one of
their destination AGPRs.  This script proves its absence on the generated ISA:

    hipcc -O3 ... -S --cuda-device-only posterior_wave_nt.hip -o wave.s ;  check_mfma_hazards.py wave.s

For every kernel whose name contains `posterior_wave_kernel`, `posterior_wave2_kernel` or `tiled_gram_wave*_kernel` it walks the instruction stream (following branches
for as long as a result is pending) and reports any instruction that reads or writes a register of an asm MFMA's
destination less than 19 wait states after that MFMA, other than an MFMA accumulating into exactly that tile.
Exit code 1 on a finding.  Run by tests/test_cabi_symbols.py::test_wave_kernel_asm_hazards (CPU, cross-compile).
"""
import re
import sys

WAIT = 19
REG = re.compile(r"\b([av])(\d+)\b|\b([av])\[(\d+):(\d+)\]")


def aregs(text):
    """Vector registers named in `text`, as (file, index) pairs: accumulators may live in AGPRs or in VGPRs."""
    out = set()
    for m in REG.finditer(text):
        if m.group(1) is not None:
            out.add((m.group(1), int(m.group(2))))
        else:
            out.update((m.group(3), i) for i in range(int(m.group(4)), int(m.group(5)) + 1))
    return out


def parse(lines):
    """-> list of (kind, payload): ('label', name) | ('ins', text, in_asm)"""
    prog, in_asm = [], False
    for ln in lines:
        t = ln.strip()
        if not t:
            continue
        if t.startswith(";;#ASMSTART"):
            in_asm = True
            continue
        if t.startswith(";;#ASMEND"):
            in_asm = False
            continue
        if t.startswith(";") or t.startswith("."):
            m = re.match(r"^(\.LBB\d+_\d+):", t)
            if m:
                prog.append(("label", m.group(1)))
            continue
        t = t.split(";")[0].strip()
        if t:
            prog.append(("ins", t, in_asm))
    return prog


def step(p, pending, findings, name):
    """Apply one instruction to the pending map; returns (is_asm_mfma, branch_target or None, ends)."""
    text, in_asm = p[1], p[2]
    op = text.split()[0]
    if op.startswith("v_mfma"):
        ops = [o.strip() for o in text[len(op):].split(",")]
        dst = aregs(ops[0]) if ops else set()
        srcc = aregs(ops[3].split()[0]) if len(ops) > 3 else set()
        bad = {r for r in aregs(",".join(ops[1:3])) if pending.get(r, 0) > 0}
        part = {r for r in (dst | srcc) if pending.get(r, 0) > 0}
        if part and dst != srcc:            # accumulating into exactly the pending tile is the one legal early use
            bad |= part
        if bad:
            findings.append((name, text, sorted(bad)[:4]))
        for r in list(pending):
            pending[r] -= 1
        if in_asm:
            for r in dst:
                pending[r] = WAIT
        return in_asm, None, False
    bad = {r for r in aregs(text) if pending.get(r, 0) > 0}
    if bad:
        findings.append((name, text, sorted(bad)[:4]))
    ws = int(text.split()[1]) + 1 if op == "s_nop" else 1
    for r in list(pending):
        pending[r] -= ws
    tgt = text.split()[-1] if (op.startswith("s_cbranch") or op == "s_branch") else None
    return False, tgt, op == "s_endpgm"


def check_kernel(name, lines):
    prog = parse(lines)
    labels = {p[1]: i for i, p in enumerate(prog) if p[0] == "label"}
    findings, n_asm = [], 0

    def follow(j, sub, depth):
        """Walk from instruction index j for as long as a result is pending (at most WAIT wait states, so this
        terminates), taking both sides of every branch."""
        while sub and j < len(prog):
            q = prog[j]
            j += 1
            if q[0] == "label":
                continue
            _, tgt, ends = step(q, sub, findings, name)
            sub = {r: v for r, v in sub.items() if v > 0}
            if ends:
                return
            if tgt in labels and sub and depth < 8:
                follow(labels[tgt], dict(sub), depth + 1)
            if q[1].split()[0] == "s_branch":
                return

    pending = {}
    for i, p in enumerate(prog):            # layout order; results are carried across fall-through boundaries
        if p[0] == "label":
            continue
        is_asm, tgt, ends = step(p, pending, findings, name)
        n_asm += is_asm
        pending = {r: v for r, v in pending.items() if v > 0}
        if tgt in labels and pending:        # and along every branch, for as long as a result is pending
            follow(labels[tgt], dict(pending), 0)
        if ends or p[1].split()[0] == "s_branch":
            pending = {}                     # no fall-through behind an unconditional branch
    seen, uniq = set(), []
    for f in findings:
        key = (f[1], tuple(f[2]))
        if key not in seen:
            seen.add(key)
            uniq.append(f)
    return uniq, n_asm


def main(path):
    text = open(path).read().split("\n")
    kernels, cur = {}, None
    for ln in text:
        m = re.match(r"^(_Z\w*(?:posterior_wave_kernel|posterior_wave2_kernel|tiled_gram_wave_kernel|tiled_gram_wave_rank1_kernel|tiled_gram_wave_hfs_kernel|tiled_hf_block_gram_kernel|tiled_gram_wave_pair_kernel|tiled_diag_wave_kernel)\w*):", ln)
        if m:
            cur = kernels.setdefault(m.group(1), [])
            continue
        if cur is not None:
            if ln.startswith(".Lfunc_end"):
                cur = None
            else:
                cur.append(ln)
    if not kernels:
        print("no kernel with inline-assembly MFMAs in", path)
        return 2
    rc, total = 0, 0
    for name, lines in kernels.items():
        findings, n = check_kernel(name, lines)
        total += n
        print(f"{name}: {n} inline-asm MFMAs, {len(findings)} hazard finding(s)")
        for f in findings[:10]:
            print("   ", f[1], " touches pending registers", f[2])
        if findings:
            rc = 1
    if total == 0:            # the listing holds none of the instructions this check is about: wrong file, or a broken parse
        print("no inline-assembly MFMA found in", path)
        rc = 1
    return rc


SELFTEST = """_Z21posterior_wave_kernelILi7EEvv:
\t;;#ASMSTART
\ts_nop 1
\tv_mfma_f64_16x16x4_f64 a[0:7], v[46:47], v[50:51], a[0:7]
\t;;#ASMEND
\t;;#ASMSTART
\ts_nop 1
\tv_mfma_f64_16x16x4_f64 a[0:7], v[46:47], v[50:51], a[0:7]
\t;;#ASMEND
\ts_nop 15
\ts_cbranch_scc1 .LBB0_2
\tv_accvgpr_read_b32 v3, a9
\ts_nop 3
\tv_accvgpr_read_b32 v3, a1
\ts_endpgm
.LBB0_2:
\tv_accvgpr_read_b32 v2, a3
\ts_endpgm
.Lfunc_end0:
"""


def selftest():
    """The accumulating MFMA and the read of a9 are legal; a3 (17 wait states, along the branch) is a finding, a1
    (22 wait states) is not."""
    lines = SELFTEST.split("\n")[1:]
    findings, n = check_kernel("selftest", lines)
    return n == 2 and [f[2] for f in findings] == [[("a", 3)]]


if __name__ == "__main__":
    if sys.argv[1] == "--selftest":
        sys.exit(0 if selftest() else 1)
    sys.exit(main(sys.argv[1]))
