#!/usr/bin/env python3
"""Static check of a gfx950 assembly listing for the wait states the hardware leaves to software, over every control-flow path:

    an MFMA writes VGPRs  ->  a VALU / LDS / vector-memory instruction reads or overwrites them too few wait states later
    (check_kernel; the failure this tool came from), and the classic gfx9 rules (check_kernel_classic): a VALU-written SGPR read by
    vector memory (5) or taken as a v_readlane / v_writelane lane select (4), a DPP instruction behind a VALU write of its source (2) or
    of EXEC (5), a VALU write of the data registers of a store of more than 64 bits (2).

gfx940 / gfx950 have no hardware interlock for this: the compiler must place `s_nop`s (CDNA3 ISA guide section 4.5; LLVM
GCNHazardRecognizer::checkMAIVALUHazards).  ROCm 7.2's recognizer walks the predecessors of a block with ONE `Visited` set for all
paths: at the join of an if-without-else (P2 -> [P1 ->] B) it reaches P2 first THROUGH the short block P1, marks it visited, and never
evaluates the direct edge P2 -> B -- an MFMA at the end of P2 whose result the first instruction of B reads gets no `s_nop` when the
branch is taken.  Which registers the join reads first is the pre-RA scheduler's choice: hipcc's default strategy and `max-ilp` happen to
finish those tiles early in `x3::segmentx_kernel`; `iterative-ilp` finishes them LAST, right in front of the `skip_last` branch, and the
kernel then reads stale accumulators now and then (the failure HISTORY.md section 13.9 recorded; diagnosis in section 14.2).

    hipcc -O3 -std=c++17 --offload-arch=gfx950 [flags] -S --cuda-device-only -o k.s kernel.hip   |   llvm-objdump -d file.o > k.s
    python tools/isa_hazard_check.py k.s [k2.s ...]          exit status 1 if any violation is found

Wait states per MFMA opcode = what hipcc itself places in straight-line code (measured: tests/test_isa_hazards.py::CALIBRATION);
`s_nop N` counts N + 1, every other instruction 1.  MFMA consumers are not checked (dependent MFMAs have their own, interlocked, rules)."""
import re
import sys

# wait states hipcc guarantees between the MFMA and a VALU / memory instruction touching its destination
NEED = {"v_mfma_f32_16x16x32_bf16": 8, "v_mfma_f32_32x32x16_bf16": 12, "v_mfma_f32_16x16x4_f32": 10, "v_mfma_f32_16x16x4f32": 10}
DEFAULT_NEED = 18          # any other MFMA: the largest figure of the family (16 passes + 2)

REG = re.compile(r"\b([va])(?:\[(\d+):(\d+)\]|(\d+)\b)")


def regs(text):
    out = set()
    for m in REG.finditer(text):
        if m.group(4) is not None:
            out.add((m.group(1), int(m.group(4))))
        else:
            out.update((m.group(1), r) for r in range(int(m.group(2)), int(m.group(3)) + 1))
    return out


SREG = re.compile(r"\bs(?:\[(\d+):(\d+)\]|(\d+)\b)")


def sregs(text):
    """SGPRs named in ``text`` (s5, s[4:7], vcc = ("vcc", 0/1), exec, m0)."""
    out = set()
    for m in SREG.finditer(text):
        if m.group(3) is not None:
            out.add(("s", int(m.group(3))))
        else:
            out.update(("s", r) for r in range(int(m.group(1)), int(m.group(2)) + 1))
    for name in ("vcc", "exec", "m0"):
        if re.search(r"\b%s(_lo|_hi)?\b" % name, text):
            out.add((name, 0))
    return out


def operands(t):
    op = t.split()[0]
    rest = t[len(op):]
    return op, [x.strip() for x in rest.split(",")] if rest.strip() else []


def is_valu(op):
    return op.startswith("v_") and not op.startswith(("v_mfma", "v_smfmac"))


def is_vmem(op):
    return op.startswith(("global_", "buffer_", "flat_", "scratch_"))


def valu_sgpr_writes(t):
    """SGPRs (and vcc / exec) a VALU instruction writes."""
    op, ops = operands(t)
    if not is_valu(op):
        return set()
    out = set()
    if op.startswith(("v_readlane", "v_readfirstlane")) and ops:
        out |= sregs(ops[0])
    elif op.startswith("v_cmpx"):
        out.add(("exec", 0))
        if op.endswith("_e64") and ops:
            out |= sregs(ops[0])
    elif op.startswith("v_cmp"):
        if ops and (ops[0].startswith("s") or ops[0].startswith("vcc")) and (op.endswith("_e64") or len(ops) == 3):
            out |= sregs(ops[0])
        else:
            out.add(("vcc", 0))
    elif re.match(r"v_(add|sub|subrev|addc|subb|subbrev)_co_", op) or op.startswith(("v_div_scale", "v_mad_u64_u32", "v_mad_i64_i32")):
        if len(ops) > 1:
            out |= sregs(ops[1])                    # the carry / flag destination is the second operand
    return out


def parse(path):
    """{kernel: [(kind, text)]}; kind in {"label", "ins"}.  Accepts hipcc -S output (labels) and `llvm-objdump -d` output of a device
    code object (no labels: an instruction's address and a branch's target `<kernel+0xOFF>` are in the trailing comment; every
    instruction then gets a label named after its address and branches are rewritten to those names)."""
    kernels, cur, name, base = {}, None, None, None
    for raw in open(path, errors="replace"):
        line = raw.rstrip("\n")
        m = re.match(r"^(?:([0-9a-f]+) <)?(_Z[\w$.]+)>?:\s*(?:;.*)?$", line)
        if m:
            name, cur = m.group(2), []
            base = int(m.group(1), 16) if m.group(1) else None
            kernels[name] = cur
            continue
        if cur is None:
            continue
        t = re.sub(r"(;|//).*$", "", line).strip()
        if not t:
            continue
        lm = re.match(r"^(?:[0-9a-f]+ <)?([.\w$]+)>?:$", t)
        if lm:
            cur.append(("label", lm.group(1)))
            continue
        if t.startswith("."):
            continue
        if base is not None:                                   # llvm-objdump
            am = re.search(r"//\s*([0-9A-Fa-f]+):", line)
            if am:
                cur.append(("label", "@%x" % int(am.group(1), 16)))
            if t.startswith(("s_cbranch", "s_branch")):
                bm = re.search(r"<%s(?:\+0x([0-9a-fA-F]+))?>\s*$" % re.escape(name), line)
                if bm:
                    t = "%s @%x" % (t.split()[0], base + int(bm.group(1) or "0", 16))
        cur.append(("ins", t))
        if t.startswith("s_endpgm"):
            cur = None
    return {k: v for k, v in kernels.items() if any(kind == "ins" and "s_endpgm" in t for kind, t in v)}


def check_kernel(name, items):
    ins = []                                   # (text, label-before or None)
    label_at = {}
    for kind, t in items:
        if kind == "label":
            label_at[t] = len(ins)
        else:
            ins.append(t)
    n = len(ins)
    preds = [[] for _ in range(n)]
    for i, t in enumerate(ins):
        op = t.split()[0]
        tgt = None
        if op.startswith(("s_cbranch", "s_branch")):
            lab = t.split()[-1]
            lab = re.sub(r"^<|>$", "", lab)
            tgt = label_at.get(lab)
            if tgt is None:
                m = re.search(r"<([.\w$]+)", t)
                tgt = label_at.get(m.group(1)) if m else None
        if tgt is not None and tgt < n:
            preds[tgt].append(i)
        if op != "s_branch" and not op.startswith("s_endpgm") and i + 1 < n:
            preds[i + 1].append(i)
    ws = []
    for t in ins:
        op = t.split()[0]
        ws.append(int(t.split()[1], 0) + 1 if op == "s_nop" else 1)
    mfma = {}
    for i, t in enumerate(ins):
        op = t.split()[0]
        if op.startswith("v_mfma") or op.startswith("v_smfmac"):
            dst = t[len(op):].split(",")[0]
            mfma[i] = (regs(dst), NEED.get(op.replace("_e64", ""), DEFAULT_NEED), op)
    max_need = max([v[1] for v in mfma.values()], default=0)
    bad = []
    if not mfma:
        return bad
    for i, t in enumerate(ins):
        op = t.split()[0]
        if i in mfma or not op.startswith(("v_", "ds_", "global_", "buffer_", "flat_", "scratch_")):
            continue
        touched = regs(t)
        if not touched:
            continue
        # backward search over all paths: (instruction index, wait states accumulated so far); a state is revisited only with FEWER
        # accumulated wait states (the recognizer's bug is exactly not doing that)
        best = {}
        stack = [(p, 0) for p in preds[i]]
        while stack:
            j, acc = stack.pop()
            if acc >= max_need or best.get(j, 1 << 30) <= acc:
                continue
            best[j] = acc
            if j in mfma:
                dregs, need, mop = mfma[j]
                if acc < need and dregs & touched:
                    bad.append((name, j, ins[j], i, t, acc, need))
                    continue
            for p in preds[j]:
                stack.append((p, acc + ws[j]))
    return bad


# ---- the classic gfx9 wait-state rules (CDNA3 ISA guide section 4.5; LLVM checkVMEMHazards / checkRWLaneHazards / checkDPPHazards /
# checkVALUHazardsHelper), searched over every path like the MFMA rule: the recognizer's hole (one visited-set for all paths) is not
# specific to MFMAs.  (name, wait states needed)
RULES = (("VALU writes an SGPR, a vector-memory instruction reads it", 5),
         ("VALU writes an SGPR / VCC, v_readlane / v_writelane takes it as the lane select", 4),
         ("VALU writes a VGPR, a DPP instruction reads it", 2),
         ("VALU writes EXEC, a DPP instruction follows", 5),
         ("a vector-memory store of more than 64 bits, a VALU instruction overwrites its data registers", 2))


def check_kernel_classic(name, items):
    ins = []
    label_at = {}
    for kind, t in items:
        if kind == "label":
            label_at[t] = len(ins)
        else:
            ins.append(t)
    n = len(ins)
    preds = [[] for _ in range(n)]
    for i, t in enumerate(ins):
        op = t.split()[0]
        tgt = None
        if op.startswith(("s_cbranch", "s_branch")):
            tgt = label_at.get(re.sub(r"^<|>$", "", t.split()[-1]))
        if tgt is not None and tgt < n:
            preds[tgt].append(i)
        if op != "s_branch" and not op.startswith("s_endpgm") and i + 1 < n:
            preds[i + 1].append(i)
    ws = [int(t.split()[1], 0) + 1 if t.split()[0] == "s_nop" else 1 for t in ins]
    swr = [valu_sgpr_writes(t) for t in ins]
    bad = []

    def search(i, need, hit):
        best = {}
        stack = [(p, 0) for p in preds[i]]
        while stack:
            j, acc = stack.pop()
            if acc >= need or best.get(j, 1 << 30) <= acc:
                continue
            best[j] = acc
            if hit(j):
                bad_pair.append((j, acc))
                continue
            for p in preds[j]:
                stack.append((p, acc + ws[j]))

    for i, t in enumerate(ins):
        op, ops = operands(t)
        dpp = " dpp" in t or "quad_perm" in t or "row_shr" in t or "row_shl" in t or "row_ror" in t or "row_bcast" in t or "row_mirror" in t or "row_half_mirror" in t or "wave_shr" in t or "wave_ror" in t or "row_newbcast" in t
        checks = []
        if is_vmem(op):
            used = sregs(t) - {("exec", 0), ("m0", 0)}
            if used:
                checks.append((0, lambda j, used=used: bool(swr[j] & used)))
        if op.startswith(("v_readlane", "v_writelane")) and len(ops) >= 3:
            sel = sregs(ops[2])
            if sel:
                checks.append((1, lambda j, sel=sel: bool(swr[j] & sel)))
        if dpp and is_valu(op):
            src = regs(",".join(ops[1:]))
            checks.append((2, lambda j, src=src: is_valu(ins[j].split()[0]) and bool(regs(operands(ins[j])[1][0] if operands(ins[j])[1] else "") & src)))
            checks.append((3, lambda j: ("exec", 0) in swr[j]))
        if is_valu(op) and ops:
            dst = regs(ops[0])
            if dst:
                def store_hit(j, dst=dst):
                    o2, p2 = operands(ins[j])
                    if not (is_vmem(o2) and "store" in o2 and ("x3" in o2 or "x4" in o2)):
                        return False
                    data = set()
                    for x in p2:
                        r = regs(x)
                        if len(r) >= 3:
                            data |= r
                    return bool(data & dst)
                checks.append((4, store_hit))
        for rule, hit in checks:
            bad_pair = []
            search(i, RULES[rule][1], hit)
            for j, acc in bad_pair:
                bad.append((name, j, ins[j], i, t, acc, RULES[rule][1], RULES[rule][0]))
    return bad


def main(paths):
    total = 0
    for path in paths:
        ks = parse(path)
        n_mfma = 0
        for name, items in ks.items():
            n_mfma += sum(1 for kind, t in items if kind == "ins" and t.startswith("v_mfma"))
            seen = set()
            for b in check_kernel(name, items):
                key = (b[1], b[3])
                if key in seen:
                    continue
                seen.add(key)
                total += 1
                print("%s: %s\n    [%d] %s\n    [%d] %s\n    %d wait state(s) on some path, %d needed" % (path, b[0], b[1], b[2], b[3], b[4], b[5], b[6]))
            seen = set()
            for b in check_kernel_classic(name, items):
                key = (b[1], b[3], b[7])
                if key in seen:
                    continue
                seen.add(key)
                total += 1
                print("%s: %s  (%s)\n    [%d] %s\n    [%d] %s\n    %d wait state(s) on some path, %d needed" % (path, b[0], b[7], b[1], b[2], b[3], b[4], b[5], b[6]))
        print("%s: %d kernels, %d MFMAs checked" % (path, len(ks), n_mfma))
    print("violations: %d" % total)
    return 1 if total else 0


if __name__ == "__main__":
    sys.exit(main(sys.argv[1:]))
