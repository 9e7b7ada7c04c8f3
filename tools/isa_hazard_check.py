#!/usr/bin/env python3
"""Static check of a gfx950 assembly listing for ONE software-managed hazard, over every control-flow path:

    an MFMA writes VGPRs  ->  a VALU / LDS / vector-memory instruction reads or overwrites them too few wait states later.

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
        print("%s: %d kernels, %d MFMAs checked" % (path, len(ks), n_mfma))
    print("violations: %d" % total)
    return 1 if total else 0


if __name__ == "__main__":
    sys.exit(main(sys.argv[1:]))
