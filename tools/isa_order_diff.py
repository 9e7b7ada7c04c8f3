#!/usr/bin/env python3
"""Which memory-order-relevant instructions does one build of a kernel issue in another order than a second build?

    hipcc -O3 -std=c++17 --offload-arch=gfx950 [-mllvm -amdgpu-sched-strategy=...] -S --cuda-device-only -o a.s  kernel.hip
    python tools/isa_order_diff.py a.s b.s segmentx_kernelILb0 [--show N]

The kernel is cut into REGIONS at everything a machine scheduler may not move an instruction across: labels, branches, s_barrier,
s_waitcnt written by the source (kept by every strategy), inline-asm blocks, s_setprio / s_sleep.  Per region the memory-class
instructions (ds_*, global_* / buffer_*, MFMA counts as one class for context) are listed with registers normalised away; regions
of the two builds are aligned by their position in the sequence of boundary instructions.  Reported: regions whose MULTISET of memory
instructions differs (an instruction crossed a boundary -- that is an ordering violation or a different code shape), and, inside
matching regions, how many (LDS write, LDS read) and (vector-memory, LDS) pairs changed relative order (legal for a scheduler where it
proves the addresses distinct; the list is where to look).  Round 6 used it on the bf16x3 kernel under hipcc's default strategy,
max-ilp and iterative-ilp (HISTORY.md section 14.2)."""
import collections
import re
import sys


def body(path, name):
    text = open(path).read().split("\n")
    start = next(i for i, l in enumerate(text) if re.match(r"^_Z\w*%s\w*:" % name, l))
    end = next(i for i in range(start, len(text)) if "s_endpgm" in text[i])
    return text[start:end + 1]


def norm(t):
    t = re.sub(r";.*$", "", t).strip()
    t = re.sub(r"\b[vsa]\[\d+:\d+\]", "R", t)
    t = re.sub(r"\b[vsa]\d+\b", "R", t)
    t = re.sub(r"\bvcc\b|\bexec\b|\bm0\b", "R", t)
    return re.sub(r"\s+", " ", t)


def regions(lines):
    """[(boundary text, [normalised memory-class instructions in order])]"""
    out = [("entry", [])]
    in_asm = False
    for raw in lines:
        t = raw.strip()
        if t.startswith(";APP"):
            in_asm = True
            out.append(("asm", []))
            continue
        if t.startswith(";NO_APP"):
            in_asm = False
            out.append(("asm-end", []))
            continue
        t0 = re.sub(r";.*$", "", t).strip()
        if not t0 or t0.startswith("."):
            if t0.endswith(":"):
                out.append(("label", []))
            continue
        op = t0.split()[0]
        if t0.endswith(":"):
            out.append(("label", []))
        elif op.startswith(("s_cbranch", "s_branch")):
            out.append(("branch", []))
        elif op in ("s_barrier", "s_sleep", "s_setprio", "s_endpgm"):
            out.append((op, []))
        elif in_asm:
            out[-1][1].append("ASM " + norm(t0))
        elif op.startswith(("ds_", "global_", "buffer_", "flat_", "scratch_")):
            out[-1][1].append(norm(t0))
        elif op.startswith("v_mfma"):
            out[-1][1].append("MFMA")
    return out


def cls(i):
    if i.startswith("ds_read") or i.startswith("ds_load"):
        return "lds_r"
    if i.startswith("ds_"):
        return "lds_w"
    if i.startswith(("global_load", "buffer_load", "flat_load")):
        return "vm_r"
    if i.startswith(("global_", "buffer_", "flat_")):
        return "vm_w"
    return "other"


def main():
    a, b, name = sys.argv[1], sys.argv[2], sys.argv[3]
    show = int(sys.argv[sys.argv.index("--show") + 1]) if "--show" in sys.argv else 12
    ra, rb = regions(body(a, name)), regions(body(b, name))
    # regions hold code only between boundaries; drop empty ones and align by order among the non-empty
    na = [(k, v) for k, v in ra if [i for i in v if i != "MFMA"]]
    nb = [(k, v) for k, v in rb if [i for i in v if i != "MFMA"]]
    print("%s: %d regions with memory instructions in %s, %d in %s" % (name, len(na), a, len(nb), b))
    # align greedily on the multiset signature
    sig = lambda v: tuple(sorted(collections.Counter(i for i in v if i != "MFMA").items()))
    ia = ib = 0
    crossed = swapped = 0
    while ia < len(na) and ib < len(nb):
        va, vb = na[ia][1], nb[ib][1]
        if sig(va) == sig(vb):
            ma = [i for i in va if i != "MFMA"]
            mb = [i for i in vb if i != "MFMA"]
            if ma != mb:
                # pairs (x, y) of different classes whose relative order differs
                inv = collections.Counter()
                pa = {}
                for n, i in enumerate(ma):
                    pa.setdefault(i, []).append(n)
                # index instances
                def inst(seq):
                    c = collections.Counter()
                    o = []
                    for i in seq:
                        o.append((i, c[i]))
                        c[i] += 1
                    return o
                oa, ob = inst(ma), inst(mb)
                posb = {k: n for n, k in enumerate(ob)}
                for x in range(len(oa)):
                    for y in range(x + 1, len(oa)):
                        cx, cy = cls(oa[x][0]), cls(oa[y][0])
                        if cx != cy and posb[oa[x]] > posb[oa[y]]:
                            inv[(cx, cy)] += 1
                if inv:
                    swapped += 1
                    if swapped <= show:
                        print("  region %d/%d (after %s): same instructions, %s" % (ia, ib, na[ia][0], ", ".join("%s before %s -> after: %d pairs" % (k[0], k[1], v) for k, v in inv.items())))
            ia += 1
            ib += 1
        else:
            # look ahead for a re-sync
            found = False
            for d in range(1, 6):
                if ia + d < len(na) and sig(na[ia + d][1]) == sig(vb):
                    for j in range(d):
                        crossed += 1
                        print("  region %d of A has no counterpart in B: %s" % (ia + j, collections.Counter(cls(i) for i in na[ia + j][1] if i != "MFMA")))
                    ia += d
                    found = True
                    break
                if ib + d < len(nb) and sig(nb[ib + d][1]) == sig(va):
                    for j in range(d):
                        crossed += 1
                        print("  region %d of B has no counterpart in A: %s" % (ib + j, collections.Counter(cls(i) for i in nb[ib + j][1] if i != "MFMA")))
                    ib += d
                    found = True
                    break
            if not found:
                crossed += 1
                da = collections.Counter(i for i in va if i != "MFMA")
                db = collections.Counter(i for i in vb if i != "MFMA")
                print("  regions %d / %d differ in content: only in A %s | only in B %s" % (ia, ib, dict(da - db), dict(db - da)))
                ia += 1
                ib += 1
    print("regions whose content differs: %d; regions with the same instructions in another cross-class order: %d" % (crossed, swapped))


if __name__ == "__main__":
    main()
