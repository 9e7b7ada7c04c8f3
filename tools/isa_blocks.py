#!/usr/bin/env python3
"""Static instruction mix of one kernel, basic block by basic block, from hipcc's assembly:

    hipcc -O3 -std=c++17 --offload-arch=gfx950 -S --cuda-device-only -o /tmp/k.s dl4vc_amd/csrc/dan_kernels_bf16x.hip
    python tools/isa_blocks.py /tmp/k.s segmentx_kernel [min_instructions]

Prints, per block (between labels / branches / barriers): MFMAs, other vector instructions (and how many of them are
v_readlane / v_writelane, i.e. scalar registers parked in vector lanes), LDS, scalar and vector-memory instructions.  Round 5 used it
to find that the bf16x3 kernel formed the NEXT row's request addresses at the head of every layer (HISTORY.md section 13)."""
import re
import sys

path, name = sys.argv[1], sys.argv[2]
floor = int(sys.argv[3]) if len(sys.argv) > 3 else 0
text = open(path).read().split("\n")
start = next(i for i, l in enumerate(text) if re.match(r"^_Z\w*%s\w*:" % name, l))
end = next(i for i in range(start, len(text)) if "s_endpgm" in text[i])
cur = dict(mfma=0, valu=0, lane=0, ds=0, salu=0, vmem=0)
tot = dict(cur)
begin = start


def flush(tag, i):
    global cur, begin
    if sum(cur.values()) - cur["lane"] >= floor:
        print("%6d-%6d %-26s mfma %4d  valu %4d (lane %3d)  ds %3d  salu %4d  vmem %3d" % (begin - start, i - start, tag[:26], cur["mfma"], cur["valu"], cur["lane"], cur["ds"], cur["salu"], cur["vmem"]))
    for k in cur:
        tot[k] += cur[k]
    cur = {k: 0 for k in cur}
    begin = i


for i in range(start, end + 1):
    t = text[i].strip()
    m = re.match(r"^([a-z_0-9]+)", t)
    if t.endswith(":") and not t.startswith(";"):
        flush("label " + t, i)
        continue
    if not m:
        continue
    op = m.group(1)
    if op == "s_barrier":
        flush("BARRIER", i)
    elif op.startswith("s_cbranch") or op == "s_branch":
        cur["salu"] += 1
        flush(op + " " + t.split()[-1], i)
    elif op.startswith("v_mfma"):
        cur["mfma"] += 1
    elif op.startswith("v_"):
        cur["valu"] += 1
        cur["lane"] += "lane" in op
    elif op.startswith("ds_"):
        cur["ds"] += 1
    elif op.startswith("s_"):
        cur["salu"] += 1
    elif op.startswith(("global_", "buffer_", "flat_")):
        cur["vmem"] += 1
flush("end", end)
print("total: " + "  ".join("%s %d" % kv for kv in tot.items()))
