#!/usr/bin/env python3
"""Condenses a rocprofv3 --kernel-trace --memory-copy-trace run of bench.py (tools/host_path_trace.sh) into the facts that explain the
host path's rate: the last pass of the run (the host-path pass: the one with host-to-device copies between kernels), its copies (bytes,
duration, GB/s, direction), the kernels' busy time, the idle gaps of the compute queue longer than 0.2 ms and what preceded them."""
import csv
import glob
import os
import sys

d = sys.argv[1]
kt = sorted(glob.glob(os.path.join(d, "**", "*_kernel_trace.csv"), recursive=True), key=os.path.getmtime)[-1]
mc = sorted(glob.glob(os.path.join(d, "**", "*_memory_copy_trace.csv"), recursive=True), key=os.path.getmtime)[-1]
ks = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"][:40]) for r in csv.DictReader(open(kt))]
cs = []
for r in csv.DictReader(open(mc)):
    a, b = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    # (this rocprofv3 writes no size column: a copy's DURATION stands in for it -- "large" = longer than 0.3 ms)
    cs.append((a, b, r.get("Direction", r.get("Kind", "?")), int(r.get("Size", r.get("Bytes", 0)) or 0) or (b - a) * 50))
ks.sort(); cs.sort()
big = [c for c in cs if c[1] - c[0] >= 300_000 and "HOST_TO_DEVICE" in c[2].upper().replace(" ", "_")]
if not big:
    print("no large host-to-device copies found; directions seen:", sorted({c[2] for c in cs})); sys.exit(0)
# the host-path pass = from the first large H2D copy that follows the timed passes' kernels to the end
t_first = big[0][0]
# large H2D copies during the initial upload of the resident batch come first: take the LAST contiguous group of copies (gap < 1 s)
groups = [[big[0]]]
for c in big[1:]:
    if c[0] - groups[-1][-1][1] > 1_000_000_000:
        groups.append([])
    groups[-1].append(c)
g = groups[-1]
t0, t1 = g[0][0], max(k[1] for k in ks)
kk = [k for k in ks if k[0] >= t0]
print("host-path pass: %.1f ms from the first copy to the last kernel end; %d large H2D copies, %d kernels" % ((t1 - t0) / 1e6, len(g), len(kk)))
tot = sum(c[3] for c in g); dur = sum(c[1] - c[0] for c in g)
print("H2D: %d large copies, summed duration %.1f ms (longest %.2f ms, median %.2f ms); first copy ends at +%.2f ms" % (
    len(g), dur / 1e6, max(c[1] - c[0] for c in g) / 1e6, sorted(c[1] - c[0] for c in g)[len(g) // 2] / 1e6, (g[0][1] - t0) / 1e6))
print("first kernel starts at +%.2f ms" % ((kk[0][0] - t0) / 1e6))
busy = sum(k[1] - k[0] for k in kk)
print("kernels busy %.1f ms of %.1f ms (%.3f)" % (busy / 1e6, (t1 - t0) / 1e6, busy / (t1 - t0)))
end = kk[0][1]
gaps = []
for k in kk[1:]:
    if k[0] - end > 200_000:
        gaps.append((end - t0, k[0] - end, k[2]))
    end = max(end, k[1])
print("%d idle gaps > 0.2 ms, %.1f ms in total:" % (len(gaps), sum(g_[1] for g_ in gaps) / 1e6))
for at, ln, name in gaps[:24]:
    copying = [c for c in g if c[0] < t0 + at + ln and c[1] > t0 + at]
    print("   at +%8.2f ms: %6.2f ms idle before %-40s (%d copies in flight then)" % (at / 1e6, ln / 1e6, name, len(copying)))
