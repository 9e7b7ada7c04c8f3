#!/usr/bin/env python3
"""Condenses the rocprofv3 outputs of tools/profile_round.sh into the small files kept under profiles/.

    python tools/summarize_profile.py r01   # reads gpurun_out/prof_r01/, writes profiles/r01_*

  * <tag>_kernel_stats_default_bench.csv   rocprofv3 --kernel-trace --stats summary (verbatim) of the default bench command,
                                          plus the segment kernel split by template instantiation / segment
  * <tag>_bench_line_under_rocprof.json    the JSON line bench.py printed in that same run
  * <tag>_traffic.json                     FETCH_SIZE / WRITE_SIZE per kernel (KB as reported), corrected bytes per launch
  * <tag>_pmc_sq_summary.csv               SQ counters summed over dispatches per kernel
"""
import csv
import glob
import json
import os
import re
import sys
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def one(pattern):
    hits = sorted(glob.glob(pattern, recursive=True), key=os.path.getmtime)      # gpurun merges runs: newest wins
    if not hits:
        raise SystemExit("no file matches " + pattern)
    return hits[-1]


def short(name):
    m = re.match(r"(?:void )?([A-Za-z_:0-9]+(?:<[^>]*>)?)", name)
    return m.group(1) if m else name


def counters(path):
    """{kernel: {counter: [values per dispatch]}} -- a dispatch's rows per counter are summed (one row per XCD/SE)."""
    per = defaultdict(lambda: defaultdict(lambda: defaultdict(float)))
    with open(path) as f:
        for row in csv.DictReader(f):
            per[short(row["Kernel_Name"])][row["Counter_Name"]][row["Dispatch_Id"]] += float(row["Counter_Value"])
    return {k: {c: list(d.values()) for c, d in cs.items()} for k, cs in per.items()}


def build_of(json_path):
    """The ``build`` object ({source_hash, chunk_sites, ...}) of the bench line a profiled run printed; {} when absent."""
    if not os.path.isfile(json_path):
        return {}
    ls = [l for l in open(json_path) if l.startswith("{")]
    return json.loads(ls[-1]).get("build", {}) if ls else {}


def tree_identity():
    """Where the summary was made: git HEAD (+ "-dirty" when tracked files differ) and the source hash of the tree as it stands."""
    import subprocess
    sys.path.insert(0, ROOT)
    from dl4vc_amd import capi
    try:
        head = subprocess.check_output(["git", "-C", ROOT, "rev-parse", "HEAD"], text=True).strip()
        if subprocess.run(["git", "-C", ROOT, "diff", "--quiet", "HEAD", "--", "dl4vc_amd/csrc"]).returncode:
            head += "-dirty"
    except Exception:
        head = "unknown"
    return {"git_head_at_summary": head, "tree_source_hash_at_summary": capi.tree_source_hash()}


def main():
    tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
    src = os.path.join(ROOT, "gpurun_out", "prof_" + tag)
    dst = os.path.join(ROOT, "profiles")
    os.makedirs(dst, exist_ok=True)

    # ---- kernel trace
    stats = one(os.path.join(src, "kt", "**", "*_kernel_stats.csv"))
    trace = one(os.path.join(src, "kt", "**", "*_kernel_trace.csv"))
    seg = defaultdict(list)
    with open(trace) as f:
        for row in csv.DictReader(f):
            if "segment_kernel" in row["Kernel_Name"]:
                seg[short(row["Kernel_Name"])].append(int(row["End_Timestamp"]) - int(row["Start_Timestamp"]))
    with open(os.path.join(dst, tag + "_kernel_stats_default_bench.csv"), "w") as out:
        out.write("# rocprofv3 --kernel-trace --stats --output-format csv -- python3 bench.py --no-cpu-baseline   (kernel_stats.csv, verbatim)\n")
        out.write(open(stats).read())
        out.write("# segment kernel launches alternate between the two layer segments of a chunk (layers 1-2, layers 3-7):\n")
        out.write("# kernel,launches,avg_ns_all,avg_ns_even_launches(segment 1),avg_ns_odd_launches(segment 2)\n")
        for k, d in seg.items():
            ev, od = d[0::2], d[1::2]
            out.write("# %s,%d,%.0f,%.0f,%.0f\n" % (k, len(d), sum(d) / len(d), sum(ev) / max(len(ev), 1), sum(od) / max(len(od), 1)))
    line = [l for l in open(os.path.join(src, "bench_under_rocprof.json")) if l.startswith("{")][-1]

    # ---- HBM traffic
    traffic = {"command": "rocprofv3 --pmc FETCH_SIZE | WRITE_SIZE (separate passes) --output-format csv -- python3 bench.py "
                          "--sites 4096 --steps 1 --warmup 0 --no-cpu-baseline   (default chunk = 2048 sites at 64 x 201, 2 segment launches per chunk)",
               "units": "rocprofv3 reports FETCH_SIZE/WRITE_SIZE in KB; FETCH_SIZE is doubled (gfx950 tallies 128-B read requests at "
                        "64 B for wide coalesced streams, MI355X_MICROARCH.md section HBM); WRITE_SIZE is exact for 16-B-per-lane stores",
               "per_kernel": {}}
    seg_bytes = {}
    for c in ("FETCH_SIZE", "WRITE_SIZE"):
        per = counters(one(os.path.join(src, c, "**", "*_counter_collection.csv")))
        traffic["per_kernel"][c] = {}
        for k, cs in per.items():
            if not k.startswith("dan::"):
                continue
            v = cs[c]
            traffic["per_kernel"][c][k] = {"dispatches": len(v), "mean_KB": sum(v) / len(v), "min_KB": min(v), "max_KB": max(v)}
            if "segment_kernel" in k:
                seg_bytes[c] = sum(v) / len(v) * 1024.0
    fetch = 2.0 * seg_bytes["FETCH_SIZE"]
    traffic.update(tree_identity())
    b_f, b_w = build_of(os.path.join(src, "FETCH_SIZE.json")), build_of(os.path.join(src, "WRITE_SIZE.json"))
    if b_f.get("source_hash") != b_w.get("source_hash"):
        raise SystemExit("the FETCH_SIZE and WRITE_SIZE passes ran on different builds")
    # bench.py reports `traffic` only for a library with this source_hash whose handle chose this chunk (bench.py::pmc_traffic)
    traffic["segment_kernel_bytes_per_launch"] = {"fetch_corrected": fetch, "write": seg_bytes["WRITE_SIZE"],
                                                  "total": fetch + seg_bytes["WRITE_SIZE"],
                                                  "source_hash": b_f.get("source_hash"), "chunk_sites": b_f.get("chunk_sites")}
    n_disp = max(d["dispatches"] for k, d in traffic["per_kernel"]["WRITE_SIZE"].items() if "segment_kernel" in k)
    # the PMC bench runs 4096 sites, 2 segment launches per chunk -- once for the timed pass and once more when the line carries
    # the host_path pass (same chunking)
    pmc_line = [l for l in open(os.path.join(src, "FETCH_SIZE.json")) if l.startswith("{")]
    passes = 2 if pmc_line and "host_path" in json.loads(pmc_line[-1]) else 1
    R, L, S = 64, 201, 4096 * 2 * passes // n_disp
    y = S * R * L * 128 * 4
    algo = (S * (3 * R * L + 3 * L) + 3 * y + 7 * S * R * L * 32 * 4 + S * L * 128 * 4) / 2.0
    traffic["segment_kernel_algorithmic_bytes_per_launch"] = {
        "note": "chunk of %d sites x 64 reads x 201: uint8 inputs + y2 write + y2 read + pool read + y7 write + bottleneck outputs h "
                "(7 layers), averaged over the two launches of a chunk" % S, "total": algo}
    with open(os.path.join(dst, tag + "_traffic.json"), "w") as out:
        json.dump(traffic, out, indent=1)
    # the profiled run printed its line before these counters existed: carry the traffic measured for the same command
    rec = json.loads(line)
    if rec.get("build", {}).get("source_hash") == b_f.get("source_hash") and rec.get("build", {}).get("chunk_sites") == b_f.get("chunk_sites"):
        rec["roofline"]["traffic"] = int(fetch + seg_bytes["WRITE_SIZE"])
        rec["roofline"]["traffic_stale"] = False
    with open(os.path.join(dst, tag + "_bench_line_under_rocprof.json"), "w") as out:
        out.write(json.dumps(rec) + "\n")

    # ---- SQ counters
    rows = defaultdict(dict)
    disp = {}
    for d in ("SQ", "SQ2"):
        per = counters(one(os.path.join(src, d, "**", "*_counter_collection.csv")))
        for k, cs in per.items():
            if k.startswith("dan::"):
                for c, v in cs.items():
                    rows[k][c] = sum(v)
                    disp[k] = len(v)
    names = sorted({c for r in rows.values() for c in r})
    with open(os.path.join(dst, tag + "_pmc_sq_summary.csv"), "w") as out:
        out.write("# rocprofv3 --pmc <SQ counters, two passes> -- python3 bench.py --sites 4096 --steps 1 --warmup 0 --no-cpu-baseline ; sums over dispatches\n")
        out.write("kernel,dispatches," + ",".join(names) + "\n")
        for k, r in rows.items():
            out.write("%s,%d,%s\n" % (k, disp[k], ",".join("%.6g" % r.get(c, float("nan")) for c in names)))
        for k, r in rows.items():
            if "segment_kernel" in k and "SQ_VALU_MFMA_BUSY_CYCLES" in r and "SQ_BUSY_CU_CYCLES" in r:
                out.write("# %s: SQ_VALU_MFMA_BUSY_CYCLES / (4 x SQ_BUSY_CU_CYCLES) = %.4f\n"
                          % (k, r["SQ_VALU_MFMA_BUSY_CYCLES"] / (4.0 * r["SQ_BUSY_CU_CYCLES"])))
    print("wrote profiles/%s_*" % tag)
    extras(tag, dst)


def extras(tag, dst):
    """Round 3 on: kernel traces and PMC passes of the bf16 config-5 command and of the training step (tools/capture_round.sh b),
    the bench lines of the other configurations, and their HBM traffic appended to <tag>_traffic.json."""
    import shutil
    import subprocess
    go = os.path.join(ROOT, "gpurun_out")
    tpath = os.path.join(dst, tag + "_traffic.json")
    traffic = json.load(open(tpath))
    for name, what, cmd in (("train", "bench.py --mode train --steps 5 --warmup 2 --no-cpu-baseline   (tools/profile_train.sh)", None),
                            ("bf16", "bench.py --precision 2 --reads 128 --window 301 --sites 4096 --steps 2 --warmup 1 --no-cpu-baseline --no-skip-pass --no-host-path", None),
                            ("bf16x3", "bench.py --precision 1 --sites 16384 --steps 2 --warmup 1 --no-cpu-baseline --no-skip-pass --no-host-path", None)):
        hits = sorted(glob.glob(os.path.join(go, "prof_%s_%s" % (name, tag), "kt", "**", "*_kernel_stats.csv"), recursive=True), key=os.path.getmtime)
        if hits:
            with open(os.path.join(dst, "%s_%s_kernel_stats.csv" % (tag, name)), "w") as out:
                out.write("# rocprofv3 --kernel-trace --stats --output-format csv -- python3 %s   (kernel_stats.csv, verbatim)\n" % what)
                out.write(open(hits[-1]).read())
    pmc = {"train": ("--mode train --steps 2 --warmup 1 --no-cpu-baseline", 3),
           "train_b10": ("--mode train --train-batch 10 --steps 2 --warmup 1 --no-cpu-baseline", 3),
           "bf16": ("--precision 2 --reads 128 --window 301 --sites 2048 --steps 1 --warmup 0 --no-cpu-baseline --no-skip-pass --no-host-path", 1),
           "bf16x3": ("--precision 1 --sites 4096 --steps 1 --warmup 0 --no-cpu-baseline --no-skip-pass --no-host-path", 1)}
    for name, (command, passes) in pmc.items():
        src = os.path.join(go, "pmc_%s_%s" % (name, tag))
        if not os.path.isdir(src):
            continue
        csv_out = os.path.join(dst, "%s_%s_pmc_summary.csv" % (tag, name))
        subprocess.check_call([sys.executable, os.path.join(ROOT, "tools", "summarize_pmc.py"), src, csv_out, "--command", command])
        rows = list(csv.DictReader(l for l in open(csv_out) if not l.startswith("#")))
        b = build_of(os.path.join(src, "FETCH_SIZE.json"))
        if b.get("source_hash") != build_of(os.path.join(src, "WRITE_SIZE.json")).get("source_hash"):
            raise SystemExit("the FETCH_SIZE and WRITE_SIZE passes of %s ran on different builds" % name)
        ident = {"source_hash": b.get("source_hash"), "chunk_sites": b.get("chunk_sites")}
        if name.startswith("train"):
            sites = 10 if name == "train_b10" else 64
            total = sum(float(r["hbm_bytes"]) * int(r["dispatches"]) for r in rows if r["hbm_bytes"] != "nan") / passes
            traffic["train_step_b%d" % sites] = {"command": "bench.py " + command, "hbm_bytes_per_step": total,
                                     "source_hash": ident["source_hash"],
                                     "note": "sum over every kernel of the step of (2 x FETCH_SIZE + WRITE_SIZE) per dispatch x dispatches, "
                                             "/ %d steps; %d sites x 100 reads x 201 bp" % (passes, sites)}
        elif name == "bf16x3":
            for r in rows:
                if "segmentx_kernel" in r["kernel"]:
                    traffic["segmentx_kernel_bytes_per_launch"] = {
                        "command": "bench.py " + command, "total": float(r["hbm_bytes"]), "dispatches": int(r["dispatches"]), **ident,
                        "note": "chunk of 2048 sites x 64 reads x 201 bp, averaged over the two launches of a chunk (layers 1-2, layers 3-7)"}
        else:
            for r in rows:
                if "segmentp_kernel" in r["kernel"]:
                    traffic["segmentp_kernel_bytes_per_launch"] = {
                        "command": "bench.py " + command, "total": float(r["hbm_bytes"]), "dispatches": int(r["dispatches"]), **ident,
                        "note": "chunk of %s sites x 128 reads x 301 bp, averaged over the two launches of a chunk (layers 1-2, layers 3-7)" % ident["chunk_sites"]}
    json.dump(traffic, open(tpath, "w"), indent=1)
    lines = os.path.join(go, "lines_" + tag)
    for f in sorted(glob.glob(os.path.join(lines, "*.json"))):
        ls = [l for l in open(f) if l.startswith("{")]
        if ls:
            rec = json.loads(ls[-1])
            base = os.path.basename(f)[:-5]
            # (a line printed before its counters existed carries the traffic measured for the same command on the same build)
            def same(sec):
                t, b = traffic.get(sec), rec.get("build", {})
                return bool(t) and t.get("source_hash") == b.get("source_hash") and t.get("chunk_sites") in (None, b.get("chunk_sites"))
            for b_, sec, key in (("train", "train_step_b64", "hbm_bytes_per_step"), ("train_b10", "train_step_b10", "hbm_bytes_per_step"),
                                      ("bf16x3", "segmentx_kernel_bytes_per_launch", "total"), ("bf16_128x301", "segmentp_kernel_bytes_per_launch", "total")):
                if base == b_ and same(sec):
                    rec["roofline"]["traffic"] = int(traffic[sec][key])
                    rec["roofline"]["traffic_stale"] = False
            with open(os.path.join(dst, "%s_bench_line_%s.json" % (tag, base)), "w") as out:
                out.write(json.dumps(rec) + "\n")
    for f in sorted(glob.glob(os.path.join(lines, "*.txt"))):
        if os.path.basename(f) != "train_kernels.txt":
            shutil.copy(f, os.path.join(dst, "%s_%s" % (tag, os.path.basename(f))))
    gate = os.path.join(go, "genotype_gate.json")
    if os.path.isfile(gate):
        shutil.copy(gate, os.path.join(dst, tag + "_genotype_gate.json"))
    print("wrote profiles/%s_* (bf16, training, lines)" % tag)


if __name__ == "__main__":
    main()
