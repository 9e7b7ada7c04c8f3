#!/bin/bash
# Kernel trace of the training bench (run on the GPU box from the repository root): tools/profile_train.sh r02
# -> gpurun_out/prof_train_<tag>/kt/**/*kernel_stats.csv and a short table on stdout.
set -eo pipefail
tag=${1:-r02}
out=gpurun_out/prof_train_$tag
mkdir -p "$out"
export TMPDIR=/tmp
root=$PWD
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$root/$out/kt" -- python3 "$root/bench.py" --mode train ${TRAIN_ARGS:-} --steps 5 --warmup 2 --no-cpu-baseline > "$root/$out/bench_under_rocprof.json" 2> "$root/$out/kt.log"
cd "$root"
python3 - "$out" <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/kt/**/*kernel_stats.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
for r in rows[:16]:
    print("%-58s %6s calls %12s ns total %10s ns avg %6s %%" % (r["Name"][:58], r["Calls"], r["TotalDurationNs"], r["AverageNs"], r["Percentage"]))
PY
# the last step's launches in order (name, microseconds): which launch of a kernel is the slow one
python3 - "$out" <<'PY' > "$out/last_step_sequence.txt"
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/kt/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
names = [r["Kernel_Name"] for r in rows]
last = max(i for i, n in enumerate(names) if "__amd_rocclr_fillBufferAligned" in n and i + 1 < len(names) and "pack" in names[i + 1]) if any("pack" in n for n in names) else 0
t0 = int(rows[last]["Start_Timestamp"])
for r in rows[last:]:
    print("%9.1f %8.1f  %s" % ((int(r["Start_Timestamp"]) - t0) / 1e3, (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3, r["Kernel_Name"][:90]))
PY
