#!/bin/bash
set -eo pipefail
root=$PWD; out=gpurun_out/prof_train_b10; mkdir -p $out
export TMPDIR=/tmp; cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$root/$out/kt" -- python3 "$root/bench.py" --mode train --train-batch 10 --steps 10 --warmup 2 --no-cpu-baseline > "$root/$out/line.json" 2> "$root/$out/kt.log"
cd "$root"
python3 - "$out" <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/kt/**/*kernel_stats.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print("total kernel time per step: %.2f ms" % (tot / 12 / 1e6))
for r in rows[:28]:
    print("%-60s %5s calls %8.3f ms/step %6s %%" % (r["Name"][:60], r["Calls"], float(r["TotalDurationNs"]) / 12 / 1e6, r["Percentage"]))
PY
