#!/bin/bash
# rocprofv3 kernel summary of the training step (tools/train_step.py) -> gpurun_out/prof_train/summary.txt
set -eo pipefail
out=gpurun_out/prof_train
mkdir -p $out
export TMPDIR=/tmp
trap 'find $out -name "*.db" -delete; find $out -name "*kernel_trace.csv" -delete; find $out -name "*agent_info.csv" -delete' EXIT
rocprofv3 --kernel-trace --stats --output-format csv -d $out -o train -- python tools/train_step.py --batch ${1:-32} --steps 4 --warmup 1 > $out/out.txt 2>&1
python - <<PY | tee $out/summary.txt
import csv, glob
f = glob.glob("$out/**/*kernel_stats.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
steps = 5
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print("kernel time per step: %.2f ms" % (tot / 1e6 / steps))
for r in rows[:45]:
    print(r["Name"][:100].ljust(100), r["Calls"].rjust(6), "%8.2f ms/step" % (float(r["TotalDurationNs"]) / 1e6 / steps), r["Percentage"])
PY
