#!/bin/bash
# A/B kernel summaries of the training step: default vs one environment switch.   usage: profile_train_ab.sh VAR=VALUE
set -eo pipefail
export TMPDIR=/tmp
for tag in a b; do
  out=gpurun_out/prof_train_$tag
  mkdir -p $out
  if [ $tag = b ]; then export "$1"; fi
  rocprofv3 --kernel-trace --stats --output-format csv -d $out -o train -- python tools/train_step.py --batch ${2:-32} --steps 4 --warmup 1 > $out/out.txt 2>&1
  python - <<PY > $out/summary.txt
import csv, glob
f = glob.glob("$out/**/*kernel_stats.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
steps = 5
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print("kernel time per step: %.2f ms" % (tot / 1e6 / steps))
for r in rows[:70]:
    print(r["Name"][:110].ljust(110), r["Calls"].rjust(6), "%8.3f ms/step" % (float(r["TotalDurationNs"]) / 1e6 / steps), r["Percentage"])
PY
  find $out -name "*.db" -delete; find $out -name "*kernel_trace.csv" -delete; find $out -name "*agent_info.csv" -delete
done
