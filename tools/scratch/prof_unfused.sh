#!/bin/bash
out=gpurun_out/prof_unfused; mkdir -p $out; export TMPDIR=/tmp
trap 'find $out -name "*.db" -delete; find $out -name "*kernel_trace.csv" -delete; find $out -name "*agent_info.csv" -delete' EXIT
rocprofv3 --kernel-trace --stats --output-format csv -d $out -o un -- python bench.py --unfused --launch eager --inflight 1 --no-cpu-baseline --no-variants --no-roofline --repeats 1 --steps 6 --warmup 2 > $out/out.txt 2>&1
python - <<PY
import csv, glob
f = glob.glob("$out/**/*kernel_stats.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print("total kernel ms", tot/1e6)
for r in rows[:28]:
    print(r["Name"][:90].ljust(90), r["Calls"].rjust(6), "%8.2f ms" % (float(r["TotalDurationNs"]) / 1e6), r["Percentage"])
PY
