#!/bin/bash
# Developer run (round 3, second session): full GPU suite on the final tree, a clean launch table, and the batch-1
# pipeline with more hardware queues / forwards in flight.
O=gpurun_out/r3d
mkdir -p $O
python -m pytest tests -m gpu -x -q > $O/tests.log 2>&1 || { tail -40 $O/tests.log; exit 1; }
tail -3 $O/tests.log
python tools/launch_table.py > $O/launch_table.txt 2>&1
tail -1 $O/launch_table.txt
B="--no-cpu-baseline --no-configs --no-variants --no-roofline --timed-seconds 1.5"
for q in 8 16 24; do for d in 4 6 8 12; do
  GPU_MAX_HW_QUEUES=$q python bench.py --batch 1 --inflight $d $B > $O/b1_q${q}_d$d.json 2> $O/b1_q${q}_d$d.err || echo "b1 q$q d$d failed"
done; done
for q in 12 16; do for d in 4 6; do
  GPU_MAX_HW_QUEUES=$q python bench.py --inflight $d $B > $O/b32_q${q}_d$d.json 2> $O/b32_q${q}_d$d.err || echo "b32 q$q d$d failed"
done; done
python - <<PY
import json, glob
for f in sorted(glob.glob("$O/b*.json")):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        print(f.split("/")[-1], round(d["value"]), round(d["ms_per_step"], 4), d["config"].get("batches_in_flight"), d["config"].get("hw_queues"))
    except Exception as e:
        print(f, "unreadable", e)
PY
