#!/bin/bash
# Developer tool: forwards in flight x hardware queues, batch 1 (BASELINE configs[1]) and batch 32 (configs[2]), and the staged
# pipeline.  Run through gpurun from the repo root; prints one line per setting (profiles/r03/r03_v4_inflight_sweep.txt,
# r03_v4_hw_queue_sweep.txt were made with it).
O=${1:-gpurun_out/sweep_inflight}
mkdir -p $O
B="--no-cpu-baseline --no-configs --no-variants --no-roofline --timed-seconds 1.5"
for q in 8 16 24; do for d in 4 6 8 12; do
  GPU_MAX_HW_QUEUES=$q python bench.py --batch 1 --inflight $d $B > $O/b1_q${q}_d$d.json 2> $O/b1_q${q}_d$d.err || echo "b1 q$q d$d failed"
done; done
for q in 8 12 16; do for d in 3 4 5 6 8; do
  GPU_MAX_HW_QUEUES=$q python bench.py --inflight $d $B > $O/b32_q${q}_d$d.json 2> $O/b32_q${q}_d$d.err || echo "b32 q$q d$d failed"
done; done
python bench.py --pipeline staged --inflight 4 $B > $O/b32_staged4.json 2> $O/b32_staged4.err || echo "staged failed"
python - <<PY
import json, glob
for f in sorted(glob.glob("$O/b*.json")):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        print(f.split("/")[-1], round(d["value"]), round(d["ms_per_step"], 4), d["config"].get("batches_in_flight"), d["config"].get("hw_queues"), d["config"].get("pipeline"))
    except Exception as e:
        print(f, "unreadable", e)
PY
