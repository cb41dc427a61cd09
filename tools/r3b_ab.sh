#!/bin/bash
# Developer A/B (round 3, second session): full GPU suite, then in-flight depth sweeps of configs[1] (batch 1) and configs[2].
O=gpurun_out/r3c
mkdir -p $O
python -m pytest tests -m gpu -x -q > $O/tests.log 2>&1 || { tail -40 $O/tests.log; exit 1; }
tail -3 $O/tests.log
B="--no-cpu-baseline --no-configs --no-variants --no-roofline --timed-seconds 1.5"
for d in 4 8 12 16; do
  python bench.py --batch 1 --inflight $d $B > $O/b1_inflight$d.json 2> $O/b1_inflight$d.err || echo "b1 inflight $d failed"
done
for d in 3 4 5 6 8; do
  python bench.py --inflight $d $B > $O/b32_inflight$d.json 2> $O/b32_inflight$d.err || echo "b32 inflight $d failed"
done
python bench.py --pipeline staged --inflight 4 $B > $O/b32_staged4.json 2> $O/b32_staged4.err || echo "staged failed"
python bench.py --batch 4 --inflight 8 $B > $O/b4_inflight8.json 2> $O/b4_inflight8.err || echo "b4 failed"
python - <<PY
import json, glob
for f in sorted(glob.glob("$O/b*.json")):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        print(f.split("/")[-1], round(d["value"]), round(d["ms_per_step"], 4), d["config"].get("batches_in_flight"))
    except Exception as e:
        print(f, "unreadable", e)
PY
