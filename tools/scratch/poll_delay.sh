#!/bin/bash
for d in 6 10 12 14 16; do
  v=$(PWCLO_FPS_COOP_POLL_DELAY=$d timeout -k 10 200 python bench.py --config 5 --no-cpu-baseline --repeats 3 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print(d['value'], d['ms_per_step'], d['stages_ms']['fps_95765_to_8192'])") || exit 1
  echo "delay $d: $v"
done
