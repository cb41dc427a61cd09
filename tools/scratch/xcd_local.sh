#!/bin/bash
timeout -k 10 400 python -m pytest tests/test_gpu_ops.py -x -q -k "fps" 2>&1 | tail -3 || exit 1
for x in 1 0; do for d in 20 8 12; do
  v=$(PWCLO_FPS_COOP_XCD_LOCAL=$x PWCLO_FPS_COOP_POLL_DELAY=$d timeout -k 10 200 python bench.py --config 5 --no-cpu-baseline --repeats 3 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print(d['value'], d['ms_per_step'], d['stages_ms']['fps_95765_to_8192'])") || exit 1
  echo "xcd_local $x delay $d: $v"
done; done
