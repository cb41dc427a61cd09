#!/bin/bash
run() { name=$1; shift; v=$(env "$@" python bench.py --no-cpu-baseline --no-variants --no-roofline --repeats 5 --steps 40 $EXTRA 2>/dev/null | python -c "import json,sys; d=json.load(sys.stdin); print('%.0f pairs/s  %.3f ms' % (d['value'], d['ms_per_step']))"); echo "$name: $v"; }
for n in 3 4 5 6 8; do EXTRA="--inflight $n" run "inflight_$n" X=1; done
EXTRA="--inflight 6" run "inflight_6_q12" GPU_MAX_HW_QUEUES=12
EXTRA="--inflight 8" run "inflight_8_q16" GPU_MAX_HW_QUEUES=16
