#!/bin/bash
run() { name=$1; shift; v=$(python bench.py --no-cpu-baseline --no-variants --no-roofline --repeats 5 --steps 20 "$@" 2>/dev/null | python -c "import json,sys; d=json.load(sys.stdin); print('%.0f pairs/s  %.3f ms' % (d['value'], d['ms_per_step']))"); echo "$name: $v"; }
run b32x4
run b64x2 --batch 64 --inflight 2
run b64x3 --batch 64 --inflight 3
run b64x4 --batch 64 --inflight 4
run b128x1 --batch 128 --inflight 1
run b128x2 --batch 128 --inflight 2
run b16x8 --batch 16 --inflight 8
