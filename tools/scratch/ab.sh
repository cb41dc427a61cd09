#!/bin/bash
# quick A/B of pipelined throughput under env settings: each line "NAME ENV=VAL ..." 
run() { name=$1; shift; v=$(env "$@" python bench.py --no-cpu-baseline --no-variants --no-roofline --repeats 5 --steps 40 2>/dev/null | python -c "import json,sys; d=json.load(sys.stdin); print('%.0f pairs/s  %.3f ms' % (d['value'], d['ms_per_step']))"); echo "$name: $v"; }
run baseline X=1
run coarse_w4_off PWCLO_COARSE_W4=0
run rounds1 PWCLO_FL_ROUNDS=1
run rounds1_w4off PWCLO_FL_ROUNDS=1 PWCLO_COARSE_W4=0
run rounds3 PWCLO_FL_ROUNDS=3
run inflight1 X=1
