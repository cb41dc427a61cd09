#!/bin/bash
run() { name=$1; shift; v=$(python bench.py --no-cpu-baseline --no-variants --no-roofline --repeats 5 --steps 40 "$@" 2>/dev/null | python -c "import json,sys; d=json.load(sys.stdin); print('%.0f pairs/s  %.3f ms  [%s]' % (d['value'], d['ms_per_step'], ' '.join('%.3f' % t for t in d['repeats']['ms_per_step_all'])))"); echo "$name: $v"; }
run base
run stagger --stagger 1
run stagger3 --stagger 1 --inflight 3
run stagger5 --stagger 1 --inflight 5
run stagger6 --stagger 1 --inflight 6
