#!/bin/bash
set -o pipefail
mkdir -p gpurun_out/r2f
timeout -k 10 400 python -m pytest tests/test_gpu_ops.py -m gpu -q -x -k knn > gpurun_out/r2f/knn_tests.log 2>&1; echo "knn tests rc=$?"; tail -4 gpurun_out/r2f/knn_tests.log
run() { name=$1; shift; v=$(env "$@" python bench.py --no-cpu-baseline --no-variants --no-roofline --repeats 5 --steps 40 2>/dev/null | python -c "import json,sys; d=json.load(sys.stdin); print('%.0f pairs/s  %.3f ms' % (d['value'], d['ms_per_step']))"); echo "$name: $v"; }
run rows_on X=1
run rows_off PWCLO_KNN_ROWS=0
PWCLO_KNN_ROWS=1 python tools/launch_table.py 2>/dev/null | grep "knn\|total" > gpurun_out/r2f/lt_rows_on.txt
PWCLO_KNN_ROWS=0 python tools/launch_table.py 2>/dev/null | grep "knn\|total" > gpurun_out/r2f/lt_rows_off.txt
paste gpurun_out/r2f/lt_rows_on.txt gpurun_out/r2f/lt_rows_off.txt | cut -c1-60,100-160
