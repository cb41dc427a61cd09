#!/bin/bash
run() { name=$1; shift; v=$(env "$@" python bench.py --no-cpu-baseline --no-variants --no-roofline --repeats 5 --steps 40 2>/dev/null | python -c "import json,sys; d=json.load(sys.stdin); print('%.0f pairs/s  %.3f ms' % (d['value'], d['ms_per_step']))"); echo "$name: $v"; }
run base X=1
run n256_s256 PWCLO_KNN_MIN_N=256 PWCLO_KNN_MIN_S=256
run n128_s64 PWCLO_KNN_MIN_N=128 PWCLO_KNN_MIN_S=64
run n64_s64 PWCLO_KNN_MIN_N=64 PWCLO_KNN_MIN_S=64
PWCLO_KNN_MIN_N=64 PWCLO_KNN_MIN_S=64 python tools/launch_table.py 2>/dev/null | grep "knn\|total"
PWCLO_KNN_MIN_N=64 PWCLO_KNN_MIN_S=64 timeout -k 10 300 python -m pytest tests/test_gpu_ops.py -m gpu -q -x -k knn 2>&1 | tail -2
