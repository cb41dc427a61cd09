#!/bin/bash
for v in "1 1" "4 8" "8 16" "4 16" "2 4"; do set -- $v; echo "settle16=$1 settle32=$2"; PWCLO_KNN_SETTLE16=$1 PWCLO_KNN_SETTLE32=$2 python tools/launch_table.py 2>/dev/null | grep "knn_point_ws" | awk '{s+=$2; printf "%s ", $2} END {print " sum", s}'; done
PWCLO_KNN_SETTLE16=4 PWCLO_KNN_SETTLE32=8 timeout -k 10 300 python -m pytest tests/test_gpu_ops.py -m gpu -q -x -k knn 2>&1 | tail -2
