#!/bin/bash
for mode in grad all; do
  echo "== PWCLO_HIP_CONV=$mode"
  PWCLO_HIP_CONV=$mode timeout -k 10 300 python bench.py --unfused --no-cpu-baseline --no-variants --no-roofline --repeats 3 --steps 10 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.readlines()[-1]); print(d['value'], d['ms_per_step'])" || exit 1
done
PWCLO_HIP_CONV=all timeout -k 10 600 python -m pytest tests/test_gpu_model.py tests/test_gpu_modules.py tests/test_gpu_fused.py -x -q 2>&1 | tail -5
