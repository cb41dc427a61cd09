#!/bin/bash
set -o pipefail
mkdir -p gpurun_out/r2h
timeout -k 10 500 python -m pytest tests/test_gpu_fused.py -m gpu -q -x -k "fps or network or determin or prediction" > gpurun_out/r2h/t.log 2>&1; echo "tests rc=$?"; tail -4 gpurun_out/r2h/t.log
run() { name=$1; shift; v=$(env "$@" python bench.py --no-cpu-baseline --no-variants --no-roofline --repeats 5 --steps 40 2>/dev/null | python -c "import json,sys; d=json.load(sys.stdin); print('%.0f pairs/s  %.3f ms' % (d['value'], d['ms_per_step']))"); echo "$name: $v"; }
run slab_on X=1
run slab_off PWCLO_FPS_SLAB=0
python tools/launch_table.py 2>/dev/null | head -12
