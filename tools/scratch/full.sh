#!/bin/bash
mkdir -p gpurun_out/r2i
timeout -k 10 900 python -m pytest tests -m gpu -q -s > gpurun_out/r2i/tests.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -4 gpurun_out/r2i/tests.log
if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then exit 1; fi
run() { name=$1; shift; v=$(env "$@" python bench.py --no-cpu-baseline --no-variants --no-roofline --repeats 5 --steps 40 2>/dev/null | python -c "import json,sys; d=json.load(sys.stdin); print('%.0f pairs/s  %.3f ms' % (d['value'], d['ms_per_step']))"); echo "$name: $v"; }
run kmajor_on X=1
run kmajor_off PWCLO_SA_KMAJOR=0
python tools/launch_table.py 2>/dev/null | grep "sa_fused_h" | head -2
timeout -k 10 120 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
