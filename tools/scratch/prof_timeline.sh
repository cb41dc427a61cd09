#!/bin/bash
set -o pipefail
out=gpurun_out/r2d
mkdir -p $out
export TMPDIR=/tmp
timeout -k 10 300 python -m pytest tests/test_gpu_eval.py tests/test_gpu_model.py -m gpu -q -s -k "evaluator or ddp or rows_to or array_functions" > $out/tests.log 2>&1; echo "pytest rc=$?"; tail -3 $out/tests.log
rocprofv3 --kernel-trace --output-format csv -d $out/trace -o t -- python bench.py --no-cpu-baseline --no-variants --no-roofline --repeats 3 --steps 20 > $out/bench_under_rocprof.json 2> $out/rocprof.err
echo "rocprof rc=$?"
f=$(find $out/trace -name '*kernel_trace.csv' | head -1)
python tools/timeline.py $f --window-ms 9 > $out/timeline.txt 2>&1
python tools/timeline.py $f --window-ms 4.6 --list > $out/timeline_list.txt 2>&1
find $out -name "*.db" -delete; find $out -name "*kernel_trace.csv" -delete; find $out -name "*agent_info.csv" -delete
timeout -k 10 200 python tools/ablate.py > $out/ablate.txt 2>&1; echo "ablate rc=$?"
cat $out/timeline.txt | head -40; cat $out/ablate.txt
