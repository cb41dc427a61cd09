#!/bin/bash
# Developer A/B of the round-3 launch folds (early cv partial products, predictor tails, one-kernel cost volume at level 3).
set -e
O=gpurun_out/r3b
mkdir -p $O
python -m pytest tests/test_gpu_fused.py tests/test_gpu_config2.py -x -q > $O/tests.log 2>&1 || { tail -30 $O/tests.log; exit 1; }
tail -3 $O/tests.log
python tools/launch_table.py > $O/lt_new.txt 2>&1
PWCLO_CV_MERGED_MIN=1025 python tools/launch_table.py > $O/lt_merged_min1025.txt 2>&1
PWCLO_CV_MERGED_MIN=1025 PWCLO_EARLY_CV=0 PWCLO_PW_TAIL=0 python tools/launch_table.py > $O/lt_old.txt 2>&1
B="--no-cpu-baseline --no-configs --no-variants"
python bench.py $B > $O/bench_new.json 2> $O/bench_new.err
PWCLO_CV_MERGED_MIN=1025 python bench.py $B > $O/bench_merged_min1025.json 2> $O/bench_b.err
PWCLO_CV_MERGED_MIN=1025 PWCLO_EARLY_CV=0 PWCLO_PW_TAIL=0 python bench.py $B > $O/bench_old.json 2> $O/bench_c.err
python bench.py $B > $O/bench_new2.json 2> $O/bench_new2.err
for f in new merged_min1025 old new2; do python - <<PY
import json
d=json.loads(open("$O/bench_$f.json").read().strip().splitlines()[-1])
print("$f", round(d["value"]), d["ms_per_step"], d["roofline"]["mlp_family"]["ms_per_step"], d["roofline"]["mlp_family"]["launches_per_step"], d["roofline"]["mlp_family"]["frac"])
PY
done
tail -1 $O/lt_new.txt $O/lt_merged_min1025.txt $O/lt_old.txt
