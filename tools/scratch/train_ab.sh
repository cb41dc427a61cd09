#!/bin/bash
mkdir -p gpurun_out/r2j
for mode in ${MODES:-grad}; do
  echo "== PWCLO_HIP_CONV=$mode"
  PWCLO_HIP_CONV=$mode timeout -k 10 300 python tools/train_step.py --batch 32 --steps 10 --warmup 3 --fused-adam 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.readlines()[-1]); print(d['value'], d['ms_per_step'], d.get('loss_first_last'))" || exit 1
done
timeout -k 10 300 python tools/train_step.py --batch 32 --steps 10 --warmup 3 --fused-adam --graph 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.readlines()[-1]); print('graph', d['value'], d['ms_per_step'], d.get('loss_first_last'))" || exit 1
timeout -k 10 600 python -m pytest tests/test_gpu_model.py tests/test_gpu_conv.py -x -q 2>&1 | tail -5
