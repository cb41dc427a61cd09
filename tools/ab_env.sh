#!/bin/bash
# Developer tool: alternating A/B of environment switches in the pipelined bench (one line per setting: runs and mean).
#   bash tools/ab_env.sh "PWCLO_COARSE_W4=0" "PWCLO_FL_ROUNDS=2" ...   (the empty setting = defaults is always included)
O=gpurun_out/ab_env; mkdir -p $O
B="--no-cpu-baseline --no-configs --no-variants --no-roofline"
settings=("" "$@")
for r in 1 2 3; do
  for i in "${!settings[@]}"; do
    env ${settings[$i]} python bench.py $B > $O/s${i}_r$r.json 2>/dev/null || echo "setting $i run $r failed"
  done
done
python - "$O" "${settings[@]}" <<'PY'
import json, sys
O, settings = sys.argv[1], sys.argv[2:]
for i, s in enumerate(settings):
    v = []
    for r in (1, 2, 3):
        try: v.append(json.loads(open("%s/s%d_r%d.json" % (O, i, r)).read().strip().splitlines()[-1])["value"])
        except Exception: pass
    print("%-40s %s  mean %.0f" % (s or "(defaults)", " ".join("%.0f" % x for x in v), sum(v) / max(len(v), 1)))
PY
