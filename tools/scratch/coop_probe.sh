#!/bin/bash
for h in 0 2 3; do for d in 12 0; do
  v=$(PWCLO_FPS_COOP_DEBUG_TIMEOUT=$h PWCLO_FPS_COOP_POLL_DELAY=$d timeout -k 10 200 python tools/scratch/coop_probe.py 2>/dev/null | tail -1)
  echo "probe $h delay $d: $v"
done; done
