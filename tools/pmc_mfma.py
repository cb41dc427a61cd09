#!/usr/bin/env python3
"""Turn a rocprofv3 --pmc pass over SQ counters into a per-kernel matrix-pipe utilisation table.

    rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU_MFMA_MOPS_F32 \
        --output-format csv -d <dir> -o mfma -- python tools/launch_table.py --reps 1
    python tools/pmc_mfma.py <dir>/.../mfma_counter_collection.csv -o profiles/r01/vNN_pmc_mfma_busy.json

Per kernel (summed over its launches): mfma_busy = SQ_VALU_MFMA_BUSY_CYCLES / (4 * SQ_BUSY_CU_CYCLES) -- the share of
SIMD time the matrix pipe was executing (4 SIMDs per CU) -- and valu_share = SQ_ACTIVE_INST_VALU / SQ_BUSY_CU_CYCLES,
the share spent issuing other vector instructions (the two add up to ~1 on the stack kernels)."""
import argparse, collections, csv, json, re

ap = argparse.ArgumentParser()
ap.add_argument("csv")
ap.add_argument("-o", "--out", required=True)
a = ap.parse_args()
acc = collections.defaultdict(lambda: collections.defaultdict(float))
calls = collections.Counter()
for r in csv.DictReader(open(a.csv)):
    if "pwclo::" not in r["Kernel_Name"]:
        continue
    name = re.sub(r"^void ", "", r["Kernel_Name"]).replace("pwclo::", "").split("(")[0]
    acc[name][r["Counter_Name"]] += float(r["Counter_Value"])
    if r["Counter_Name"] == "SQ_BUSY_CU_CYCLES":
        calls[name] += 1
out = {}
for k, c in sorted(acc.items()):
    busy = c.get("SQ_BUSY_CU_CYCLES", 0.0)
    if busy <= 0:
        continue
    out[k] = {"launches": calls[k], "mfma_busy": c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / (4.0 * busy),
              "valu_share": c.get("SQ_ACTIVE_INST_VALU", 0.0) / busy,
              "counters": {n: v for n, v in sorted(c.items())}}
out["_method"] = ("rocprofv3 --pmc (one pass, SQ block) over tools/launch_table.py --reps 1 at batch 32, 2x8192 points; "
                  "sums over all launches of a kernel")
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pwclonet_pylidarslam_amd.build import source_stamp
out["_source"] = {"csrc_sha16": source_stamp(), "git_head": os.environ.get("GIT_HEAD", "unknown (no .git on the GPU box)")}
json.dump(out, open(a.out, "w"), indent=1, sort_keys=True)
for k, v in out.items():
    if not k.startswith("_") and v["mfma_busy"] > 0.01:
        print(f"{k:48s} launches {v['launches']:3d}  mfma_busy {v['mfma_busy']:.3f}  valu_share {v['valu_share']:.3f}")
