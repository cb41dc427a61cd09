#!/usr/bin/env python3
"""Turn two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; separate runs, TCC slots do not fit
both) into profiles/pmc_traffic.json: per kernel, average HBM-side bytes per launch.

    python tools/pmc_traffic.py <fetch counter_collection.csv> <write counter_collection.csv> -o profiles/pmc_traffic.json

Corrections as /opt/skills/guides/MI355X_MICROARCH.md "HBM" prescribes: both counters are in KiB;
on gfx950 FETCH_SIZE tallies 128-byte requests at 64 bytes for 16-byte-per-lane reads, so it is
doubled (checked here on fps_reg_kernel<512,...>: 3.13 MB reported vs 6.29 MB of coordinates
read exactly once); WRITE_SIZE is exact for 16-byte stores.  Counts are the L2's fabric-side
requests: Infinity-Cache hits are included."""
import argparse, csv, json, re, collections

ap = argparse.ArgumentParser()
ap.add_argument("fetch_csv")
ap.add_argument("write_csv")
ap.add_argument("-o", "--out", required=True)
ap.add_argument("--skip", type=int, default=0, help="ignore the first N dispatches of every kernel (warm-up passes)")
a = ap.parse_args()


def per_kernel(path, counter):
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] != counter or "pwclo::" not in r["Kernel_Name"]:
            continue
        name = re.sub(r"^void ", "", r["Kernel_Name"]).replace("pwclo::", "").split("(")[0]
        acc[name].append(float(r["Counter_Value"]) * 1024.0)
    return {k: v[a.skip:] if len(v) > a.skip else v for k, v in acc.items()}


fetch, write = per_kernel(a.fetch_csv, "FETCH_SIZE"), per_kernel(a.write_csv, "WRITE_SIZE")
out = {}
for k in sorted(set(fetch) | set(write)):
    f = fetch.get(k, [0.0]); w = write.get(k, [0.0])
    fb, wb = 2.0 * sum(f) / len(f), sum(w) / len(w)
    out[k] = {"launches_sampled": min(len(f), len(w)), "fetch_bytes_per_launch": fb, "write_bytes_per_launch": wb,
              "hbm_bytes_per_launch": fb + wb}
out["_method"] = ("rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) on tools/launch_table.py "
                  "--reps 1 at batch 32, 2x8192 points; FETCH_SIZE x2 (gfx950 128-byte requests tallied at 64), "
                  "KiB -> bytes; average over the launches of each kernel in the run (all problem sizes it is "
                  "launched at, like bench.py's avg_launch_us)")
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pwclonet_pylidarslam_amd.build import source_stamp
out["_source"] = {"csrc_sha16": source_stamp(), "git_head": os.environ.get("GIT_HEAD", "unknown (no .git on the GPU box)")}
json.dump(out, open(a.out, "w"), indent=1, sort_keys=True)
print("wrote", a.out, len(out) - 1, "kernels")
