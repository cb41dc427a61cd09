#!/usr/bin/env python3
"""Evidence for the HBM-roofline target of the grouping kernel (BASELINE north_star: >= 60 % of the memory roofline).

    rocprofv3 --kernel-trace --output-format csv -d DIR -o gp -- python tools/group_roofline.py run
    python tools/group_roofline.py report DIR/.../gp_kernel_trace.csv > profiles/rNN/rNN_group_points_roofline.txt

`run` launches `group_points` through the C ABI at the shapes one PWCLO-Net forward uses at batch 32 (SURVEY.md appendix
A; 25 launches each after 5 warm-up launches).  `report` takes the DEVICE-side durations of those dispatches from the
profiler's kernel trace (not wall time around a Python call), tells the shapes apart by their launch grid, and prints
algorithmic bytes 4*(S*K + C*N + C*S*K)*B / average duration against the 8 TB/s HBM peak.
"""
import csv
import os
import sys

SHAPES = [(64, 1024, 2048, 8), (64, 2048, 2048, 4), (64, 256, 256, 32), (64, 1024, 1024, 4), (32, 1024, 1024, 6),
          (16, 2048, 2048, 6), (16, 2048, 1024, 32), (32, 1024, 256, 16), (64, 256, 64, 16), (3, 8192, 2048, 32),
          (3, 2048, 2048, 6)]
B, WARM, REPS = 32, 5, 25
HBM_PEAK = 8.0e12


def run():
    import torch
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from pwclonet_pylidarslam_amd.pointnet2_ops import _ext as E
    dev = torch.device("cuda:0")
    for c, n, s, k in SHAPES:
        p = torch.randn(B, c, n, device=dev)
        idx = torch.randint(0, n, (B, s, k), device=dev, dtype=torch.int32)
        for _ in range(WARM + REPS):
            E.group_points(p, idx)
        torch.cuda.synchronize()


def report(path):
    rows = [r for r in csv.DictReader(open(path)) if "group_points" in r["Kernel_Name"]]
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    per = WARM + REPS
    assert len(rows) == per * len(SHAPES), (len(rows), per * len(SHAPES))
    print("group_points at batch %d: device-side duration (rocprofv3 --kernel-trace, average of the last %d of %d launches "
          "of each shape, in launch order)" % (B, REPS, per))
    print("%-22s %-26s %-16s %9s %10s %8s %8s" % ("C, N, S, K", "kernel", "grid (workgroups)", "us", "MB (alg.)", "GB/s",
                                                   "of 8TB/s"))
    for i, (c, n, s, k) in enumerate(SHAPES):
        chunk = rows[i * per:(i + 1) * per][-REPS:]
        d = [int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in chunk]
        us = sum(d) / len(d) / 1e3
        r0 = chunk[0]
        name = r0["Kernel_Name"].replace("pwclo::", "").replace("void ", "").split("(")[0]
        grid = "%dx%dx%d" % (int(r0["Grid_Size_X"]) // int(r0["Workgroup_Size_X"]), int(r0["Grid_Size_Y"]), int(r0["Grid_Size_Z"]))
        nbytes = 4.0 * B * (s * k + c * n + c * s * k)
        print("%-22s %-26s %-16s %9.2f %10.1f %8.0f %8.2f" % ("%d, %d, %d, %d" % (c, n, s, k), name, grid, us, nbytes / 1e6,
                                                             nbytes / us / 1e3, nbytes / (us * 1e-6) / HBM_PEAK))


if __name__ == "__main__":
    if len(sys.argv) >= 3 and sys.argv[1] == "report":
        report(sys.argv[2])
    else:
        run()
