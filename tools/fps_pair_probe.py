#!/usr/bin/env python3
"""Do several furthest-point-sampling chains share a compute unit productively?  (round 3 experiment)

Times the level-1 call (8192 -> 2048) for 64 / 256 / 512 / 1024 clouds: with ONE workgroup per cloud and 256 CUs, 256 clouds are
one chain per CU; 512 / 1024 clouds are two / four per CU when the kernel's LDS / registers allow them to be co-resident,
otherwise the extra workgroups queue and the time doubles.  Variants: the register-resident kernel (LDS table / global
table) and the bucket-pruned slab kernel (LDS table / global table)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from pwclonet_pylidarslam_amd import fused

dev = torch.device("cuda:0")
def timeit(fn, reps=3):
    fn(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / reps
which = sys.argv[1] if len(sys.argv) > 1 else "reg"
x1, x2 = bench.make_batch(32, 8192, 1000, dev)
base = torch.cat((x1, x2)).permute(0, 2, 1).contiguous()     # 64 clouds
for nb in (64, 256, 512, 1024):
    x = base.repeat(nb // 64, 1, 1).contiguous()
    if which == "slab":
        ms = timeit(lambda: fused.fps_slab_with_xyz(x, 2048))
    else:
        ms = timeit(lambda: fused.fps_with_xyz(x, 2048))
    print("%s table=%s/%s clouds %4d: %8.3f ms  (%.3f ms per 256 clouds)" % (which, os.environ.get("PWCLO_FPS_TABLE", "1"),
          os.environ.get("PWCLO_FPS_SLAB_TABLE", "1"), nb, ms, ms * 256 / max(nb, 256)))
