#!/usr/bin/env python3
"""Developer tool: per-launch table of one eager fused forward (HIP events on the launch stream):
kernel, time, algorithmic GFLOP / MB, achieved rate.  `--reps R` averages over R passes."""
import argparse, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from pwclonet_pylidarslam_amd import _lib
from pwclonet_pylidarslam_amd.pwclonet import PWCLONet


class Rec:
    def __init__(self): self.rows = []
    def add(self, name, meta, s, e): self.rows.append((name, meta or {}, s, e))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--npoints", type=int, default=8192)
    ap.add_argument("--reps", type=int, default=5)
    ap.add_argument("--fake-knn", action="store_true", help="sequential neighbour lists (gather-cost probe)")
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    if a.fake_knn:
        from pwclonet_pylidarslam_amd import fused
        def fake(k, xyz, new_xyz):
            S = new_xyz.shape[1]
            return (torch.arange(S * k, device=xyz.device, dtype=torch.int32).reshape(1, S, k) % xyz.shape[1]) \
                .expand(xyz.shape[0], -1, -1).contiguous()
        fused.knn = fake
    torch.manual_seed(0)
    net = PWCLONet(dict(num_input_channels=3, sequence_len=2, device="cuda:0", scalar_last=False,
                        log_mode="none")).to(dev).eval().prepare_fused()
    x1, x2 = bench.make_batch(a.batch, a.npoints, 1000, dev)
    with torch.no_grad():
        for _ in range(3): net(x1, None, x2, None)
        torch.cuda.synchronize()
        acc = None
        for _ in range(a.reps):
            rec = Rec(); _lib.profiler = rec
            net(x1, None, x2, None)
            _lib.profiler = None
            torch.cuda.synchronize()
            us = [s.elapsed_time(e) * 1e3 for _, _, s, e in rec.rows]
            acc = us if acc is None else [x + y for x, y in zip(acc, us)]
    tot = 0.0
    for (name, meta, _, _), t in zip(rec.rows, acc):
        t /= a.reps; tot += t
        fl, by = meta.get("flops", 0.0), meta.get("bytes", 0.0)
        print(f"{name.replace('_kernel_wrapper',''):32s} {t:8.1f} us  {fl/1e9:7.2f} GF {fl/t/1e6 if fl else 0:6.1f} TF/s  "
              f"{by/1e6:7.1f} MB {by/t/1e3 if by else 0:7.0f} GB/s")
    print("total %.1f us over %d launches" % (tot, len(acc)))


if __name__ == "__main__":
    main()
