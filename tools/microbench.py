#!/usr/bin/env python3
"""Per-kernel microbenchmark at the shapes of one PWCLO-Net forward (developer tool).

    python tools/microbench.py [--batch 32] [--ops fps,knn,group,...] [--reps 20]

Prints, per call shape: average kernel time (HIP events on the launch stream, back-to-back
launches), algorithmic bytes or work, and the achieved rate.  Used to fill DESIGN.md's roofline
table; bench.py does the headline measurement.
"""
import argparse
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pwclonet_pylidarslam_amd import synthetic  # noqa: E402
from pwclonet_pylidarslam_amd.pointnet2_ops import _ext as E  # noqa: E402


def timeit(fn, reps, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) * 1e3 / reps  # us


def clouds(batch, n, dev, seed=5):
    pc1, _, _, _ = synthetic.kitti_like_pair(seed, 8192, min(batch, 4))
    x = torch.from_numpy(np.ascontiguousarray(pc1[:, :, :3]))
    reps = (batch + x.shape[0] - 1) // x.shape[0]
    x = x.repeat(reps, 1, 1)[:batch]
    x = x + torch.arange(batch).reshape(-1, 1, 1) * 1e-3
    if n < 8192:
        x = x[:, torch.randperm(8192, generator=torch.Generator().manual_seed(1))[:n]]
    return x.contiguous().to(dev)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--reps", type=int, default=20)
    ap.add_argument("--ops", default="fps,knn,group,gather,warp,ball,three,grads")
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    B = a.batch
    ops = set(a.ops.split(","))
    print("device:", torch.cuda.get_device_name(0), "batch", B)

    if "fps" in ops:
        for n, m in ((8192, 2048), (2048, 1024), (1024, 256), (256, 64)):
            x = clouds(B, n, dev)
            us = timeit(lambda: E.furthest_point_sampling(x, m), a.reps)
            print(f"fps   N={n:5d} m={m:5d}: {us:9.1f} us  {1e3 * us / (m - 1):7.1f} ns/iter "
                  f"{B * (m - 1) * n / us / 1e3:8.2f} Gpoint-visits/s")
    if "knn" in ops:
        shapes = [(32, 8192, 2048), (32, 2048, 1024), (16, 1024, 256), (16, 256, 64), (32, 256, 256),
                  (4, 256, 256), (8, 64, 256), (6, 256, 256), (8, 256, 1024), (6, 1024, 1024),
                  (4, 1024, 1024), (8, 1024, 2048), (6, 2048, 2048), (4, 2048, 2048)]
        for k, n, s in shapes:
            x = clouds(B, n, dev)
            # queries as in the model: samples of the same scene (own cloud, or the other frame)
            q = (x[:, :s] if s <= n else clouds(B, s, dev))[:, :s].contiguous() + 0.01
            us = timeit(lambda: E.knn_point(k, x, q), a.reps)
            us0 = timeit(lambda: E.knn_point(k, x, q, exhaustive=True), a.reps)
            print(f"knn   K={k:2d} N={n:5d} S={s:5d}: {us:9.1f} us  (exhaustive {us0:8.1f} us, "
                  f"{B * s * n / us0 / 1e3:7.1f} Gdist/s)")
    if "group" in ops:
        shapes = [(3, 8192, 2048, 32), (16, 2048, 1024, 32), (32, 1024, 256, 16), (64, 256, 64, 16),
                  (64, 256, 256, 32), (64, 1024, 2048, 8), (64, 2048, 2048, 4), (16, 2048, 2048, 6),
                  (3, 2048, 2048, 6)]
        for c, n, s, k in shapes:
            p = torch.randn(B, c, n, device=dev)
            idx = torch.randint(0, n, (B, s, k), device=dev, dtype=torch.int32)
            us = timeit(lambda: E.group_points(p, idx), a.reps)
            nbytes = 4.0 * B * (s * k + c * n + c * s * k)
            print(f"group C={c:3d} N={n:5d} S={s:5d} K={k:2d}: {us:9.1f} us  {nbytes / us / 1e3:8.1f} GB/s "
                  f"({nbytes / 1e6:.1f} MB)")
    if "gather" in ops:
        for c, n, m in ((3, 8192, 2048), (3, 2048, 1024), (3, 256, 64)):
            p = torch.randn(B, c, n, device=dev)
            idx = torch.randint(0, n, (B, m), device=dev, dtype=torch.int32)
            us = timeit(lambda: E.gather_points(p, idx), a.reps)
            print(f"gather C={c} N={n} M={m}: {us:9.1f} us")
    if "warp" in ops:
        for n in (2048, 1024, 256):
            x = torch.randn(B, 3, n, device=dev)
            q = torch.randn(B, 4, 1, device=dev)
            t = torch.randn(B, 3, 1, device=dev)
            us = timeit(lambda: E.quat_warp(x, q, t), a.reps)
            print(f"warp  N={n}: {us:9.1f} us  {24.0 * B * n / us / 1e3:8.1f} GB/s")
    if "ball" in ops:
        for m, n, k, r in ((2048, 8192, 32, 0.5), (1024, 2048, 32, 1.0), (256, 1024, 16, 2.0), (64, 256, 16, 4.0)):
            x = clouds(B, n, dev)
            q = x[:, :m].contiguous()
            us = timeit(lambda: E.ball_query(q, x, r, k), a.reps)
            nbytes = B * (12.0 * (m + n) + 4.0 * m * k)
            print(f"ball  M={m} N={n} K={k} r={r}: {us:9.1f} us  {nbytes / us / 1e3:8.1f} GB/s  "
                  f"{B * m * n / us / 1e3:8.1f} Gtest/s")
    if "three" in ops:
        for n, m in ((256, 64), (1024, 256), (2048, 1024)):
            u, kn = clouds(B, n, dev), clouds(B, m, dev, seed=7)
            us = timeit(lambda: E.three_nn(u, kn), a.reps)
            print(f"three_nn n={n} m={m}: {us:9.1f} us  {B * (12.0 * (n + m) + 24.0 * n) / us / 1e3:8.1f} GB/s  "
                  f"{B * n * m / us / 1e3:8.1f} Gdist/s")
            c = 64
            feats = torch.randn(B, c, m, device=dev)
            dist2, idx = E.three_nn(u, kn)
            w = torch.softmax(-dist2, dim=2).contiguous()
            us = timeit(lambda: E.three_interpolate(feats, idx, w), a.reps)
            nbytes = B * (4.0 * (c * m + c * n) + 24.0 * n)
            print(f"three_interpolate c={c} n={n} m={m}: {us:9.1f} us  {nbytes / us / 1e3:8.1f} GB/s")
            go = torch.randn(B, c, n, device=dev)
            us = timeit(lambda: E.three_interpolate_grad(go, idx, w, m), a.reps)
            print(f"three_interpolate_grad c={c} n={n} m={m}: {us:9.1f} us  {nbytes / us / 1e3:8.1f} GB/s")
    if "grads" in ops:
        for c, n, s, k in ((64, 1024, 2048, 8), (16, 2048, 1024, 32), (64, 256, 256, 32)):
            idx = torch.randint(0, n, (B, s, k), device=dev, dtype=torch.int32)
            go = torch.randn(B, c, s, k, device=dev)
            nbytes = 4.0 * B * (s * k + c * n + c * s * k)
            us = timeit(lambda: E.group_points_grad(go, idx, n), a.reps)
            us2 = timeit(lambda: E.scatter_grad_deterministic(go, idx, n), a.reps)
            print(f"group_points_grad C={c} N={n} S={s} K={k}: {us:9.1f} us  {nbytes / us / 1e3:8.1f} GB/s  "
                  f"(deterministic incl. sort: {us2:9.1f} us)")


if __name__ == "__main__":
    main()
