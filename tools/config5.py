#!/usr/bin/env python3
"""Developer tool: BASELINE.json configs[4] end to end in fp32 -- a batch of raw KITTI-360-sized frame pairs
(8 x 2 x ~120k rows of x,y,z,intensity) -> on-device filter + compaction -> furthest point sampling to 8192
(cooperative multi-workgroup sampler) -> full pyramid -> poses.  Prints the time of each stage (HIP events)."""
import argparse
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import pwclonet_pylidarslam_amd  # noqa: F401
pwclonet_pylidarslam_amd.configure_hw_queues(8)
import torch

from pwclonet_pylidarslam_amd import preprocess
from pwclonet_pylidarslam_amd.pwclonet import PWCLONet


def raw_frames(seed, b, n, device):
    g = torch.Generator().manual_seed(seed)
    xy = torch.randn(b, n, 2, generator=g) * 18.0
    z = torch.rand(b, n, 1, generator=g) * 5.0 - 2.0
    inten = torch.rand(b, n, 1, generator=g)
    return torch.cat((xy, z, inten), dim=2).contiguous().to(device)


def timed(fn, reps=3):
    fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        out = fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps, out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=8)
    ap.add_argument("--rows", type=int, default=120000)
    ap.add_argument("--npoints", type=int, default=8192)
    ap.add_argument("--near", type=float, default=35.0)
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    torch.manual_seed(1234)
    net = PWCLONet(dict(num_input_channels=3, sequence_len=2, device=str(dev), scalar_last=False)).to(dev).eval()
    net.prepare_fused()
    f1 = raw_frames(1, args.batch, args.rows, dev)
    f2 = raw_frames(2, args.batch, args.rows, dev)
    frames = torch.cat((f1, f2), dim=0)                       # both frames of every pair in one batch

    t_filter, (xyz, keep) = timed(lambda: preprocess.kitti360_filter(frames, args.near))
    t_compact, (packed, counts) = timed(lambda: preprocess.compact(xyz, keep))
    t_all, (clouds, _) = timed(lambda: preprocess.frames_to_clouds(frames, args.npoints, dataset="kitti360",
                                                                   near_threshold=args.near), reps=2)
    c1, c2 = clouds[:args.batch].contiguous(), clouds[args.batch:].contiguous()

    def forward():
        with torch.no_grad():
            return net(c1.transpose(1, 2).contiguous(), None, c2.transpose(1, 2).contiguous(), None)[0]
    t_fwd, pose = timed(forward)
    cnt = counts.cpu()
    print(f"config 5 (fp32): {args.batch} pairs x 2 x {args.rows} rows, survivors {int(cnt.min())}..{int(cnt.max())}")
    print(f"  filter {t_filter:.3f} ms   compaction {t_compact:.3f} ms   filter+compaction+FPS->{args.npoints}+gather "
          f"{t_all:.1f} ms   pyramid forward (eager launches) {t_fwd:.2f} ms")
    print(f"  total {t_all + t_fwd:.1f} ms per batch = {args.batch / (t_all + t_fwd) * 1e3:.0f} pairs/s; "
          f"pose finite: {bool(torch.isfinite(pose).all())}")


if __name__ == "__main__":
    main()
