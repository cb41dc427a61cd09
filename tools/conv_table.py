#!/usr/bin/env python3
"""Developer tool: every pointwise-convolution call of one training step (shape, count) timed in isolation --
forward (plain, and with the BatchNorm-statistics epilogue + its finish launch), input gradient, weight gradient -- with the HBM and
fp32-MFMA bounds beside it.

    python tools/conv_table.py [--batch 32]
"""
import argparse
import collections
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch  # noqa: E402

import bench  # noqa: E402
from pwclonet_pylidarslam_amd import _lib, conv1x1  # noqa: E402
from pwclonet_pylidarslam_amd.loss import PWCLONetLossModule  # noqa: E402
from pwclonet_pylidarslam_amd.pwclonet import PWCLONet  # noqa: E402
from pwclonet_pylidarslam_amd.training import PWCLONetWithLoss  # noqa: E402

HBM, MFMA = 8.0e12, 157.3e12


def timed(fn, reps=5):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e-3


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=32)
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    torch.manual_seed(7)
    net = PWCLONet(dict(num_input_channels=3, sequence_len=2, device=str(dev), scalar_last=False, log_mode="none")).to(dev).train()
    unit = PWCLONetWithLoss(net, PWCLONetLossModule(dict(with_exp_weights=True, init_weights=[0.0, -2.5], loss_option="l2_norm",
                                                         nb_levels=4, scalar_last=False)).to(dev))
    x1, x2 = bench.make_batch(a.batch, 8192, 2000, dev)
    gt = torch.zeros(a.batch, 7, device=dev)
    gt[:, 3] = 1.0
    shapes = collections.Counter()
    orig_call = _lib.call
    fwd_names = ("conv1x1_forward_kernel_wrapper", "conv1x1_bnrelu_forward_kernel_wrapper",
                 "conv1x1_forward_bnstats_kernel_wrapper")

    def spy(name, device, *args):          # every forward convolution of the step (B, cin, cout, pixels); dgrad calls are transposed
        if name in fwd_names and not (name == fwd_names[0] and args[6] == 1):
            shapes[tuple(int(v) for v in args[:4])] += 1
        return orig_call(name, device, *args)
    _lib.call = spy
    loss, _, _ = unit(x1, x2, gt)
    _lib.call = orig_call
    loss.backward()
    torch.cuda.synchronize()
    lib = _lib.load()
    tot = [0.0, 0.0, 0.0, 0.0]
    print("%4s %4s %4s %8s %3s | %21s | %9s | %21s | %21s" % ("B", "cin", "cout", "pixels", "n", "forward us (hbm/mfma)", "+stats us",
                                                              "dgrad us (hbm/mfma)", "wgrad us (hbm/mfma)"))
    for (B, cin, cout, P), n in sorted(shapes.items(), key=lambda kv: -kv[0][3] * kv[0][1] * kv[0][2]):
        x = torch.randn(B, cin, P, device=dev)
        w = torch.randn(cout, cin, device=dev)
        dy = torch.randn(B, cout, P, device=dev)
        y = torch.empty(B, cout, P, device=dev)
        dx = torch.empty(B, cin, P, device=dev)
        dw = torch.empty(cout, cin, device=dev)
        ws = torch.empty(lib.conv1x1_wgrad_workspace_bytes(B, cin, cout, P) // 4, device=dev)
        f = timed(lambda: _lib.call("conv1x1_forward_kernel_wrapper", dev, B, cin, cout, P, x.data_ptr(), w.data_ptr(), 0, y.data_ptr()))
        sws = torch.empty(lib.conv1x1_stats_workspace_bytes(B, cin, cout, P) // 8, dtype=torch.float64, device=dev)
        mean, invstd = torch.empty(cout, device=dev), torch.empty(cout, device=dev)
        fs = timed(lambda: _lib.call("conv1x1_forward_bnstats_kernel_wrapper", dev, B, cin, cout, P, x.data_ptr(), w.data_ptr(), 0, 0, 0,
                                     0, y.data_ptr(), 1e-5, 0.1, 0, 0, mean.data_ptr(), invstd.data_ptr(), sws.data_ptr()))
        d = timed(lambda: _lib.call("conv1x1_forward_kernel_wrapper", dev, B, cout, cin, P, dy.data_ptr(), w.data_ptr(), 1, dx.data_ptr()))
        g = timed(lambda: _lib.call("conv1x1_wgrad_kernel_wrapper", dev, B, cin, cout, P, dy.data_ptr(), x.data_ptr(), dw.data_ptr(), ws.data_ptr()))
        byt = 4.0 * B * P * (cin + cout)
        fl = 2.0 * B * P * cin * cout
        bound = "%5.0f/%5.0f" % (byt / HBM * 1e6, fl / MFMA * 1e6)
        print("%4d %4d %4d %8d %3d | %8.1f %12s | %9.1f | %8.1f %12s | %8.1f %12s" % (B, cin, cout, P, n, f * 1e6, bound, fs * 1e6, d * 1e6,
                                                                                    bound, g * 1e6, bound))
        tot[0] += n * f
        tot[1] += n * d
        tot[2] += n * g
        tot[3] += n * fs
    print("per step (every layer timed alone): forward %.2f ms (%.2f ms with the statistics epilogue + finish launch), input gradients "
          "%.2f ms, weight gradients %.2f ms" % (tot[0] * 1e3, tot[3] * 1e3, tot[1] * 1e3, tot[2] * 1e3))


if __name__ == "__main__":
    main()
