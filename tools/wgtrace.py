#!/usr/bin/env python3
"""Developer tool: per-workgroup trace of the REAL pipelined run (rocprofv3's kernel trace serialises it).

Every workgroup of the kernels on the fused forward path records (kernel id, start, end) on the 100 MHz constant
clock (csrc/common.hpp: TraceScope, enabled through pwclo_trace_enable).  Prints, for a steady-state window:
  * per kernel family: workgroups, summed workgroup time ("CU-time" for the MFMA stacks, whose workgroups own a CU
    through their LDS footprint), mean / max workgroup duration;
  * how many workgroups of each family are resident, averaged over the window, and the share of the window during
    which fewer than 128 / 192 CU-owning workgroups (MFMA stacks + FPS) were resident.

    python tools/wgtrace.py [--steps 12] [--inflight 4] [--batch 32]
"""
import argparse
import ctypes
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
os.environ["PWCLO_TRACE_LIB"] = "1"          # the variant built by `python -m pwclonet_pylidarslam_amd.build --trace`
import pwclonet_pylidarslam_amd  # noqa: E402

pwclonet_pylidarslam_amd.configure_hw_queues(8)
import numpy as np  # noqa: E402
import torch  # noqa: E402

import bench  # noqa: E402
from pwclonet_pylidarslam_amd import _lib  # noqa: E402
from pwclonet_pylidarslam_amd.graphed import GraphedForward, PipelinedForward  # noqa: E402
from pwclonet_pylidarslam_amd.pwclonet import PWCLONet  # noqa: E402

NAMES = {1: "ingest", 2: "fps", 3: "knn", 4: "knn_build", 5: "knn_pruned", 6: "linear_jobs", 7: "sa_h", 8: "upconv_h",
         9: "upconv_lane", 10: "cv_a1_h", 11: "cv_a2", 12: "cv_a2_dense6", 13: "cv_a2_lane6", 14: "cv_b_h",
         15: "pointwise", 16: "pose_head", 17: "warp", 18: "other"}
CU_OWNERS = {2, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15}          # one workgroup per CU (LDS footprint)
MFMA = CU_OWNERS - {2}
SAMPLED = {5: 16, 3: 4, 1: 16}                                 # kernels that record every n-th workgroup only


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=12)
    ap.add_argument("--inflight", type=int, default=4)
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--cap", type=int, default=4 << 20)
    ap.add_argument("--dump", default=None, help="write the raw records of the window as .npy")
    ap.add_argument("--timeline", default=None, help="write a text timeline (resident workgroups per family per bin)")
    ap.add_argument("--bin-us", type=float, default=20.0)
    ap.add_argument("--ablate", default="", help="comma list of families replaced by stand-ins: fps,knn (tools/ablate.py)")
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    torch.manual_seed(1234)
    net = PWCLONet(dict(num_input_channels=3, sequence_len=2, device=str(dev), scalar_last=False,
                        log_mode="none")).to(dev).eval().prepare_fused()
    x1, x2 = bench.make_batch(a.batch, 8192, 1000, dev)
    if a.ablate:
        from pwclonet_pylidarslam_amd import fused

        def fake_fps(xyz, npoint, tie_out=None, tie_iters=0, prefix_in=None):
            idx = torch.arange(npoint, device=xyz.device, dtype=torch.int32).unsqueeze(0).expand(xyz.shape[0], -1).contiguous()
            if tie_out is not None:
                tie_out.zero_()
            return idx, xyz[:, :npoint].contiguous()

        def fake_knn(k, xyz, new_xyz):
            S = new_xyz.shape[1]
            return (torch.arange(S * k, device=xyz.device, dtype=torch.int32).reshape(1, S, k) % xyz.shape[1]) \
                .expand(xyz.shape[0], -1, -1).contiguous()
        if "fps" in a.ablate:
            fused.fps_with_xyz = fake_fps
        if "knn" in a.ablate:
            fused.knn = fake_knn
    pipe = PipelinedForward(net, depth=a.inflight) if a.inflight > 1 else GraphedForward(net)
    step = (lambda: pipe(x1, x2)[0]) if a.inflight > 1 else (lambda: pipe(x1, x2))
    for _ in range(6):
        step()
    torch.cuda.synchronize()
    rec = torch.zeros((a.cap, 4), dtype=torch.int64, device=dev)          # 32-byte records
    cnt = torch.zeros((1,), dtype=torch.int32, device=dev)
    lib = _lib.load()
    lib.pwclo_trace_enable(ctypes.c_void_p(rec.data_ptr()), ctypes.c_void_p(cnt.data_ptr()), a.cap)
    import time
    torch.cuda.synchronize()
    t_host = time.perf_counter()
    for _ in range(a.steps):
        step()
    torch.cuda.synchronize()
    t_host = time.perf_counter() - t_host
    lib.pwclo_trace_enable(None, None, 0)
    n = min(int(cnt.item()), a.cap)
    raw = rec[:n].cpu().numpy()
    t0, t1 = raw[:, 0].astype(np.float64), raw[:, 1].astype(np.float64)
    meta = np.ascontiguousarray(raw[:, 2]).view(np.uint32).reshape(-1, 2)
    kid = meta[:, 0].astype(np.int64)
    base = t0.min()
    t0, t1 = (t0 - base) * 0.01, (t1 - base) * 0.01                        # microseconds (100 MHz clock)
    span = t1.max()
    print("traced %d workgroups over %.2f ms of device time; host: %.3f ms per step (%d steps, %d in flight)"
          % (n, span / 1e3, 1e3 * t_host / a.steps, a.steps, a.inflight))
    w0, w1 = 0.25 * span, 0.75 * span                                       # steady-state window
    win = w1 - w0
    steps_in_window = a.steps * win / span
    print("window %.2f ms = %.2f steps" % (win / 1e3, steps_in_window))
    print("%-14s %9s %14s %12s %10s %10s %12s" % ("kernel", "wgs/step", "wg-ms/step", "CU-share", "mean us", "max us",
                                                 "resident avg"))
    tot_owner = 0.0
    for k in sorted(set(kid.tolist())):
        m = kid == k
        a0, a1 = np.clip(t0[m], w0, w1), np.clip(t1[m], w0, w1)
        inside = a1 > a0
        rate = SAMPLED.get(k, 1)
        busy = float((a1 - a0).sum()) * rate
        dur = (t1[m] - t0[m])[inside]
        if not inside.any():
            continue
        if k in CU_OWNERS:
            tot_owner += busy
        print("%-14s %9.0f %14.3f %11.1f%% %10.1f %10.1f %12.1f" % (
            NAMES.get(k, str(k)), rate * inside.sum() / steps_in_window, busy / 1e3 / steps_in_window,
            100.0 * busy / (win * 256.0) if k in CU_OWNERS else float("nan"), dur.mean(), dur.max(), busy / win))
    print("CU-owning workgroups (MFMA stacks + FPS): %.1f resident on average = %.1f %% of 256 CUs"
          % (tot_owner / win, 100.0 * tot_owner / win / 256.0))
    # residency histogram of CU-owning workgroups
    m = np.isin(kid, list(CU_OWNERS))
    ev = np.concatenate((np.stack((t0[m], np.ones(m.sum())), 1), np.stack((t1[m], -np.ones(m.sum())), 1)))
    ev = ev[np.argsort(ev[:, 0], kind="stable")]
    level = np.cumsum(ev[:, 1])
    ts = ev[:, 0]
    seg = np.diff(np.append(ts, ts[-1]))
    inwin = (ts >= w0) & (ts < w1)
    for thr in (64, 128, 192, 240):
        print("  fewer than %3d CU-owning workgroups resident: %5.1f %% of the window"
              % (thr, 100.0 * seg[inwin & (level < thr)].sum() / max(seg[inwin].sum(), 1e-9)))
    m = np.isin(kid, list(MFMA))
    ev = np.concatenate((np.stack((t0[m], np.ones(m.sum())), 1), np.stack((t1[m], -np.ones(m.sum())), 1)))
    ev = ev[np.argsort(ev[:, 0], kind="stable")]
    level, ts = np.cumsum(ev[:, 1]), ev[:, 0]
    seg = np.diff(np.append(ts, ts[-1]))
    inwin = (ts >= w0) & (ts < w1)
    for thr in (1, 16, 64, 128, 180):
        print("  fewer than %3d MFMA-stack workgroups resident: %5.1f %% of the window"
              % (thr, 100.0 * seg[inwin & (level < thr)].sum() / max(seg[inwin].sum(), 1e-9)))
    if a.timeline:
        groups = [("fps", {2}), ("knn", {3, 4, 5}), ("mfma", MFMA), ("other", {1, 16, 17, 18})]
        nb = int((w1 - w0) / a.bin_us)
        with open(a.timeline, "w") as f:
            f.write("# bin %.0f us; resident workgroups (time-averaged over the bin): fps knn(x rate) mfma other | dominant mfma kernels\n" % a.bin_us)
            cols = {}
            for name, ids in groups:
                acc = np.zeros(nb)
                for k in ids:
                    mk = kid == k
                    if not mk.any():
                        continue
                    rate = SAMPLED.get(k, 1)
                    for s_, e_ in zip(t0[mk], t1[mk]):
                        if e_ <= w0 or s_ >= w1:
                            continue
                        b0, b1 = max(s_, w0) - w0, min(e_, w1) - w0
                        i0, i1 = int(b0 / a.bin_us), min(int(b1 / a.bin_us), nb - 1)
                        if i0 == i1:
                            acc[i0] += rate * (b1 - b0) / a.bin_us
                        else:
                            acc[i0] += rate * ((i0 + 1) * a.bin_us - b0) / a.bin_us
                            acc[i0 + 1:i1] += rate
                            acc[i1] += rate * (b1 - i1 * a.bin_us) / a.bin_us
                cols[name] = acc
            per = {}
            for k in MFMA:
                mk = kid == k
                acc = np.zeros(nb)
                for s_, e_ in zip(t0[mk], t1[mk]):
                    if e_ <= w0 or s_ >= w1:
                        continue
                    i0, i1 = int((max(s_, w0) - w0) / a.bin_us), min(int((min(e_, w1) - w0) / a.bin_us), nb - 1)
                    acc[i0:i1 + 1] += 1
                per[k] = acc
            for i in range(nb):
                dom = sorted(((per[k][i], NAMES[k]) for k in per if per[k][i] > 0), reverse=True)[:3]
                f.write("%8.0f %5.0f %6.0f %5.0f %4.0f | %s\n" % (i * a.bin_us, cols["fps"][i], cols["knn"][i], cols["mfma"][i],
                                                             cols["other"][i], " ".join("%s:%d" % (n_, c_) for c_, n_ in dom)))
    if a.dump:
        np.save(a.dump, np.stack((t0, t1, kid.astype(np.float64), meta[:, 1].astype(np.float64)), 1)[(t1 > w0) & (t0 < w1)])


if __name__ == "__main__":
    main()
