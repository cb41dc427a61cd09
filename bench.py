#!/usr/bin/env python3
"""Headline benchmark: PWCLO-Net forward frame-pairs/s on 2x8192-point KITTI-shaped pairs,
batch 32 per GPU, fp32, eval mode (BASELINE.json metric, configs[2]).

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

One process per GPU.  Started WITHOUT torchrun and with --gpus N > 1, the process only supervises: before
making any GPU call it starts N fresh rank processes of itself (dist_util.spawn_ranks) and exits with their
code; rank 0 of those prints the line.  A step = one forward over one batch of 32 synthetic frame pairs already
resident in HBM.  The forward has no cross-rank exchange (independent frame pairs), so the ranks
are replicas ("weak" scaling); the only collectives are the barrier around the timed region and
the MAX-reduce of its duration.  Rank 0 prints ONE JSON line.

Timed region: W warm-up steps, then exactly K steps between two fences (device sync + barrier +
device sync), MAX over ranks.  The region is repeated `--repeats` R times back to back (each repeat again
exactly K steps between fences); `value` / `ms_per_step` are the MEDIAN repeat, `repeats` lists min / median /
max (the GPU is busy for R*K steps, so an outside sampler sees the work).  By default a step is one hipGraph
replay and four batches are kept in flight (`--inflight 4`, each a forward over its own batch of 32; DESIGN.md
"Launch structure").

`roofline`: after the timed region the same step runs three more times eagerly with HIP events
around every launch of the library (events recorded on the launch stream); the object reports the
dominant MFMA kernel by time (one template instantiation, named as rocprofv3 names it) as
algorithmic FLOP / measured time against the fp32 MFMA peak, `traffic` from the committed PMC
passes (profiles/pmc_traffic.json), `mlp_family` the same figure over every MFMA-stack launch, and
`kernels` lists the other families (FPS, knn) with the bound that applies to them.  `cpu_baseline`: the CPU oracle
(bit-identical to the imported reference) on a few B=1 pairs on this box's host cores.
`variants` (N=1 only, after everything above): the same step with the opt-in PWCLO_BF16X3=1 stack layers, for
comparison; it is never the headline `value`.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
import pwclonet_pylidarslam_amd  # noqa: E402

import numpy as np  # noqa: E402
import torch  # noqa: E402

from pwclonet_pylidarslam_amd import _lib, dist_util, synthetic  # noqa: E402
from pwclonet_pylidarslam_amd.pwclonet import PWCLONet  # noqa: E402

# /opt/skills/guides/MI355X_MICROARCH.md, chip-level parameters
HBM_PEAK_GBS = 8000.0
MFMA_F32_PEAK_TFLOPS = 157.3


def make_batch(batch, npoints, seed, device):
    """`batch` KITTI-shaped pairs -> two (B,3,N) fp32 tensors on `device`.  8 distinct ray-cast
    scenes, the rest are jittered copies (ray casting 32 scenes on the host would dominate the
    bench's wall clock; the kernels' work does not depend on which scene a cloud comes from)."""
    scenes = min(batch, 8)
    pc1, pc2, _, _ = synthetic.kitti_like_pair(seed, npoints, scenes)
    rng = np.random.default_rng(seed + 1)
    reps = (batch + scenes - 1) // scenes
    out = []
    for pc in (pc1, pc2):
        x = np.concatenate([pc[:, :, :3] + (rng.normal(0, 2e-3, pc[:, :, :3].shape) if r else 0.0)
                            for r in range(reps)], axis=0)[:batch].astype(np.float32)
        out.append(torch.from_numpy(np.ascontiguousarray(x)).permute(0, 2, 1).contiguous().to(device))
    return out[0], out[1]


class LaunchProfiler:
    """Collects (launcher name, annotated work, start/end HIP events) for every C-ABI launch."""

    def __init__(self):
        self.rows = []

    def add(self, name, meta, start, end):
        self.rows.append((name, meta or {}, start, end))

    def summary(self):
        fam = {}
        for name, meta, s, e in self.rows:
            f = meta.get("family", "other")
            d = fam.setdefault(f, {"ms": 0.0, "launches": 0, "flops": 0.0, "bytes": 0.0, "units": 0.0,
                                   "iters": 0, "by_name": {}})
            ms = s.elapsed_time(e)
            d["ms"] += ms
            d["launches"] += 1
            for k in ("flops", "bytes", "units", "iters"):
                d[k] += meta.get(k, 0)
            bn = d["by_name"].setdefault(meta.get("kernel", name), {"ms": 0.0, "launches": 0, "flops": 0.0,
                                                                     "bytes": 0.0})
            bn["ms"] += ms
            bn["launches"] += 1
            bn["flops"] += meta.get("flops", 0)
            bn["bytes"] += meta.get("bytes", 0)
        return fam


def instrumented_pass(net, x1, x2, passes=3):
    """`passes` eager forwards of the same step with per-launch HIP events (graphs cannot be timed
    kernel by kernel from inside the process); returns the per-family summary of their average."""
    prof = LaunchProfiler()
    with torch.no_grad():
        net(x1, None, x2, None)           # warm (allocator, one-time attributes)
        torch.cuda.synchronize()
        _lib.profiler = prof
        try:
            for _ in range(passes):
                net(x1, None, x2, None)
        finally:
            _lib.profiler = None
    torch.cuda.synchronize()
    fam = prof.summary()
    for d in fam.values():                # totals -> per-step averages
        for k in ("ms", "flops", "bytes", "units"):
            d[k] /= passes
        d["launches"] //= passes
        d["iters"] //= passes
        for v in d["by_name"].values():
            for k in ("ms", "flops", "bytes"):
                v[k] /= passes
            v["launches"] //= passes
    return fam


def _pmc_file(name, kernel, field):
    """(value, stale, source) of `field` for `kernel` from a committed PMC artefact.  Hardware counters cannot be read
    from inside the process, so these come from rocprofv3 --pmc passes kept under profiles/; every such file carries
    the stamp of the source tree it was measured on (build.source_stamp(): csrc/, the ABI header, fused.py) and
    `stale` says whether that differs from the tree that is running now -- a stale figure is still printed, but marked
    (nothing used to tie these files to the code beside whose timings they are reported)."""
    from pwclonet_pylidarslam_amd.build import source_stamp
    try:
        with open(os.path.join(ROOT, "profiles", name)) as f:
            d = json.load(f)
    except (OSError, ValueError):
        return None, None, None
    src = d.get("_source", {})
    stale = src.get("csrc_sha16") != source_stamp()
    return d.get(kernel, {}).get(field), stale, src


def pmc_traffic(kernel):
    """HBM bytes per launch of `kernel` from the committed PMC passes (profiles/: FETCH_SIZE and
    WRITE_SIZE collected in separate rocprofv3 --pmc runs, corrected as MI355X_MICROARCH.md's HBM
    section prescribes)."""
    return _pmc_file("pmc_traffic.json", kernel, "hbm_bytes_per_launch")


def pmc_mfma_busy(kernel):
    """Share of SIMD time the matrix pipe was executing during `kernel`, from the committed SQ counter pass
    (profiles/pmc_mfma_busy.json, tools/pmc_mfma.py): includes the MFMA work spent on padded channels / pixels,
    which `achieved` (algorithmic FLOP) does not count."""
    return _pmc_file("pmc_mfma_busy.json", kernel, "mfma_busy")


def roofline_objects(fam):
    total_ms = sum(d["ms"] for d in fam.values()) or 1.0
    mlp = fam.get("mlp", {"ms": 0.0, "launches": 0, "flops": 0.0, "bytes": 0.0, "by_name": {}})
    # the dominant kernel: the MFMA-stack kernel (one template instantiation) with the most time
    name, dom = max(mlp["by_name"].items(), key=lambda kv: kv[1]["ms"], default=("none", None))
    dom = dom or {"ms": 0.0, "launches": 0, "flops": 0.0, "bytes": 0.0}
    n = max(dom["launches"], 1)
    tf = dom["flops"] / 1e12 / (dom["ms"] / 1e3) if dom["ms"] > 0 else 0.0
    fam_tf = mlp["flops"] / 1e12 / (mlp["ms"] / 1e3) if mlp["ms"] > 0 else 0.0
    traffic, traffic_stale, traffic_src = pmc_traffic(name)
    busy, busy_stale, _busy_src = pmc_mfma_busy(name)
    roof = {"kernel": name, "bound": "mfma", "achieved": tf, "peak": MFMA_F32_PEAK_TFLOPS, "unit": "TFLOP/s",
            "frac": tf / MFMA_F32_PEAK_TFLOPS, "traffic": traffic, "traffic_stale": traffic_stale,
            "mfma_busy_pmc": busy, "mfma_busy_stale": busy_stale, "pmc_source": traffic_src,
            "launches_per_step": dom["launches"], "avg_launch_us": 1e3 * dom["ms"] / n,
            "algorithmic_gflop_per_launch": dom["flops"] / 1e9 / n,
            "algorithmic_mb_per_launch": dom["bytes"] / 1e6 / n,
            "share_of_kernel_time": dom["ms"] / total_ms,
            "timing": "HIP events on the launch stream, average of 3 eager passes after the timed region "
                      "(isolated launches; in the pipelined timed region kernels of different batches share CUs)",
            "mlp_family": {"kernels": "every mlp_core MFMA stack kernel (sa/upconv/pointwise/cv_a1/cv_a2/cv_b/"
                                      "linear_jobs)", "launches_per_step": mlp["launches"],
                           "ms_per_step": mlp["ms"], "algorithmic_gflop_per_step": mlp["flops"] / 1e9,
                           "achieved_tflops": fam_tf, "frac": fam_tf / MFMA_F32_PEAK_TFLOPS,
                           "share_of_kernel_time": mlp["ms"] / total_ms}}
    kernels = {}
    if "fps" in fam:
        d = fam["fps"]
        kernels["furthest_point_sampling"] = {
            "bound": "latency (dependent arg-max chain, one workgroup per cloud)", "ms": d["ms"],
            "launches": d["launches"], "ns_per_iteration": 1e6 * d["ms"] / max(d["iters"], 1),
            "gpoint_visits_per_s": d["units"] / 1e9 / (d["ms"] / 1e3) if d["ms"] else 0.0,
            "hbm_gbs": d["bytes"] / 1e9 / (d["ms"] / 1e3) if d["ms"] else 0.0,
            "share_of_kernel_time": d["ms"] / total_ms}
    if "knn" in fam:
        d = fam["knn"]
        kernels["knn_point"] = {
            "bound": "valu (distance + select), HBM traffic negligible", "ms": d["ms"], "launches": d["launches"],
            "gdist_per_s": d["units"] / 1e9 / (d["ms"] / 1e3) if d["ms"] else 0.0,
            "hbm_gbs": d["bytes"] / 1e9 / (d["ms"] / 1e3) if d["ms"] else 0.0,
            "share_of_kernel_time": d["ms"] / total_ms}
    if "other" in fam:
        kernels["other"] = {"ms": fam["other"]["ms"], "launches": fam["other"]["launches"],
                            "share_of_kernel_time": fam["other"]["ms"] / total_ms}
    return roof, kernels


def cpu_baseline(net, npoints, timed=5, threads=0, seconds=12.0):
    """the CPU oracle (restatement of the reference path, pinned bit-identically to the
    imported reference in the build container) on the same KITTI-shaped pairs, batch 1 and batch 4, eval mode,
    fp32, 1 warm-up + at least `timed` (>= 3) forwards each and as many more as fit `seconds` / 2 of CPU work per
    batch size (a bounded sample of 10-30 s in total), MEDIAN reported.  All host cores are used: torch's CPU
    convolutions with its default intra-op threads, the C ops' FPS (over clouds) and knn (over queries) loops
    with OpenMP.  `value` is the better of the two batch sizes.  A baseline, not a target."""
    from oracle import model as omodel, ops as oops
    cores = oops.usable_host_cores()                 # cgroup quota, not the host's 256 hardware threads
    if threads:
        cores = threads
    torch.set_num_threads(cores)
    oops.set_num_threads(cores)
    sd = {k: v.detach().cpu() for k, v in net.state_dict().items()}
    x1, x2 = make_batch(4, npoints, 999, torch.device("cpu"))
    rows, t_all = {}, time.perf_counter()
    for bsz in (1, 4):
        omodel.pwclonet_forward(sd, x1[:bsz], x2[:bsz])  # warm-up
        ts, t_b = [], time.perf_counter()
        while len(ts) < max(3, timed) or (time.perf_counter() - t_b < seconds / 2 and len(ts) < 400):
            t0 = time.perf_counter()
            omodel.pwclonet_forward(sd, x1[:bsz], x2[:bsz])
            ts.append(time.perf_counter() - t0)
        ts.sort()
        rows["batch%d" % bsz] = {"median_s_per_forward": ts[len(ts) // 2], "min_s": ts[0], "max_s": ts[-1],
                                 "pairs_per_s": bsz / ts[len(ts) // 2], "timed_forwards": len(ts)}
    best = max(rows.values(), key=lambda r: r["pairs_per_s"])
    return {"value": best["pairs_per_s"], "unit": "frame-pairs/s", "cores": cores, "kind": "port",
            "os_cpu_count": os.cpu_count(), "torch_threads": torch.get_num_threads(),
            "c_ops_threads": oops.num_threads(), **rows,
            "sample": "oracle.model on 2x%d-pt pairs: batch 1 and batch 4, 1 warm-up + %d / %d timed forwards, median; "
                      "os.cpu_count() = %d, torch.get_num_threads() = %d, C ops (FPS over clouds, knn over queries) on "
                      "%d OpenMP threads; %.1f s of CPU work in total"
                      % (npoints, rows["batch1"]["timed_forwards"], rows["batch4"]["timed_forwards"], os.cpu_count(),
                         torch.get_num_threads(), oops.num_threads(),
                         time.perf_counter() - t_all)}


def bf16x3_variant(args, dev, x1, x2, pose_ref, streams, dtype="bf16x3"):
    """The same step with reduced-format stack layers, measured AFTER the headline region on the same inputs and
    weights and reported beside the headline number, never as it: "bf16x3" = fp32 operands split into three bf16
    terms on the bf16 matrix pipe, fp32 accumulate (fp32-accurate); "bf16" = the configs[4]
    dtype (operands and hoisted / per-pixel rows rounded to bf16) at THIS workload."""
    from pwclonet_pylidarslam_amd.graphed import PipelinedForward
    try:
        torch.manual_seed(1234)
        net = PWCLONet(dict(num_input_channels=3, sequence_len=2, device=str(dev), scalar_last=False,
                            log_mode=args.log_mode)).to(dev).eval()
        net.prepare_fused(dtype=dtype)
        pipe = PipelinedForward(net, depth=args.inflight, streams=streams)
        pipe.prepare(x1, x2)
        for _ in range(args.warmup):
            pipe(x1, x2)
        dist_util.fence(dev)
        t0 = time.perf_counter()
        for _ in range(args.steps):
            pose = pipe(x1, x2)[0]
        dist_util.fence(dev)
        dt = time.perf_counter() - t0
    finally:
        pass
    return {"value": args.batch * args.steps / dt, "unit": "frame-pairs/s", "ms_per_step": 1e3 * dt / args.steps,
            "max_abs_pose_diff_vs_headline": float((pose - pose_ref).abs().max()),
            "note": ("opt-in dtype=%s; not the headline path.  " % dtype) +
                    ("The pose difference is dominated by pairs whose level-1 neighbour list differs between the two paths "
                     "after the warp (DESIGN.md section 2)" if dtype == "bf16x3" else
                     "bf16 operands: poses agree to ~1e-2 of their scale (tests/test_gpu_fused.py)")}


def other_configs(args, budget_s=90.0):
    """The other single-GPU configurations of BASELINE.json measured by the SAME default run, after the headline region
    (only the default line is driver-visible): configs[1] = one 2x8192 pair (`--batch 1`) and configs[4] =
    raw 120k-row frames -> exact FPS to 8192 -> bf16 pyramid, batch 8 (`--config 5`, with its own roofline and
    cpu_baseline).  Each runs as a CHILD process of this one (started, never exec'd; this process sits idle meanwhile,
    so the GPU is the child's), bounded in time, and a failure is recorded, not raised: the headline line is printed
    either way."""
    import subprocess
    runs = {"1": ["--batch", "1", "--no-cpu-baseline", "--no-variants", "--no-configs", "--timed-seconds", "1.0"],
            "4": ["--config", "5", "--no-configs", "--repeats", "3", "--cpu-threads", str(args.cpu_threads)]}
    out, t_all = {}, time.perf_counter()
    for key, extra in runs.items():
        left = budget_s - (time.perf_counter() - t_all)
        if left < 20.0:
            out[key] = {"error": "skipped: the time budget for the extra configurations was used up"}
            continue
        t0 = time.perf_counter()
        try:
            r = subprocess.run([sys.executable, os.path.abspath(__file__)] + extra, capture_output=True, text=True,
                               timeout=left, cwd=ROOT)
            line = [l for l in r.stdout.splitlines() if l.startswith("{")]
            if r.returncode != 0 or not line:
                out[key] = {"error": "exit code %d: %s" % (r.returncode, r.stderr.strip()[-300:])}
                continue
            d = json.loads(line[-1])
            keep = ("metric", "value", "unit", "ms_per_step", "steps", "dtype", "config", "repeats", "stages_ms",
                    "roofline", "cpu_baseline")
            d = {k: d[k] for k in keep if k in d}
            if key == "1" and "roofline" in d:      # the batch-1 run's per-kernel breakdown: only the summary figures
                d["roofline"] = {k: d["roofline"][k] for k in ("kernel", "achieved", "frac", "avg_launch_us") if k in d["roofline"]}
            d["wall_s"] = time.perf_counter() - t0
            out[key] = d
        except subprocess.TimeoutExpired:
            out[key] = {"error": "timed out after %.0f s" % left}
        except (OSError, ValueError) as e:
            out[key] = {"error": repr(e)}
    return out


def step_roofline(fam, ms_per_step):
    """Step-level figures: every MFMA-stack launch's algorithmic FLOP against (a) the serial sum of their
    isolated launch times and (b) the pipelined step time the headline `value` is made of."""
    mlp = fam.get("mlp", {"flops": 0.0, "ms": 0.0})
    gflop = mlp["flops"] / 1e9
    fam_tf = gflop / mlp["ms"] if mlp["ms"] > 0 else 0.0            # GFLOP / ms = TFLOP/s
    step_tf = gflop / ms_per_step if ms_per_step > 0 else 0.0
    return {"algorithmic_gflop_per_step": gflop, "ms_per_step": ms_per_step,
            "achieved_tflops": step_tf, "frac": step_tf / MFMA_F32_PEAK_TFLOPS,
            "mlp_family_serial_ms": mlp["ms"], "mlp_family_tflops": fam_tf,
            "mlp_family_frac": fam_tf / MFMA_F32_PEAK_TFLOPS,
            "note": "frac = all MFMA-stack FLOP of a step / the WHOLE pipelined step time (FPS, knn and glue "
                    "included) / fp32 MFMA peak; mlp_family_frac = the same FLOP / serial sum of the stack "
                    "kernels' isolated launch times"}


BF16_MFMA_PEAK_TFLOPS = 2516.0     # MI355X_MICROARCH.md: dense bf16 MFMA peak


def raw_frames(seed, b, n, device):
    """KITTI-360-sized raw sensor frames (b, n, 4) = x, y, z, intensity in the sensor frame: lidar-like radial
    density, z from below the wheel axis (ground, filtered out) to above it."""
    g = torch.Generator().manual_seed(seed)
    xy = torch.randn(b, n, 2, generator=g) * 18.0
    z = torch.rand(b, n, 1, generator=g) * 5.0 - 2.0
    inten = torch.rand(b, n, 1, generator=g)
    return torch.cat((xy, z, inten), dim=2).contiguous().to(device)


def config5_cpu_baseline(sd, f1, f2, npoints, near, threads):
    """The oracle's version of one configs[4] step on a bounded sample (2 pairs): NumPy KITTI-360 filter
    (oracle/preprocess.py), C furthest point sampling of the ~95k survivors to `npoints` (OpenMP over the clouds),
    oracle.model forward in fp32."""
    from oracle import model as omodel, ops as oops, preprocess as opre
    cores = threads or oops.usable_host_cores()
    torch.set_num_threads(cores)
    oops.set_num_threads(cores)
    pairs = f1.shape[0]
    t0 = time.perf_counter()
    clouds = []
    for fr in list(f1) + list(f2):
        pts, keep = opre.kitti360_filter(fr.numpy(), near)
        clouds.append(torch.from_numpy(np.ascontiguousarray(pts[keep])))
    t_filter = time.perf_counter() - t0
    cap = max(c.shape[0] for c in clouds)
    packed = torch.zeros((len(clouds), cap, 3))
    for i, c in enumerate(clouds):
        packed[i, :c.shape[0]] = c
    idx = oops.furthest_point_sampling(packed, npoints)
    sampled = torch.gather(packed, 1, idx.long().unsqueeze(-1).expand(-1, -1, 3))
    t_fps = time.perf_counter() - t0 - t_filter
    x1 = sampled[:pairs].permute(0, 2, 1).contiguous()
    x2 = sampled[pairs:].permute(0, 2, 1).contiguous()
    omodel.pwclonet_forward(sd, x1, x2)
    dt = time.perf_counter() - t0
    return {"value": pairs / dt, "unit": "frame-pairs/s", "cores": cores, "kind": "port",
            "stages_s": {"filter": t_filter, "fps_to_%d" % npoints: t_fps, "pyramid_fp32": dt - t_filter - t_fps},
            "sample": "%d pairs of raw %d-row frames: NumPy KITTI-360 filter, C oracle FPS of the survivors (max %d) to %d "
                      "on %d OpenMP threads, oracle.model forward (fp32, torch CPU) -- %.1f s of CPU work"
                      % (pairs, f1.shape[1], cap, npoints, cores, dt)}


def run_config5(args):
    """BASELINE.json configs[4]: a batch of 8 raw KITTI-360-sized frame pairs (2 x ~120k rows of x,y,z,intensity,
    resident in HBM) -> on-device ground / range filter + compaction -> furthest point sampling of the ~95k survivors
    to 8192 (cooperative multi-workgroup sampler, exact) -> full pyramid with dtype="bf16" stack layers -> poses.
    A step = all of that for one batch; launches are eager (the cooperative launch cannot be graph-captured and the
    24 ms sampler dwarfs launch gaps).  One JSON line like the headline's."""
    from pwclonet_pylidarslam_amd import preprocess
    rank, local_rank, world = dist_util.env_world()
    dev = torch.device("cuda", local_rank)
    torch.cuda.set_device(dev)
    dist_util.init("nccl", dev)
    _lib.load()
    B, rows, npts, near = args.batch, args.rows, args.npoints, 35.0
    torch.manual_seed(1234)
    net = PWCLONet(dict(num_input_channels=3, sequence_len=2, device=str(dev), scalar_last=False,
                        log_mode="none")).to(dev).eval()
    net.prepare_fused(dtype=args.dtype)
    f1, f2 = raw_frames(1 + 10 * rank, B, rows, dev), raw_frames(2 + 10 * rank, B, rows, dev)
    frames = torch.cat((f1, f2), dim=0)

    def step():
        clouds, counts = preprocess.frames_to_clouds(frames, npts, dataset="kitti360", near_threshold=near)
        x1 = clouds[:B].transpose(1, 2).contiguous()
        x2 = clouds[B:].transpose(1, 2).contiguous()
        with torch.no_grad():
            pose, _ = net(x1, None, x2, None)
        return pose, counts

    # Batches in flight: the sampler of one batch is 16 clouds x 8 workgroups = 128 of the 256 CUs for ~23 ms, so TWO
    # batches (each on its own stream) fill the chip; never more than two -- a third cooperative launch could not become
    # resident while the first two run and its resident part would wait for its peers until the spin bound.
    depth = max(1, min(2, args.config5_inflight))
    from pwclonet_pylidarslam_amd.pointnet2_ops import _ext as _ext_mod
    # spatial order of the large-cloud sampler: the hand-written device kernels (default of the library: faster by 1.1 ms,
    # cells half the size, one batch in flight 20.6 instead of 21.3 ms) -- except with TWO sampler streams in flight, where the
    # same kernels measured 12.0 ms per batch against 10.2 ms with the torch-sort front end (profiles/r03/README.md: the slower,
    # chip-wide front end leaves the pyramids of the other batch a window; padding the device path by 1.1 ms recovers half of
    # the difference).  `--config5-order` overrides; the choice is reported in config.spatial_order.
    order = args.config5_order if args.config5_order != "auto" else ("torch" if depth > 1 else "device")
    _ext_mod.LARGE_CLOUD_ORDER = order
    if depth > 1:
        # two samplers side by side need the plain launch: the cooperative-launch API has one queue per device and would
        # run them one after the other (include/pwclo_ops.h: pwclo_fps_large_cloud_launch); at most 2 x 128 workgroups
        # of the sampler are ever in flight here, and a peer that is not resident in time ends in PWCLO_ECOOP_TIMEOUT
        _lib.load().pwclo_fps_large_cloud_launch(0)
    side = [torch.cuda.Stream(device=dev) for _ in range(depth)] if depth > 1 else None

    def front():                                        # filter, compaction, exact sampling: the sampler's stream
        clouds, counts = preprocess.frames_to_clouds(frames, npts, dataset="kitti360", near_threshold=near)
        return clouds[:B].transpose(1, 2).contiguous(), clouds[B:].transpose(1, 2).contiguous(), counts

    def back(x1, x2):                                   # the pyramid
        with torch.no_grad():
            pose, _ = net(x1, None, x2, None)
        return pose

    pyr = torch.cuda.Stream(device=dev) if depth > 1 and args.config5_split else None

    def run_steps(k):
        out = None
        if side is None:
            for _ in range(k):
                out = step()
            return out
        main = torch.cuda.current_stream(dev)
        for s_ in side + ([pyr] if pyr is not None else []):
            s_.wait_stream(main)
        for i in range(k):
            if pyr is None:
                with torch.cuda.stream(side[i % depth]):
                    out = step()
            else:
                # sampler streams carry only front ends: the next sampler starts the moment this one ends, the pyramid
                # of the finished batch runs on its own stream beside it
                with torch.cuda.stream(side[i % depth]):
                    x1, x2, counts = front()
                    ready = torch.cuda.Event()
                    ready.record()
                with torch.cuda.stream(pyr):
                    pyr.wait_event(ready)
                    x1.record_stream(pyr)
                    x2.record_stream(pyr)
                    out = (back(x1, x2), counts)
        for s_ in side + ([pyr] if pyr is not None else []):
            main.wait_stream(s_)
        return out

    run_steps(max(1, args.warmup))
    times = []
    for _ in range(max(3, args.repeats)):
        dist_util.fence(dev)
        t0 = time.perf_counter()
        pose, counts = run_steps(args.steps)
        dist_util.fence(dev)
        times.append(dist_util.max_over_ranks(time.perf_counter() - t0, dev))
    _lib.synchronize(dev)                       # raises if the cooperative sampler reported a timeout
    assert torch.isfinite(pose).all()
    ordered = sorted(times)
    dt = ordered[len(ordered) // 2]
    if rank != 0:
        dist_util.finish()
        return
    # per-stage times (HIP events on the launch stream) and the per-launch profile of one more step
    def ev_time(fn, reps=2):
        fn(); torch.cuda.synchronize(dev)
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(reps):
            out = fn()
        b.record(); torch.cuda.synchronize(dev)
        return a.elapsed_time(b) / reps, out
    t_filter, (xyz, keep) = ev_time(lambda: preprocess.kitti360_filter(frames, near))
    t_compact, (packed, cnts) = ev_time(lambda: preprocess.compact(xyz, keep))
    from pwclonet_pylidarslam_amd.pointnet2_ops import _ext
    t_fps, idx = ev_time(lambda: _ext.furthest_point_sampling(packed, npts))
    clouds = torch.gather(packed, 1, idx.long().unsqueeze(-1).expand(-1, -1, 3))
    x1, x2 = clouds[:B].transpose(1, 2).contiguous(), clouds[B:].transpose(1, 2).contiguous()
    fam = instrumented_pass(net, x1, x2, passes=2)
    mlp = fam.get("mlp", {"ms": 0.0, "flops": 0.0})
    t_pyr = sum(d["ms"] for d in fam.values())
    n_surv = int(cnts.max())
    fps_bytes = 2 * B * (12.0 * n_surv + 16.0 * npts)
    out = {
        "metric": "PWCLO-Net forward frame-pairs/sec, raw ~120k-pt KITTI-360 frames -> 8192-pt FPS -> pyramid, batch 8",
        "value": world * B * args.steps / dt, "unit": "frame-pairs/s", "n_gpus": world, "steps": args.steps,
        "warmup": max(1, args.warmup), "ms_per_step": 1e3 * dt / args.steps, "higher_is_better": True,
        "scaling": "weak", "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
        "repeats": {"n": len(times), "reported": "median", "ms_per_step_all": [1e3 * t / args.steps for t in times]},
        "config": {"workload": "BASELINE.json configs[4]: %d pairs x 2 x %d raw rows (x,y,z,i) per GPU, KITTI-360 filter "
                               "(survivors up to %d per frame), exact furthest point sampling to %d, full 4-level pyramid; "
                               "stack layers in %s (fp32 accumulate), coordinates / distances / indices fp32"
                               % (B, rows, n_surv, npts, args.dtype),
                   "global_batch": world * B, "npoints": npts, "parallelism": "replicas x%d" % world,
                   "launch": "eager; large-cloud sampler: %s" % ("plain launch, two batches in flight on two streams"
                                                                  if depth > 1 else "cooperative launch, one batch in flight"),
                   "batches_in_flight": depth, "pyramids_on_their_own_stream": bool(pyr is not None),
                   "spatial_order": order + (" (torch argsort front end: see profiles/r03/README.md)" if order == "torch"
                                             else " (csrc/sampling.hip fps_spatial_order kernels)")},
        "stages_ms": {"kitti360_filter": t_filter, "compaction": t_compact, "fps_%d_to_%d" % (n_surv, npts): t_fps,
                      "pyramid_kernels_%s" % args.dtype: t_pyr},
        "roofline": {"kernel": "fps_coop_kernel<16>", "bound": "hbm",
                     "achieved": fps_bytes / 1e9 / (t_fps / 1e3), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": fps_bytes / 1e9 / (t_fps / 1e3) / HBM_PEAK_GBS, "traffic": None,
                     "share_of_step": t_fps / (depth * 1e3 * dt / args.steps),
                     "launch_duration_ms": t_fps, "batches_in_flight": depth,
                     "ns_per_iteration": 1e6 * t_fps / (npts - 1),
                     "note": "the dominant kernel is a chain of %d dependent arg-max iterations shared by 8 workgroups per "
                             "cloud: latency-bound (cross-workgroup exchange + distance update), its %d MB of algorithmic "
                             "HBM bytes are irrelevant; reported against the HBM roof because the contract asks for one.  "
                             "One launch (16 clouds) occupies 128 of the 256 CUs, so with two batches in flight two "
                             "launches run side by side: share_of_step = launch duration / (batches in flight x step time)"
                             % (npts - 1, int(fps_bytes / 1e6)),
                     "mlp_family": {"dtype": args.dtype, "ms_per_step": mlp["ms"],
                                    "algorithmic_gflop_per_step": mlp["flops"] / 1e9,
                                    "achieved_tflops": mlp["flops"] / 1e9 / mlp["ms"] if mlp["ms"] else 0.0,
                                    "peak_tflops": BF16_MFMA_PEAK_TFLOPS if args.dtype == "bf16" else MFMA_F32_PEAK_TFLOPS,
                                    "note": "batch 8: every stack launch is launch / fill bound (<= 16 clouds)"}}}
    if not args.no_cpu_baseline and world == 1:
        sd = {k: v.detach().cpu() for k, v in net.state_dict().items()}
        out["cpu_baseline"] = config5_cpu_baseline(sd, f1[:2].cpu(), f2[:2].cpu(), npts, near, args.cpu_threads)
    print(json.dumps(out), flush=True)
    dist_util.finish()


def dry_run(args):
    """--dry-run: the launch / rendezvous / aggregation plumbing with the GPU work replaced by a host sleep
    (gloo, CPU).  For the CPU test of the self-spawn path; prints a line marked `dry_run`, never a result."""
    rank, _local, world = dist_util.env_world()
    dist_util.init("gloo")
    dist_util.fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        time.sleep(0.001 * (1 + rank))
    dist_util.fence()
    dt = dist_util.max_over_ranks(time.perf_counter() - t0)
    pairs = dist_util.sum_over_ranks(args.batch * args.steps)
    if rank == 0:
        print(json.dumps({"dry_run": True, "metric": "plumbing only (no GPU work)", "n_gpus": world,
                          "steps": args.steps, "global_pairs": pairs, "ms_per_step": 1e3 * dt / args.steps}),
              flush=True)
    dist_util.finish()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=None, help="default 40 (--config 5: 5)")
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--repeats", type=int, default=0,
                    help="how many times the K-step timed region is run back to back (median reported); 0 = as many "
                         "as fit in about --timed-seconds (at least 5, at most 200), decided from the first repeat")
    ap.add_argument("--timed-seconds", type=float, default=3.0)
    ap.add_argument("--batch", type=int, default=None, help="frame pairs per GPU per step: default 32 (--config 5: 8)")
    ap.add_argument("--npoints", type=int, default=8192)
    ap.add_argument("--log-mode", default="host", choices=["host", "device", "none"],
                    help="host = the reference's log_dict on the host (lazy: built when read)")
    ap.add_argument("--launch", default="graph", choices=["graph", "eager"],
                    help="graph = replay one captured hipGraph per step (default); eager = Python launches")
    ap.add_argument("--config5-inflight", type=int, default=2,
                    help="--config 5: batches in flight, 1 or 2 (each sampler launch occupies half of the CUs)")
    ap.add_argument("--config5-order", default="auto", choices=["auto", "device", "torch"],
                    help="--config 5: how the large-cloud sampler's spatial order is built (auto: device kernels with one batch "
                         "in flight, torch sorts with two -- measured faster there)")
    ap.add_argument("--config5-split", type=int, default=1,
                    help="--config 5 with two batches in flight: 1 = pyramids on a third stream (sampler streams run only "
                         "front ends)")
    ap.add_argument("--inflight", type=int, default=4,
                    help="batches in flight (graph launch only), each on its own stream: one batch's FPS chain "
                         "overlaps the others' neighbour-search/MLP kernels; 1 = strictly serial steps")
    ap.add_argument("--pipeline", default="whole", choices=["staged", "whole"],
                    help="with --inflight > 1: staged = sampling chains of successive batches back to back on "
                         "one stream, the rest on two others (graphed.StagedPipeline, --inflight = slots); "
                         "whole = one whole-forward graph per slot (graphed.PipelinedForward)")
    ap.add_argument("--unfused", action="store_true",
                    help="reference-shaped module graph on the HIP ops (torch conv/BN) instead of the fused kernels")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--cpu-forwards", type=int, default=5, help="timed CPU forwards per batch size (>= 3), at least")
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="CPU-baseline sample: seconds of CPU work in total")
    ap.add_argument("--cpu-threads", type=int, default=0,
                    help="host threads of the CPU baseline (0 = the cgroup CPU quota / affinity of this process)")
    ap.add_argument("--no-configs", action="store_true",
                    help="skip the extra measurement of BASELINE configs[1] (--batch 1) and configs[4] (--config 5) that the "
                         "default single-GPU run appends under \"configs\"")
    ap.add_argument("--no-variants", action="store_true",
                    help="skip the extra (untimed-region) measurement of the opt-in bf16x3 split path")
    ap.add_argument("--config", type=int, default=3, choices=[3, 5],
                    help="3 = the headline workload (BASELINE.json configs[2]: B=32 2x8192 fp32); 5 = configs[4]: raw "
                         "~120k-row frames -> FPS to 8192 -> pyramid in bf16, batch 8")
    ap.add_argument("--rows", type=int, default=120000, help="--config 5: raw rows per frame")
    ap.add_argument("--dtype", default=None, choices=["f32", "bf16", "bf16x3"],
                    help="stack-layer format of the fused kernels (default: f32 for the headline, bf16 for --config 5)")
    ap.add_argument("--dry-run", action="store_true",
                    help="plumbing only: rendezvous + fences + aggregation over gloo with a host sleep as the step")
    args = ap.parse_args()
    # explicit, before the first HIP call of the process: 4 batches in flight need more than the default 4 hardware
    # queues (importing this file -- tests and tools do, for make_batch -- configures nothing)
    pwclonet_pylidarslam_amd.configure_hw_queues(8)

    if args.steps is None:
        args.steps = 16 if args.config == 5 else 40
    if args.batch is None:
        args.batch = 8 if args.config == 5 else 32
    if args.gpus > 1 and not dist_util.launched_by_torchrun():
        # The driver's `python bench.py --gpus N`: this process has made no GPU call and makes none; it starts
        # N rank processes (one per GPU) and passes their exit code on.
        if not args.dry_run and torch.cuda.device_count() < args.gpus:   # device_count() does not initialise HIP
            sys.stderr.write("bench.py: --gpus %d but only %d GPU(s) visible\n" % (args.gpus, torch.cuda.device_count()))
            sys.exit(2)
        sys.exit(dist_util.spawn_ranks(os.path.abspath(__file__), sys.argv[1:], args.gpus))

    rank, local_rank, world = dist_util.env_world()
    if world != args.gpus:
        sys.stderr.write("bench.py: WORLD_SIZE=%d but --gpus %d (launch with --nproc-per-node == --gpus)\n"
                         % (world, args.gpus))
        sys.exit(2)
    if args.dry_run:
        return dry_run(args)
    if args.config == 5:
        args.dtype = args.dtype or "bf16"
        if args.repeats <= 0:
            args.repeats = 3
        return run_config5(args)
    dev = torch.device("cuda", local_rank)
    torch.cuda.set_device(dev)
    dist_util.init("nccl", dev)

    _lib.load()
    torch.manual_seed(1234)
    net = PWCLONet(dict(num_input_channels=3, sequence_len=2, device=str(dev), scalar_last=False,
                        log_mode=args.log_mode, fused="off" if args.unfused else "auto")).to(dev).eval()
    if not args.unfused:
        net.prepare_fused(dtype=args.dtype)
    x1, x2 = make_batch(args.batch, args.npoints, 1000 + rank, dev)

    pipe = None
    if args.launch == "graph" and args.inflight > 1:
        from pwclonet_pylidarslam_amd.graphed import PipelinedForward, StagedPipeline
        if args.pipeline == "staged" and not args.unfused:
            pipe = StagedPipeline(net, slots=args.inflight)
        else:
            pipe = PipelinedForward(net, depth=args.inflight)

        def step():
            return pipe(x1, x2)[0]
    elif args.launch == "graph":
        from pwclonet_pylidarslam_amd.graphed import GraphedForward
        graphed = GraphedForward(net)

        def step():
            return graphed(x1, x2)
    else:
        def step():
            with torch.no_grad():
                pose, _ = net(x1, None, x2, None)
            return pose

    if pipe is not None:
        pipe.prepare(x1, x2)          # graph capture of every slot is set-up, not a warm-up step
    for _ in range(args.warmup):
        step()

    times, repeats = [], (args.repeats if args.repeats > 0 else 5)
    while len(times) < repeats:
        dist_util.fence(dev)
        t0 = time.perf_counter()
        for _ in range(args.steps):
            pose = step()
        dist_util.fence(dev)
        times.append(dist_util.max_over_ranks(time.perf_counter() - t0, dev))
        if args.repeats <= 0 and len(times) == 1:     # same decision on every rank: times[0] is the MAX over ranks
            repeats = int(min(200, max(5, round(args.timed_seconds / max(times[0], 1e-6)))))
    assert torch.isfinite(pose).all()
    _lib.synchronize(dev)             # raises if any kernel reported a device-side failure
    ordered = sorted(times)
    dt = ordered[len(ordered) // 2]   # median repeat

    if rank == 0:
        out = {
            "metric": "PWCLO-Net forward frame-pairs/sec, 2x%d-pt KITTI pair, batch %d" % (args.npoints, args.batch),
            "value": world * args.batch * args.steps / dt, "unit": "frame-pairs/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1e3 * dt / args.steps, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": args.dtype or "f32", "data": "synthetic",
            "repeats": {"n": len(times), "reported": "median",
                        "ms_per_step_min": 1e3 * ordered[0] / args.steps,
                        "ms_per_step_median": 1e3 * dt / args.steps,
                        "ms_per_step_max": 1e3 * ordered[-1] / args.steps,
                        "ms_per_step_all": [1e3 * t / args.steps for t in times]},
            "config": {"workload": "BASELINE.json configs[2]: synthetic KITTI-shaped 2x%d-pt pairs, "
                                   "batch %d per GPU, full 4-level pyramid, eval mode, fp32"
                                   % (args.npoints, args.batch),
                       "global_batch": world * args.batch, "npoints": args.npoints,
                       "parallelism": "replicas x%d (no forward collective)" % world,
                       "launch": args.launch, "batches_in_flight": args.inflight if pipe else 1,
                       "pipeline": (args.pipeline if pipe else "none"),
                       "hw_queues": pwclonet_pylidarslam_amd.hw_queues(),
                       "kernels": "module graph + torch conv/BN" if args.unfused
                       else "fused gather+MFMA-MLP kernels (BN folded)",
                       "log_dict": args.log_mode + " (lazy)"},
        }
        if not args.no_roofline and not args.unfused:
            fam = instrumented_pass(net, x1, x2)
            roof, kernels = roofline_objects(fam)
            roof["step"] = step_roofline(fam, out["ms_per_step"])
            out["roofline"] = roof
            out["kernels"] = kernels
        if not args.no_variants and world == 1 and pipe is not None and not args.unfused \
                and args.pipeline == "whole" and (args.dtype or "f32") == "f32" \
                and os.environ.get("PWCLO_BF16X3", "0") == "0":
            ref_pose = pose.clone()
            out["variants"] = {"bf16x3": bf16x3_variant(args, dev, x1, x2, ref_pose, pipe.streams, "bf16x3"),
                               "bf16": bf16x3_variant(args, dev, x1, x2, ref_pose, pipe.streams, "bf16")}
        if not args.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline(net, args.npoints, args.cpu_forwards, args.cpu_threads, args.cpu_seconds)
        default_workload = (args.batch == 32 and args.npoints == 8192 and not args.unfused and args.launch == "graph"
                            and (args.dtype or "f32") == "f32")
        if not args.no_configs and world == 1 and default_workload:
            torch.cuda.synchronize(dev)
            out["configs"] = other_configs(args)
        print(json.dumps(out), flush=True)
    dist_util.finish()


if __name__ == "__main__":
    main()
