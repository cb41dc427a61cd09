#!/usr/bin/env python3
"""Training-step throughput of PWCLO-Net on the HIP ops (BASELINE.json configs[3]).

    python tools/train_step.py [--batch 8] [--steps 10]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 tools/train_step.py --gpus N

One process per GPU; the reference-shaped module graph in TRAIN mode (BatchNorm batch statistics,
dropout; torch conv/BN autograd + the HIP gather/group forward and backward kernels), the
reference's supervised loss (pwclonet_pylidarslam_amd.loss), Adam, and -- for N > 1 --
DistributedDataParallel over RCCL: one 3.1 MB gradient all-reduce per step, BN buffers not
broadcast.  Not the headline benchmark (bench.py measures forward pairs/s); prints one JSON line.
"""
import argparse, json, os, sys, time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import pwclonet_pylidarslam_amd  # noqa: E402,F401
pwclonet_pylidarslam_amd.configure_hw_queues(8)
import torch  # noqa: E402

import bench  # noqa: E402
from pwclonet_pylidarslam_amd import dist_util  # noqa: E402
from pwclonet_pylidarslam_amd.loss import PWCLONetLossModule  # noqa: E402
from pwclonet_pylidarslam_amd.pwclonet import PWCLONet  # noqa: E402
from pwclonet_pylidarslam_amd.training import PWCLONetWithLoss, TrainStep, ddp_wrap, gradient_bucket_values  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--batch", type=int, default=8, help="frame pairs per GPU per step")
    ap.add_argument("--npoints", type=int, default=8192)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--fused-adam", action="store_true", help="torch.optim.Adam(fused=True): one multi-tensor kernel")
    ap.add_argument("--graph", action="store_true",
                    help="capture forward + loss + backward + Adam into one hipGraph (single GPU only)")
    ap.add_argument("--sample-ahead", action="store_true",
                    help="draw the next batch's furthest-point samples on a second stream while this batch's step runs "
                         "(training.TrainStep(sample_ahead=True); the synthetic batch is the same every step)")
    a = ap.parse_args()
    if a.gpus > 1 and not dist_util.launched_by_torchrun():      # supervise N fresh ranks; no GPU call made here
        sys.exit(dist_util.spawn_ranks(os.path.abspath(__file__), sys.argv[1:], a.gpus))
    rank, local_rank, world = dist_util.env_world()
    assert world == a.gpus, "WORLD_SIZE=%d but --gpus %d" % (world, a.gpus)
    dev = torch.device("cuda", local_rank)
    torch.cuda.set_device(dev)
    dist_util.init("nccl", dev)
    torch.manual_seed(7)                                   # same initial weights on every rank
    net = PWCLONet(dict(num_input_channels=3, sequence_len=2, device=str(dev), scalar_last=False,
                        log_mode="none")).to(dev).train()
    loss_mod = PWCLONetLossModule(dict(with_exp_weights=True, init_weights=[0.0, -2.5], loss_option="l2_norm",
                                       nb_levels=4, scalar_last=False)).to(dev)
    # network + loss in ONE module: the all-reduce carries the 775 068 network gradients and the loss module's two
    # learnable weights -- with the network alone under DDP the replicas' loss weights drift
    unit = PWCLONetWithLoss(net, loss_mod)
    model = ddp_wrap(unit, dev) if world > 1 else unit
    opt = torch.optim.Adam(unit.parameters(), lr=1e-4, capturable=a.graph, fused=True if a.fused_adam else None)
    x1, x2 = bench.make_batch(a.batch, a.npoints, 2000 + rank, dev)
    g = torch.Generator().manual_seed(3 + rank)
    gt = torch.randn(a.batch, 7, generator=g) * 0.1
    gt[:, 3:] = torch.nn.functional.normalize(gt[:, 3:] + torch.tensor([1.0, 0, 0, 0]), dim=1)
    gt = gt.to(dev)

    if a.graph:
        assert world == 1, "--graph is the single-GPU variant (DDP's bucketed all-reduce is not captured here)"
    step = TrainStep(model, opt, x1, x2, gt, graph=a.graph, sample_ahead=a.sample_ahead).step

    losses = [step().item() for _ in range(a.warmup)]
    dist_util.fence(dev)
    t0 = time.perf_counter()
    for _ in range(a.steps):
        loss = step()
    dist_util.fence(dev)
    dt = dist_util.max_over_ranks(time.perf_counter() - t0, dev)
    losses.append(loss.item())
    if rank == 0:
        print(json.dumps({"metric": "PWCLO-Net training frame-pairs/sec (fwd+bwd+Adam), 2x%d-pt pairs" % a.npoints,
                          "value": world * a.batch * a.steps / dt, "unit": "frame-pairs/s", "n_gpus": world,
                          "ms_per_step": 1e3 * dt / a.steps, "batch_per_gpu": a.batch, "dtype": "f32",
                          "launch": ("one hipGraph per step" if a.graph else "eager (module graph, torch autograd)")
                          + (", next batch's sampling chain on a second stream" if a.sample_ahead else ""), "loss_first_last": [losses[0], losses[-1]],
                          "collective": ("DDP all-reduce of %d fp32 gradient values (network + loss weights), one bucket"
                                         % gradient_bucket_values(unit)) if world > 1 else "none"}), flush=True)
    dist_util.finish()


if __name__ == "__main__":
    main()
