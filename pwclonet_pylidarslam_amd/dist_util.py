"""Multi-GPU plumbing of the benchmark: one process per GPU, replicas, no data-path collective.

PWCLO-Net's forward treats every frame pair independently (every kernel's outer dimension is the
batch), so N GPUs run N replicas over disjoint shards of the global batch; the
only communication is the barrier around the timed region and a MAX-reduce of its duration.
Backend "nccl" is RCCL on ROCm; "gloo" is used by the CPU tests.
"""
import os

import torch
import torch.distributed as dist


def env_world():
    """(rank, local_rank, world_size) from the torch.distributed.run environment (1-process default)."""
    return (int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")),
            int(os.environ.get("WORLD_SIZE", "1")))


def launched_by_torchrun():
    """True when the process already is one rank of a torch.distributed.run (or self-spawned) job."""
    return "RANK" in os.environ and "WORLD_SIZE" in os.environ


def spawn_ranks(script, argv, nprocs, env_extra=None, timeout=None):
    """``python script argv`` with --gpus N but no torchrun around it: start N fresh rank processes (one per GPU)
    with the torch.distributed.run environment and wait for them.  MUST be called before the calling process
    makes any GPU call: the parent only supervises -- it never touches the device and is never re-executed
    (replacing a process that has initialised the GPU takes the machine down on this pool).  The children
    inherit stdout/stderr, so rank 0's JSON line is the job's.  Returns the worst exit code."""
    import socket
    import subprocess
    import sys
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = []
    for rank in range(nprocs):
        env = dict(os.environ, RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(nprocs),
                   LOCAL_WORLD_SIZE=str(nprocs), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        env.update(env_extra or {})
        procs.append(subprocess.Popen([sys.executable, script] + list(argv), env=env))
    import time
    worst, t0 = 0, time.monotonic()
    try:
        while any(p.poll() is None for p in procs):
            failed = [p.returncode for p in procs if p.poll() not in (None, 0)]
            if failed or (timeout is not None and time.monotonic() - t0 > timeout):
                worst = failed[0] if failed else 124      # a dead rank leaves its peers in a barrier: stop them
                break
            time.sleep(0.05)
        for p in procs:
            if p.poll() is not None:
                worst = worst or p.returncode
    finally:
        for p in procs:           # end exactly the PIDs started here, never by pattern
            if p.poll() is None:
                p.terminate()
        for p in procs:
            try:
                p.wait(timeout=30)
            except Exception:
                p.kill()
    return worst


def init(backend, device=None):
    rank, _, world = env_world()
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29531")
        kw = {"device_id": device} if (backend == "nccl" and device is not None) else {}
        dist.init_process_group(backend, rank=rank, world_size=world, **kw)
    return rank, world


def shard(global_batch, rank, world):
    """[start, stop) of this rank's frame pairs; the first (global_batch % world) ranks get one more."""
    base, rem = divmod(global_batch, world)
    start = rank * base + min(rank, rem)
    return start, start + base + (1 if rank < rem else 0)


def fence(device=None):
    """device sync + barrier + device sync: brackets the timed region on every rank."""
    if device is not None and device.type == "cuda":
        torch.cuda.synchronize(device)
    if dist.is_available() and dist.is_initialized():
        dist.barrier()
    if device is not None and device.type == "cuda":
        torch.cuda.synchronize(device)


def max_over_ranks(value, device=None):
    """MAX all-reduce of a python float (the slowest rank defines the step time)."""
    if not (dist.is_available() and dist.is_initialized()):
        return float(value)
    dev = device if (device is not None and dist.get_backend() == "nccl") else torch.device("cpu")
    t = torch.tensor([value], dtype=torch.float64, device=dev)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def sum_over_ranks(value, device=None):
    if not (dist.is_available() and dist.is_initialized()):
        return float(value)
    dev = device if (device is not None and dist.get_backend() == "nccl") else torch.device("cpu")
    t = torch.tensor([value], dtype=torch.float64, device=dev)
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return float(t.item())


def finish():
    if dist.is_available() and dist.is_initialized():
        dist.destroy_process_group()
