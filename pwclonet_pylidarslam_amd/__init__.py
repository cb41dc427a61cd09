"""MI355X-native PWCLO-Net point-cloud operator path.

Importing the package has NO side effects on the host process (round 1 set ``GPU_MAX_HW_QUEUES`` here).
The throughput pipelines (``graphed.PipelinedForward`` / ``StagedPipeline``) keep 3-4 forwards in flight on
their own HIP streams; the HIP runtime multiplexes streams onto ``GPU_MAX_HW_QUEUES`` hardware queues
(default 4) round-robin and serialises streams that share a queue (measured on MI355X: 7.4k pairs/s with
4 queues against 12.8k with 8).  A program that wants that pipelining calls ``configure_hw_queues()``
EXPLICITLY, before its first HIP call -- the runtime reads the variable once, when it initialises.
``bench.py`` does, and reports the value it ran with.
"""
import os as _os


def configure_hw_queues(n=8):
    """Ask the HIP runtime for ``n`` hardware queues (``GPU_MAX_HW_QUEUES``) unless the environment already
    says otherwise.  Must run before the process initialises HIP: raises ``RuntimeError`` if torch has already
    initialised the device (the setting would silently not apply).  Returns the value in effect."""
    if "GPU_MAX_HW_QUEUES" in _os.environ:
        return int(_os.environ["GPU_MAX_HW_QUEUES"])
    import torch
    if torch.cuda.is_initialized():
        raise RuntimeError("configure_hw_queues() must be called before the first HIP call of the process "
                           "(GPU_MAX_HW_QUEUES is read when the runtime initialises)")
    _os.environ["GPU_MAX_HW_QUEUES"] = str(int(n))
    return int(n)


def hw_queues():
    """The hardware-queue count this process asked the runtime for (None = runtime default, 4)."""
    v = _os.environ.get("GPU_MAX_HW_QUEUES")
    return int(v) if v is not None else None
