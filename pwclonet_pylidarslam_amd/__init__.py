"""MI355X-native PWCLO-Net point-cloud operator path."""
import os as _os

# The HIP runtime multiplexes streams onto GPU_MAX_HW_QUEUES hardware queues (default 4) round-robin
# and work of two streams that land on one queue is serialised.  The throughput pipelines
# (graphed.PipelinedForward / StagedPipeline) keep 3-4 forwards in flight on their own streams
# beside torch's capture/side streams; with 4 queues two of them collide (measured on MI355X:
# 7.4k pairs/s with 3 in flight vs 10.3k with 8 queues).  Read by the runtime when it initialises,
# i.e. this must run before the first HIP call of the process; an explicit setting wins.
_os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
