"""HIP-graph replay of an eval-mode PWCLO-Net forward.

One PWCLO-Net forward is several hundred short kernels; launched eagerly from Python the GPU
idles between them (measured: 28 ms of kernels in a 100 ms step).  ``GraphedForward`` captures
one forward on fixed-shape buffers into a hipGraph (``torch.cuda.CUDAGraph``; the C-ABI launchers
allocate nothing and never synchronise, so they are capturable) and replays it per batch.
"""
import torch


class GraphedForward:
    """``GraphedForward(net)(xyz_f1, xyz_f2) -> pose_params (B,4,7)`` for a ``PWCLONet`` in eval
    mode.  Inputs (B,3,N) fp32 on the net's device; a new graph is captured per input shape.
    The returned tensor is a static buffer that the next call overwrites -- clone to keep it."""

    def __init__(self, net, warmup=2, branch=True):
        assert not net.training, "graph capture is for eval mode (fixed control flow, no dropout)"
        self.net = net
        self.warmup = warmup
        self.branch = branch      # fork/join the fused forward's independent branches inside the graph
        self._graphs = {}
        self._saved_log_mode = None

    def _capture(self, xyz_f1, xyz_f2):
        net = self.net
        log_mode = net.log_mode
        if log_mode == "host":          # a D2H copy cannot be captured; keep the values on device
            net.log_mode = "device"
        try:
            s1, s2 = xyz_f1.clone(), xyz_f2.clone()
            side = torch.cuda.Stream(device=s1.device)
            side.wait_stream(torch.cuda.current_stream(s1.device))
            with torch.cuda.stream(side), torch.no_grad():
                for _ in range(self.warmup):   # allocator warm-up + one-time kernel attributes
                    net(s1, None, s2, None)
            torch.cuda.current_stream(s1.device).wait_stream(side)
            torch.cuda.synchronize(s1.device)
            fused = getattr(net, "_fused", None)
            saved_branch = fused.branch if fused is not None else None
            if fused is not None:
                fused.branch = fused.branch and self.branch
            graph = torch.cuda.CUDAGraph()
            try:
                with torch.cuda.graph(graph), torch.no_grad():
                    pose, log = net(s1, None, s2, None)
            finally:
                if fused is not None:
                    fused.branch = saved_branch
        finally:
            net.log_mode = log_mode
        return graph, s1, s2, pose, log

    def __call__(self, xyz_f1, xyz_f2):
        key = (tuple(xyz_f1.shape), xyz_f1.device)
        if key not in self._graphs:
            self._graphs[key] = self._capture(xyz_f1, xyz_f2)
        graph, s1, s2, pose, log = self._graphs[key]
        s1.copy_(xyz_f1)
        s2.copy_(xyz_f2)
        graph.replay()
        self.last_log_dict = self._replay_log(log, xyz_f1.device)
        return pose

    def _replay_log(self, log, device):
        """log_dict of THIS replay: a fresh lazy view of the graph's static buffers (nothing cached from an
        earlier batch) that waits for the replay before its first read and honours the net's log_mode
        (``host`` -> host tensors, like the reference's log_dict)."""
        if not hasattr(log, "fresh"):
            return log
        ev = torch.cuda.Event()
        ev.record(torch.cuda.current_stream(device))
        return log.fresh(to_host=self.net.log_mode == "host", ready=ev.synchronize)


class PipelinedForward:
    """Keeps `depth` forwards in flight: one captured graph + stream per slot, used round-robin.

    Within one forward the furthest-point-sampling chain is a long dependent sequence that
    occupies one CU per cloud (64 of 256 CUs at batch 32) while the rest of the chip waits; with
    two batches in flight one batch's FPS runs under the other's neighbour-search / MLP kernels.
    Throughput tool: each call returns the (static) output tensor of the slot it used, valid once
    that slot's stream has been synchronised (``wait(slot)`` / ``wait_all()``)."""

    def __init__(self, net, depth=2, streams=None):
        # single-branch graphs: two multi-branch graphs in flight serialise on this runtime
        # (measured: 5.35 ms/step with branches vs 4.0 ms without, 2 in flight)
        import os
        branch = os.environ.get("PWCLO_PIPE_BRANCH", "0") != "0"
        self.slots = [GraphedForward(net, branch=branch) for _ in range(depth)]
        # `streams`: reuse another pipeline's streams -- the runtime binds every new stream to the next
        # hardware queue round-robin, so a second set of streams in one process can collide with itself
        self.streams = list(streams) if streams is not None else None
        assert self.streams is None or len(self.streams) == depth
        self.events = [None] * depth
        self._next = 0

    def __call__(self, xyz_f1, xyz_f2):
        if self.streams is None:
            self.streams = [torch.cuda.Stream(device=xyz_f1.device) for _ in self.slots]
        i = self._next
        self._next = (i + 1) % len(self.slots)
        st = self.streams[i]
        st.wait_stream(torch.cuda.current_stream(xyz_f1.device))   # inputs produced on the caller's stream
        with torch.cuda.stream(st):
            out = self.slots[i](xyz_f1, xyz_f2)
            ev = torch.cuda.Event()
            ev.record(st)
        self.events[i] = ev
        return out, i

    def prepare(self, xyz_f1, xyz_f2):
        """Capture every slot's graph now (otherwise a slot is captured on its first use)."""
        for _ in self.slots:
            self(xyz_f1, xyz_f2)
        self.wait_all()

    def wait(self, slot):
        if self.events[slot] is not None:
            self.events[slot].synchronize()

    def wait_all(self):
        for s in self.streams or []:
            s.synchronize()


class StagedPipeline:
    """Throughput pipeline that never runs two sampling chains at once.

    A forward is captured as TWO graphs per slot: F = ingest + the furthest-point-sampling chain
    (depends on the input coordinates only; 64 workgroups that each pin a whole CU through their
    LDS table, latency-bound, ~2.5 ms at batch 32) and R = everything else (neighbour search, MFMA
    stacks, pose heads).  All F graphs are replayed on ONE stream, in batch order, so the chains of
    successive batches run back to back on the same 64 CUs; the R graphs alternate between two
    other streams and fill the remaining CUs.  With whole-forward graphs (``PipelinedForward``) two
    batches regularly sit in their F phase together (128 CUs pinned, the rest idle) or in their R
    phase together (no sampling in flight); here exactly one chain is in flight at any time.
    F(i) -> R(i) and R(i) -> F(i + slots) (buffer reuse) are event edges.

    ``pipe(xyz_f1, xyz_f2) -> (pose, slot)``: pose is the slot's static output, valid after
    ``wait(slot)``; a slot is reused every `slots` calls."""

    def __init__(self, net, slots=3, warmup=2):
        assert not net.training and getattr(net, "_fused", None) is not None, \
            "StagedPipeline needs an eval-mode net after prepare_fused()"
        self.net, self.nslots, self.warmup = net, slots, warmup
        self._slots = None
        self._i = 0

    def _capture_slot(self, xyz_f1, xyz_f2):
        fused = self.net._fused
        s1, s2 = xyz_f1.clone(), xyz_f2.clone()
        g_f, g_r = torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph()
        with torch.cuda.graph(g_f), torch.no_grad():
            state = fused.sample(s1, s2)
        with torch.cuda.graph(g_r, pool=g_f.pool()), torch.no_grad():   # same pool: `state` stays live
            pose, inter = fused.rest(state, return_intermediates=True)
        saved = self.net.log_mode
        self.net.log_mode = "device" if saved == "host" else saved      # no D2H inside a pipeline
        try:
            log = self.net._fused_log_dict(inter)
        finally:
            self.net.log_mode = saved
        return dict(s1=s1, s2=s2, g_f=g_f, g_r=g_r, state=state, pose=pose, log=log, ev_f=None, ev_r=None)

    def _setup(self, xyz_f1, xyz_f2):
        dev = xyz_f1.device
        side = torch.cuda.Stream(device=dev)
        side.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(side), torch.no_grad():
            for _ in range(self.warmup):          # allocator warm-up + one-time kernel attributes
                self.net._fused(xyz_f1, xyz_f2)
        torch.cuda.current_stream(dev).wait_stream(side)
        torch.cuda.synchronize(dev)
        self._slots = [self._capture_slot(xyz_f1, xyz_f2) for _ in range(self.nslots)]
        self.stream_f = torch.cuda.Stream(device=dev)
        self.streams_r = [torch.cuda.Stream(device=dev) for _ in range(2)]
        self._shape = (tuple(xyz_f1.shape), dev)

    def __call__(self, xyz_f1, xyz_f2):
        if self._slots is None:
            self._setup(xyz_f1, xyz_f2)
        assert (tuple(xyz_f1.shape), xyz_f1.device) == self._shape, "StagedPipeline is captured for one input shape"
        i = self._i
        self._i += 1
        k = i % self.nslots
        sl = self._slots[k]
        sf, sr = self.stream_f, self.streams_r[i % 2]
        sf.wait_stream(torch.cuda.current_stream(xyz_f1.device))     # inputs produced on the caller's stream
        if sl["ev_r"] is not None:
            sf.wait_event(sl["ev_r"])                                  # the slot's previous batch is done
        with torch.cuda.stream(sf):
            sl["s1"].copy_(xyz_f1)
            sl["s2"].copy_(xyz_f2)
            sl["g_f"].replay()
            sl["ev_f"] = torch.cuda.Event()
            sl["ev_f"].record(sf)
        sr.wait_event(sl["ev_f"])
        with torch.cuda.stream(sr):
            sl["g_r"].replay()
            sl["ev_r"] = torch.cuda.Event()
            sl["ev_r"].record(sr)
        log = sl["log"]
        self.last_log_dict = (log.fresh(to_host=self.net.log_mode == "host", ready=sl["ev_r"].synchronize)
                              if hasattr(log, "fresh") else log)
        return sl["pose"], k

    def prepare(self, xyz_f1, xyz_f2):
        """Capture all slots now (done on first use otherwise)."""
        if self._slots is None:
            self._setup(xyz_f1, xyz_f2)

    def wait(self, slot):
        ev = self._slots[slot]["ev_r"] if self._slots else None
        if ev is not None:
            ev.synchronize()

    def wait_all(self):
        if self._slots:
            self.stream_f.synchronize()
            for s in self.streams_r:
                s.synchronize()
