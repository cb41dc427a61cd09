#!/usr/bin/env python3
"""Developer probe: does the next batch's sampling chain overlap with the current training step?
(training.TrainStep(sample_ahead=True); result in profiles/r03/r03_train_sample_ahead_probe.txt)"""
import sys, os, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import pwclonet_pylidarslam_amd
pwclonet_pylidarslam_amd.configure_hw_queues(8)
import torch
import bench
from pwclonet_pylidarslam_amd.loss import PWCLONetLossModule
from pwclonet_pylidarslam_amd.pwclonet import PWCLONet
from pwclonet_pylidarslam_amd.training import PWCLONetWithLoss, TrainStep
dev = torch.device("cuda:0")
torch.manual_seed(7)
net = PWCLONet(dict(num_input_channels=3, sequence_len=2, device=str(dev), scalar_last=False, log_mode="none")).to(dev).train()
unit = PWCLONetWithLoss(net, PWCLONetLossModule(dict(with_exp_weights=True, init_weights=[0.0, -2.5], loss_option="l2_norm", nb_levels=4, scalar_last=False)).to(dev))
opt = torch.optim.Adam(unit.parameters(), lr=1e-4, capturable=True, fused=True)
x1, x2 = bench.make_batch(32, 8192, 2000, dev)
gt = torch.zeros(32, 7, device=dev); gt[:, 3] = 1
ts = TrainStep(unit, opt, x1, x2, gt, graph=True, sample_ahead=True)
def timeit(fn, n=10):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
print("main graph alone      %.2f ms" % timeit(lambda: ts.graph.replay()))
def side_only():
    with torch.cuda.stream(ts.side): ts.sample_graph.replay()
print("sampling graph alone  %.2f ms" % timeit(side_only))
print("full step             %.2f ms" % timeit(lambda: ts.step()))
s2 = torch.cuda.Stream()
def both_nondefault():
    s2.wait_stream(torch.cuda.current_stream()); ts.side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(ts.side): ts.sample_graph.replay()
    with torch.cuda.stream(s2): ts.graph.replay()
    torch.cuda.current_stream().wait_stream(s2); torch.cuda.current_stream().wait_stream(ts.side)
print("both on non-default streams %.2f ms" % timeit(both_nondefault))
def main_then_side():
    ts.side.wait_stream(torch.cuda.current_stream())
    ts.graph.replay()
    with torch.cuda.stream(ts.side): ts.sample_graph.replay()
    torch.cuda.current_stream().wait_stream(ts.side)
print("main first, then side %.2f ms" % timeit(main_then_side))
def eager_side():
    ts.side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(ts.side): ts._sample_next()
    ts.graph.replay()
    torch.cuda.current_stream().wait_stream(ts.side)
print("side eager kernels + main graph %.2f ms" % timeit(eager_side))
def main_on_s2_alone():
    s2.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s2): ts.graph.replay()
    torch.cuda.current_stream().wait_stream(s2)
print("main graph alone on a non-default stream %.2f ms" % timeit(main_on_s2_alone))
a = torch.randn(4096, 4096, device=dev); b = torch.randn(4096, 4096, device=dev)
def mm_only():
    for _ in range(20): torch.mm(a, b)
print("20 matmuls alone %.2f ms" % timeit(mm_only))
def mm_and_side():
    ts.side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(ts.side): ts._sample_next()
    for _ in range(20): torch.mm(a, b)
    torch.cuda.current_stream().wait_stream(ts.side)
print("20 matmuls + side eager sampling %.2f ms" % timeit(mm_and_side))
def mm_and_side_graph():
    ts.side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(ts.side): ts.sample_graph.replay()
    for _ in range(20): torch.mm(a, b)
    torch.cuda.current_stream().wait_stream(ts.side)
print("20 matmuls + side sampling graph %.2f ms" % timeit(mm_and_side_graph))
# the step graph captured ON the non-default stream it is replayed on
s3 = torch.cuda.Stream()
g3 = torch.cuda.CUDAGraph()
opt.zero_grad(set_to_none=True)
torch.cuda.synchronize()
with torch.cuda.graph(g3, stream=s3):
    loss3, _, _ = ts._forward()
    loss3.backward()
    opt.step()
torch.cuda.synchronize()
def g3_alone():
    s3.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s3): g3.replay()
    torch.cuda.current_stream().wait_stream(s3)
print("step graph captured and replayed on the same non-default stream, alone %.2f ms" % timeit(g3_alone))
def g3_and_side():
    s3.wait_stream(torch.cuda.current_stream()); ts.side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(ts.side): ts.sample_graph.replay()
    with torch.cuda.stream(s3): g3.replay()
    torch.cuda.current_stream().wait_stream(s3); torch.cuda.current_stream().wait_stream(ts.side)
print("... beside the sampling graph on another stream %.2f ms" % timeit(g3_and_side))
