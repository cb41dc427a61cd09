import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench
from pwclonet_pylidarslam_amd.loss import PWCLONetLossModule
from pwclonet_pylidarslam_amd.pwclonet import PWCLONet
from pwclonet_pylidarslam_amd.training import PWCLONetWithLoss
dev = torch.device("cuda:0")
torch.manual_seed(7)
net = PWCLONet(dict(num_input_channels=3, sequence_len=2, device=str(dev), scalar_last=False, log_mode="none")).to(dev).train()
unit = PWCLONetWithLoss(net, PWCLONetLossModule(dict(with_exp_weights=True, init_weights=[0.0, -2.5], loss_option="l2_norm", nb_levels=4, scalar_last=False)).to(dev))
opt = torch.optim.Adam(unit.parameters(), lr=1e-4, fused=True)
x1, x2 = bench.make_batch(32, 8192, 2000, dev)
gt = torch.zeros(32, 7, device=dev); gt[:, 3] = 1.0
def step():
    opt.zero_grad(set_to_none=True)
    loss, _, _ = unit(x1, x2, gt)
    fwd_peak = torch.cuda.max_memory_allocated()
    loss.backward()
    opt.step()
    return fwd_peak, loss.item()
for _ in range(2): step()
torch.cuda.synchronize(); torch.cuda.reset_peak_memory_stats()
base = torch.cuda.memory_allocated()
fwd_peak, l = step()
torch.cuda.synchronize()
print("PWCLO_HIP_STACK=%s: resident %.2f GB, peak at end of forward %.2f GB, peak of the step %.2f GB, loss %.6f"
      % (os.environ.get("PWCLO_HIP_STACK", "1"), base / 2**30, fwd_peak / 2**30, torch.cuda.max_memory_allocated() / 2**30, l))
