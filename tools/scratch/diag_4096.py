import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import torch
from oracle import model as om, params
from pwclonet_pylidarslam_amd import synthetic, fused
from pwclonet_pylidarslam_amd.pwclonet import PWCLONet
cuda = torch.device("cuda:0")
pc1, pc2, _, _ = synthetic.kitti_like_pair(41, 4096, 3)
x1 = torch.from_numpy(pc1[:, :, :3]).permute(0, 2, 1).contiguous()
x2 = torch.from_numpy(pc2[:, :, :3]).permute(0, 2, 1).contiguous()
net = PWCLONet(dict(num_input_channels=3, sequence_len=2, device=str(cuda), scalar_last=False, log_mode="none"))
params.fill_state_dict(net.state_dict())
net = net.to(cuda).eval()
sd = {k: v.detach().cpu() for k, v in net.state_dict().items()}
with torch.no_grad():
    c, _ = net(x1.to(cuda), None, x2.to(cuda), None)
a, inter = fused.FusedPWCLONet(net)(x1.to(cuda), x2.to(cuda), return_intermediates=True)
taps = {}
want = om.pwclonet_forward(sd, x1, x2, taps)
print("fused vs oracle per pair/level:\n", (a.cpu() - want).abs().amax(dim=2))
print("module vs oracle per pair/level:\n", (c.cpu() - want).abs().amax(dim=2))
for l in (3, 2, 1):
    for k in ("idx_q", "idx"):
        key = "pwr%d.cv.%s" % (l, k)
        d = (inter["lists"][key].cpu() != taps[key]).any(dim=2).sum(dim=1)
        print(key, "rows differing fused vs oracle per pair:", d.tolist())
