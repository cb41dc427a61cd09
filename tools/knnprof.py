import sys, os, torch, numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "tools"))
from microbench import clouds
from pwclonet_pylidarslam_amd.pointnet2_ops import _ext as E
dev = torch.device("cuda:0")
for (k, n, s) in ((32, 8192, 2048), (32, 2048, 1024), (8, 1024, 2048), (4, 2048, 2048)):
    x = clouds(32, n, dev); q = (x[:, :s] if s <= n else clouds(32, s, dev))[:, :s].contiguous() + 0.01
    for _ in range(3): E.knn_point(k, x, q)
torch.cuda.synchronize()
