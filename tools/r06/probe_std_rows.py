import sys
sys.path.insert(0, "tests"); sys.path.insert(0, "on-policy_amd")
import torch
from onpolicy.algorithms.utils import fused_mlp
from onpolicy.utils.shared_buffer import SharedReplayBuffer
from helpers import Box, Discrete, make_args
dev = torch.device("cuda", 0)
args = make_args(episode_length=4, n_rollout_threads=4, hidden_size=64)
buf = SharedReplayBuffer(args, 2, Box((48,)), Box((384,)), Discrete(5), device=dev)
torch.manual_seed(0)
for n, D in ((204800, 384), (204800, 48), (102400, 384), (51200, 384), (1000, 384), (204801, 54), (13107200, 48)):
    x = torch.randn(n, D, device=dev)
    a = fused_mlp.standardize_rows(x)
    b = torch.full_like(a, 7.0)
    b = buf._standardize_field(x, out=b)
    torch.cuda.synchronize()
    bad = ((a - b).abs() > 1e-5).any(1)
    print(n, D, "rows off by > 1e-5:", int(bad.sum()), "first bad rows", bad.nonzero().flatten()[:6].tolist(), "untouched (7.0) rows", int((b == 7.0).all(1).sum()))
