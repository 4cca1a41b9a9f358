import sys
sys.path.insert(0, "on-policy_amd"); sys.path.insert(0, "tests")
import numpy as np, torch
from onpolicy.algorithms.utils import fused_mlp
from onpolicy.utils.shared_buffer import SharedReplayBuffer
from helpers import Box, Discrete, make_args
dev = torch.device("cuda", 0)
args = make_args(episode_length=4, n_rollout_threads=4, hidden_size=64)
buf = SharedReplayBuffer(args, 2, Box((48,)), Box((384,)), Discrete(5), device=dev)
torch.manual_seed(0)
for D in (48, 384):
    x = torch.randn(200000, D, device=dev)
    x64 = x.double()
    ref = (x64 - x64.mean(1, keepdim=True)) / torch.sqrt(x64.var(1, unbiased=False, keepdim=True) + 1e-5)
    cpu = torch.nn.functional.layer_norm(x.cpu(), (D,), eps=1e-5).to(dev)           # what the reference evaluates (PyTorch CPU)
    for name, got in (("mlp kernel", fused_mlp.standardize_rows(x)[:, :D]), ("K2 code", buf._standardize_field(x)[:, :D]), ("torch cpu layer_norm", cpu)):
        e = (got.double() - ref)
        scale = ((got.double() ** 2).sum(1) / D).mean().item() - ((ref ** 2).sum(1) / D).mean().item()
        print(D, "%-22s max|err| %.3e  mean|err| %.3e  mean signed err*x %.3e  scale bias %.3e  vs cpu max %.3e mean %.3e" % (
            name, e.abs().max().item(), e.abs().mean().item(), (e * ref).mean().item(), scale,
            (got - cpu).abs().max().item(), (got - cpu).abs().mean().item()))
