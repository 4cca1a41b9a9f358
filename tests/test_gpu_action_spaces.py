"""-m gpu: continuous / MultiDiscrete heads through the HBM buffer and the trainer on the device against the
reference fixtures of oracle/make_golden_spaces.py (same CPU seed => same permutations)."""
import pytest
import torch

from test_action_spaces_cpu import BUF, DISCRETE, build_space_case, check_final

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("cname", ["box", "multidiscrete", "cnn", "naive_gru", "gru2_xavier"])
def test_other_action_heads_on_device_vs_reference(gold, cname):
    from onpolicy.utils.shared_buffer import SharedReplayBuffer
    dev = torch.device("cuda", 0)
    z = gold.npz("space_cases")
    key = "spc_%s_" % cname
    meta, args, spaces, policy, trainer = build_space_case(gold, cname, device=dev)
    args.sampler_rng = "host"
    buf = SharedReplayBuffer(args, meta["A"], *spaces, device=dev)
    for name in BUF + (("available_actions",) if cname in DISCRETE else ()):
        dst = getattr(buf, name)
        if dst.stride()[0] != 0:
            dst.copy_(torch.from_numpy(z[key + "buf_" + name]))
    buf.compute_returns(z[key + "next_value"], trainer.value_normalizer)
    trainer.prep_training()
    torch.manual_seed(21)
    info = trainer.train(buf)
    check_final(z, key, meta, info, policy, rel=1e-3, atol=1e-4)
