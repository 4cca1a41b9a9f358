"""Gradient clipping + Adam step of one network as two launches (K13, ``mappo_clip_adam``).

What the reference does per network between ``backward()`` and the next minibatch (r_mappo.py:146-167:
``nn.utils.clip_grad_norm_(parameters, max_grad_norm)`` or ``get_gard_norm``, then ``optimizer.step()`` of the
``torch.optim.Adam`` built in rMAPPOPolicy.py:31-37) is ~8 launches per network in PyTorch -- foreach norms, stack, norm,
clamp, multiply, the fused Adam kernel, the step counters.  The kernel pair does the same arithmetic on the optimiser's OWN
state tensors (``exp_avg``, ``exp_avg_sq``, ``step``), so ``state_dict()``, ``lr_decay`` and checkpoints keep working and
a later ``optimizer.step()`` continues from the same state.  ``MAPPO_FUSED_OPTIM=0`` keeps the PyTorch calls.
"""
import os

import torch

from onpolicy import _native


def enabled():
    return os.environ.get("MAPPO_FUSED_OPTIM", "1") != "0"


def supported(optimizer, params):
    """torch.optim.Adam without amsgrad / maximize / foreach-only features, one parameter group, at most
    ``ADAM_MAX_TENSORS`` float32 contiguous HIP tensors that all have a gradient."""
    if not enabled() or type(optimizer) is not torch.optim.Adam or len(optimizer.param_groups) != 1:
        return False
    g = optimizer.param_groups[0]
    if g.get("amsgrad") or g.get("maximize") or g.get("differentiable") or g.get("decoupled_weight_decay"):
        return False
    if not 0 < len(params) <= _native.ADAM_MAX_TENSORS or len(params) != len(g["params"]):
        return False
    for p in params:
        if not (p.is_cuda and p.dtype == torch.float32 and p.is_contiguous()):
            return False
        if p.grad is None or p.grad.dtype != torch.float32 or not p.grad.is_contiguous():
            return False
    return not torch.is_tensor(g["lr"])


def clip_and_step(optimizer, params, max_grad_norm, lr_device=None):
    """-> the total L2 norm of the gradients before clipping (device scalar).  ``max_grad_norm`` None / <= 0: no clipping
    (the reference's get_gard_norm branch).  ``lr_device``: a float64 device tensor [1] the kernel reads the learning rate
    from instead of the optimiser's Python float (launches captured into a HIP graph, algorithms/r_mappo/update_graph.py)."""
    lib, p = _native.lib(), _native.ptr
    group = optimizer.param_groups[0]
    dev = params[0].device
    a = _native.Adam()
    for i, prm in enumerate(params):
        st = optimizer.state[prm]
        if len(st) == 0:        # what torch.optim.Adam._init_group creates for a fused optimiser on first use
            st["step"] = torch.zeros((), dtype=torch.float32, device=dev)
            st["exp_avg"] = torch.zeros_like(prm, memory_format=torch.preserve_format)
            st["exp_avg_sq"] = torch.zeros_like(prm, memory_format=torch.preserve_format)
        if not (torch.is_tensor(st["step"]) and st["step"].is_cuda and st["step"].dtype == torch.float32):
            st["step"] = torch.as_tensor(float(st["step"]), dtype=torch.float32, device=dev)
        a.param[i], a.grad[i] = prm.data_ptr(), prm.grad.data_ptr()
        a.exp_avg[i], a.exp_avg_sq[i], a.step[i] = st["exp_avg"].data_ptr(), st["exp_avg_sq"].data_ptr(), st["step"].data_ptr()
        a.numel[i] = prm.numel()
    a.n = len(params)
    a.lr, (a.beta1, a.beta2) = float(group["lr"]), group["betas"]
    a.eps, a.weight_decay = float(group["eps"]), float(group["weight_decay"])
    a.max_grad_norm = float(max_grad_norm) if max_grad_norm else 0.0
    norm = torch.empty(1, dtype=torch.float32, device=dev)
    ws = torch.empty(lib.mappo_adam_workspace_floats(), dtype=torch.float32, device=dev)
    a.grad_norm, a.workspace = p(norm), p(ws)
    if lr_device is not None:
        assert lr_device.dtype == torch.float64 and lr_device.is_cuda and lr_device.numel() == 1
        a.lr_device = p(lr_device)
    _native.check(lib.mappo_clip_adam(a, _native.stream_of(dev)), "mappo_clip_adam")
    return norm[0]
