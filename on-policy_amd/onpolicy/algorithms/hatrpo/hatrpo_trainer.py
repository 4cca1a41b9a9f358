"""HATRPO trainer: per-agent trust-region step (natural gradient by conjugate gradients + backtracking line search
on the factor-weighted surrogate) next to an Adam step of the critic.  Interface and arithmetic of the reference's
onpolicy/algorithms/hatrpo/hatrpo_trainer.py (cal_value_loss :54, kl_divergence :124, conjugate_gradient :141,
fisher_vector_product :158, trpo_update :167, train :302).  Value-normaliser quirks and the advantage rule are
HAPPO's (see happo_trainer.py), which this class extends.

Evaluated with fewer passes than the reference, same numbers:
  * Fisher-vector products: the reference runs the actor twice and differentiates the KL twice for EVERY product
    (10 conjugate-gradient iterations + 1).  The KL of the policy to its own detached copy and its gradient graph do
    not depend on the vector, so they are built once per minibatch -- from the forward pass that also gives the
    surrogate -- and each product is one backward pass through that graph.
  * line search: the reference instantiates a fresh ``R_Actor`` per minibatch to hold the old parameters and runs it
    at every backtracking step.  The old policy's outputs on the minibatch are constants; they are taken once from
    the first forward pass.  (Consequence: no network is constructed here, so the global random stream is not
    advanced by an update, unlike the reference where the throw-away actor's initialisation draws from it.)
  * each backtracking step needs its accept / reject decision on the host: one sync per step (three scalars).
"""
import torch
import torch.nn as nn
from torch.nn.utils import parameters_to_vector, vector_to_parameters

from onpolicy.algorithms.happo.happo_trainer import HAPPO
from onpolicy.algorithms.utils.util import check
from onpolicy.utils.util import get_gard_norm

_CG_STEPS = 10
_CG_RESIDUAL = 1e-10
_DAMPING = 0.1


def _flat(grads, params):
    """Gradients as one vector laid out like ``parameters_to_vector`` (zeros where a parameter is unused)."""
    return torch.cat([(torch.zeros_like(p) if g is None else g).reshape(-1) for g, p in zip(grads, params)])


class _Dist(object):
    """Parameters of the action distribution on a minibatch: normalised logits (categorical heads) or mean / std."""

    def __init__(self, mean, std, logits):
        self.mean, self.std, self.logits = mean, std, logits

    def detach(self):
        return _Dist(*(None if x is None else x.detach() for x in (self.mean, self.std, self.logits)))


def kl_divergence(new, old):
    """Per-sample KL(old || new) [B, 1] the way the reference measures it (:124-139): for categorical heads the
    estimator r - 1 - log r summed over the actions, with r the probability ratio per action; the closed form for
    diagonal Gaussians.  ``old`` is treated as a constant."""
    if new.logits is not None:
        q, p = old.logits.detach(), new.logits
        kl = torch.exp(p - q) - 1 - p + q
    else:
        mu_old, std_old = old.mean.detach(), old.std.detach()
        kl = torch.log(new.std) - torch.log(std_old) + \
            (std_old.pow(2) + (mu_old - new.mean).pow(2)) / (2.0 * new.std.pow(2)) - 0.5
    return kl.sum(1, keepdim=True) if kl.dim() > 1 else kl


class HATRPO(HAPPO):
    def __init__(self, args, policy, device=torch.device("cpu")):
        super(HATRPO, self).__init__(args, policy, device=device)
        self.kl_threshold, self.ls_step, self.accept_ratio = args.kl_threshold, args.ls_step, args.accept_ratio
        if self.dp.active:
            raise NotImplementedError("the trust-region update has no data-parallel form here")

    def _fused_loss_allowed(self):
        return False

    # -- pieces of the trust-region step
    def _surrogate(self, ratio, factor, adv, active):
        objective = torch.sum(ratio * factor * adv, dim=-1, keepdim=True)
        if self._use_policy_active_masks:
            return (objective * active).sum() / active.sum()
        return objective.mean()

    def conjugate_gradient(self, fvp, b, nsteps=_CG_STEPS, residual_tol=_CG_RESIDUAL):
        """Solve F x = b with ``fvp(v) = F v``; stops early once the squared residual is below the tolerance."""
        x = torch.zeros_like(b)
        r, p = b.clone(), b.clone()
        rdotr = torch.dot(r, r)
        for _ in range(nsteps):
            Fp = fvp(p)
            alpha = rdotr / torch.dot(p, Fp)
            x += alpha * p
            r -= alpha * Fp
            new_rdotr = torch.dot(r, r)
            p = r + (new_rdotr / rdotr) * p
            rdotr = new_rdotr
            if rdotr < residual_tol:
                break
        return x

    def _fisher_operator(self, dist, params):
        """v -> (Hessian of mean KL(pi_detached || pi) at pi) v + 0.1 v, from ONE double-differentiable graph."""
        kl = kl_divergence(dist, dist.detach()).mean()
        kl_grad = _flat(torch.autograd.grad(kl, params, create_graph=True, allow_unused=True), params)

        def fvp(v):
            hv = torch.autograd.grad((kl_grad * v.detach()).sum(), params, retain_graph=True, allow_unused=True)
            return _flat(hv, params).detach() + _DAMPING * v
        return fvp

    def trpo_update(self, sample, update_actor=True):
        """One minibatch -> (value_loss, critic_grad_norm, kl, loss_improve, expected_improve, dist_entropy, ratio);
        kl / dist_entropy / ratio are those of the last backtracking step, accepted or not, as in the reference."""
        share_obs, obs, rnn_a, rnn_c, actions, value_preds, returns, masks, active, old_logp, adv, avail, factor = sample
        old_logp, adv, value_preds, returns, active, factor = (
            check(x).to(**self.tpdv) for x in (old_logp, adv, value_preds, returns, active, factor))
        policy, actor = self.policy, self.policy.actor
        evaluate = lambda: policy.evaluate_actions(share_obs, obs, rnn_a, rnn_c, actions, masks, avail, active)  # noqa: E731

        values, logp, dist_entropy, mean, std, logits = evaluate()
        # critic: one Adam step on the clipped value loss
        value_loss = self._value_loss(values, value_preds, returns, active, False)
        policy.critic_optimizer.zero_grad()
        (value_loss * self.value_loss_coef).backward()
        if self._use_max_grad_norm:
            critic_grad_norm = nn.utils.clip_grad_norm_(policy.critic.parameters(), self.max_grad_norm)
        else:
            critic_grad_norm = get_gard_norm(policy.critic.parameters())
        policy.critic_optimizer.step()

        # actor: search direction F^-1 g and the largest step inside the KL ball
        params = list(actor.parameters())
        ratio = torch.prod(torch.exp(logp - old_logp), dim=-1, keepdim=True)
        loss = self._surrogate(ratio, factor, adv, active)
        loss_grad = _flat(torch.autograd.grad(loss, params, retain_graph=True, allow_unused=True), params).detach()
        here = _Dist(mean, std, logits)
        fvp = self._fisher_operator(here, params)
        step_dir = self.conjugate_gradient(fvp, loss_grad)
        shs = 0.5 * torch.dot(step_dir, fvp(step_dir))
        full_step = step_dir / torch.sqrt(shs / self.kl_threshold)
        old_dist = here.detach()
        del fvp, here                                     # frees the double-backward graph

        start = parameters_to_vector(params).detach().clone()
        loss = float(loss.detach())
        expected_improve = float(torch.dot(loss_grad, full_step))
        fraction, accepted = 1.0, False
        kl = loss_improve = None
        for _ in range(self.ls_step):
            vector_to_parameters(start + fraction * full_step, params)
            with torch.no_grad():
                _, logp, dist_entropy, mean, std, logits = evaluate()
                ratio = torch.exp(logp - old_logp)                    # sic: not the product over action dims (:268)
                new_loss = self._surrogate(ratio, factor, adv, active)
                kl = kl_divergence(_Dist(mean, std, logits), old_dist).mean()
                new_loss_host, kl_host = torch.stack([new_loss, kl]).tolist()
            loss_improve = new_loss_host - loss
            if kl_host < self.kl_threshold and loss_improve / expected_improve > self.accept_ratio and loss_improve > 0:
                accepted = True
                break
            expected_improve *= 0.5
            fraction *= 0.5
        if not accepted:
            vector_to_parameters(start, params)
            print('policy update does not impove the surrogate')
        return value_loss, critic_grad_norm, kl, loss_improve, expected_improve, dist_entropy, ratio

    def train(self, buffer, update_actor=True):
        """One pass over the buffer in ``num_mini_batch`` minibatches (no ppo_epoch loop, :345-352) -> means of
        value_loss, kl, dist_entropy, loss_improve, expected_improve, critic_grad_norm, ratio."""
        advantages = self._advantages(buffer)
        if self._use_recurrent_policy:
            data_generator = buffer.recurrent_generator(advantages, self.num_mini_batch, self.data_chunk_length)
        elif self._use_naive_recurrent:
            data_generator = buffer.naive_recurrent_generator(advantages, self.num_mini_batch)
        else:
            data_generator = buffer.feed_forward_generator(advantages, self.num_mini_batch)
        keys = ('value_loss', 'kl', 'dist_entropy', 'loss_improve', 'expected_improve', 'critic_grad_norm', 'ratio')
        totals = dict.fromkeys(keys, 0.0)
        for sample in data_generator:
            value_loss, critic_grad_norm, kl, loss_improve, expected_improve, dist_entropy, imp_weights = \
                self.trpo_update(sample, update_actor)
            for k, v in zip(keys, (value_loss, kl, dist_entropy, loss_improve, expected_improve, critic_grad_norm,
                                   imp_weights.mean())):
                totals[k] += float(v.detach()) if torch.is_tensor(v) else float(v)
        return {k: v / self.num_mini_batch for k, v in totals.items()}
