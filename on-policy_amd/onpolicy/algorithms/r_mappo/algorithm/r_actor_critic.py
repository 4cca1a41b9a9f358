"""Actor and critic networks of (R)MAPPO: trunk -> [GRU] -> action head / value head.

Class names, constructor signatures, forward signatures and parameter names are those of the
reference's onpolicy/algorithms/r_mappo/algorithm/r_actor_critic.py (R_Actor :12, forward :44,
evaluate_actions :73; R_Critic :120, forward :156).  The maths stays in PyTorch (rocBLAS / MIOpen on
ROCm).  Inputs may be numpy arrays (reference runners) or tensors that already live in HBM (the
device buffer and its samplers): ``_to_device`` is a no-op for the latter, so the update path does
no host<->device traffic.
"""
import torch
import torch.nn as nn

from onpolicy.algorithms.utils.util import init, check
from onpolicy.algorithms.utils.cnn import CNNBase
from onpolicy.algorithms.utils.mlp import MLPBase
from onpolicy.algorithms.utils.rnn import RNNLayer
from onpolicy.algorithms.utils.act import ACTLayer
from onpolicy.algorithms.utils.popart import PopArt
from onpolicy.algorithms.utils.tall_linear import TallLinear
from onpolicy.algorithms.utils import fused_mlp
from onpolicy.algorithms.utils.fused_mlp import RowSource
from onpolicy.utils.util import get_shape_from_obs_space


def _trunk(args, shape):
    return (CNNBase if len(shape) == 3 else MLPBase)(args, shape)


class _DeviceMixin(object):
    def _to_device(self, *xs):
        out = []
        for x in xs:
            # a RowSource (rows of the HBM buffer named by a sampler minibatch) is already on the device
            out.append(x if (x is None or isinstance(x, RowSource)) else check(x).to(**self.tpdv))
        return out

    def _fuses_head(self, x, head):
        """The trunk AND this output Linear run as one fused kernel launch for input ``x`` (feed-forward nets)."""
        return not self._recurrent and hasattr(self.base, "fuses") and self.base.fuses(x) and fused_mlp.head_supported(head)


class R_Actor(nn.Module, _DeviceMixin):
    def __init__(self, args, obs_space, action_space, device=torch.device("cpu")):
        super(R_Actor, self).__init__()
        self.hidden_size = args.hidden_size
        self._gain = args.gain
        self._use_orthogonal = args.use_orthogonal
        self._use_policy_active_masks = args.use_policy_active_masks
        self._use_naive_recurrent_policy = args.use_naive_recurrent_policy
        self._use_recurrent_policy = args.use_recurrent_policy
        self._recurrent_N = args.recurrent_N
        self.tpdv = dict(dtype=torch.float32, device=device)
        self.algo = args.algorithm_name

        self.base = _trunk(args, get_shape_from_obs_space(obs_space))
        if self._recurrent:
            self.rnn = RNNLayer(self.hidden_size, self.hidden_size, self._recurrent_N, self._use_orthogonal)
            self.rnn.matrix_arithmetic = getattr(args, "matrix_arithmetic", None)       # (K12, like the trunk's K9)
        self.act = ACTLayer(action_space, self.hidden_size, self._use_orthogonal, self._gain, args)
        # built on the CPU, then moved: init draws come from the CPU generator on every device
        self.to(device)

    @property
    def _recurrent(self):
        return self._use_naive_recurrent_policy or self._use_recurrent_policy

    def _features(self, obs, rnn_states, masks, obs_standardized=False):
        feats = self.base(obs, standardized=True) if obs_standardized else self.base(obs)
        if self._recurrent:
            feats, rnn_states = self.rnn(feats, rnn_states, masks)
        return feats, rnn_states

    def forward(self, obs, rnn_states, masks, available_actions=None, deterministic=False):
        obs, rnn_states, masks, available_actions = self._to_device(obs, rnn_states, masks, available_actions)
        obs = fused_mlp.rollout_rows(obs, self.base)      # rollout on the device: the trunk (+ head) as K9 launches
        if isinstance(obs, RowSource) and self.act.action_type == "Discrete" and \
                self._fuses_head(obs, self.act.action_out.linear):
            logits = self.base(obs, head=self.act.action_out.linear)          # standardise + trunk + head: two launches
            actions, action_log_probs = self.act.from_logits(logits, available_actions, deterministic)
            return actions, action_log_probs, rnn_states
        feats, rnn_states = self._features(obs, rnn_states, masks)
        actions, action_log_probs = self.act(feats, available_actions, deterministic)
        return actions, action_log_probs, rnn_states

    def evaluate_actions(self, obs, rnn_states, action, masks, available_actions=None, active_masks=None,
                         obs_standardized=False):
        obs, rnn_states, action, masks, available_actions, active_masks = self._to_device(
            obs, rnn_states, action, masks, available_actions, active_masks)
        feats, _ = self._features(obs, rnn_states, masks, obs_standardized)
        # "hatrpo": the trust-region trainer also needs the distribution's parameters (r_actor_critic.py:102-108)
        evaluate = self.act.evaluate_actions_trpo if self.algo == "hatrpo" else self.act.evaluate_actions
        return evaluate(feats, action, available_actions,
                        active_masks=active_masks if self._use_policy_active_masks else None)


    def evaluate_logits(self, obs, rnn_states, masks, obs_standardized=False):
        """Raw outputs of the Discrete action head for the fused PPO loss (K7), [B, n_actions]."""
        obs, rnn_states, masks = self._to_device(obs, rnn_states, masks)
        head = self.act.action_out.linear
        if self._fuses_head(obs, head):
            return self.base(obs, head=head)          # gather + trunk + head: one launch
        if self._recurrent:        # the head inside the GRU chunk kernels (K12) where they take the layer
            feats = self.base(obs, standardized=True) if obs_standardized else self.base(obs)
            if self.rnn.head_ok(feats, head):
                return self.rnn(feats, rnn_states, masks, head=head)[0]
            return head(self.rnn(feats, rnn_states, masks)[0])
        feats, _ = self._features(obs, rnn_states, masks, obs_standardized)
        return head(feats)


class R_Critic(nn.Module, _DeviceMixin):
    def __init__(self, args, cent_obs_space, device=torch.device("cpu")):
        super(R_Critic, self).__init__()
        self.hidden_size = args.hidden_size
        self._use_orthogonal = args.use_orthogonal
        self._use_naive_recurrent_policy = args.use_naive_recurrent_policy
        self._use_recurrent_policy = args.use_recurrent_policy
        self._recurrent_N = args.recurrent_N
        self._use_popart = args.use_popart
        self.tpdv = dict(dtype=torch.float32, device=device)
        w_init = nn.init.orthogonal_ if self._use_orthogonal else nn.init.xavier_uniform_

        self.base = _trunk(args, get_shape_from_obs_space(cent_obs_space))
        if self._recurrent:
            self.rnn = RNNLayer(self.hidden_size, self.hidden_size, self._recurrent_N, self._use_orthogonal)
            self.rnn.matrix_arithmetic = getattr(args, "matrix_arithmetic", None)       # (K12, like the trunk's K9)
        # built and initialised on the host like every other layer; self.to(device) below moves it
        head = PopArt(self.hidden_size, 1) if self._use_popart else TallLinear(self.hidden_size, 1)
        self.v_out = init(head, w_init, lambda b: nn.init.constant_(b, 0))
        self.to(device)

    @property
    def _recurrent(self):
        return self._use_naive_recurrent_policy or self._use_recurrent_policy

    def forward(self, cent_obs, rnn_states, masks, obs_standardized=False):
        cent_obs, rnn_states, masks = self._to_device(cent_obs, rnn_states, masks)
        if not obs_standardized:
            cent_obs = fused_mlp.rollout_rows(cent_obs, self.base)        # rollout / bootstrap value: K9 (see R_Actor.forward)
        if self._fuses_head(cent_obs, self.v_out):
            return self.base(cent_obs, head=self.v_out), rnn_states
        feats = self.base(cent_obs, standardized=True) if obs_standardized else self.base(cent_obs)
        if self._recurrent:
            if type(self.v_out) is not PopArt and self.rnn.head_ok(feats, self.v_out):
                return self.rnn(feats, rnn_states, masks, head=self.v_out)      # v_out inside the GRU chunk kernels
            feats, rnn_states = self.rnn(feats, rnn_states, masks)
        return self.v_out(feats), rnn_states
