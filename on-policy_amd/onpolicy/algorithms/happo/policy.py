"""HAPPO policy: the MAPPO actor / critic pair (reference onpolicy/algorithms/happo/policy.py:5-131 is
R_MAPPOPolicy with ``self.args`` kept; one such policy exists per agent)."""
from onpolicy.algorithms.r_mappo.algorithm.rMAPPOPolicy import R_MAPPOPolicy


class HAPPO_Policy(R_MAPPOPolicy):
    def __init__(self, args, obs_space, cent_obs_space, act_space, device=None):
        kwargs = {} if device is None else {"device": device}
        super(HAPPO_Policy, self).__init__(args, obs_space, cent_obs_space, act_space, **kwargs)
        self.args = args
