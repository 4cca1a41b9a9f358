
from onpolicy.envs import _extend as _extend_path   # noqa: E402

_extend_path(__path__, "mpe")      # other scenarios / MPE_env from an external env tree (MAPPO_ENVS_PATH)
