"""``MultiDiscrete`` action space: several discrete sub-actions, each with its own inclusive ``[min, max]`` range
(the MPE scenarios with communication use it).  Interface of the reference's onpolicy/utils/multi_discrete.py
(MultiDiscrete :7: ``low``, ``high``, ``num_discrete_space``, ``n``, ``sample``, ``contains``, ``shape``) without
the gym base class -- the networks and buffers recognise spaces by class name (utils/util.py:31-52,
algorithms/utils/act.py:24-30), which is ``'MultiDiscrete'`` here as well.
"""
import numpy as np


class MultiDiscrete(object):
    """``MultiDiscrete([[0, 4], [0, 1], [0, 1]])``: a 5-way, a 2-way and another 2-way choice; value 0 is the
    no-op of each sub-action."""

    def __init__(self, array_of_param_array):
        bounds = np.asarray(array_of_param_array, dtype=np.int64).reshape(-1, 2)
        self.low, self.high = bounds[:, 0].copy(), bounds[:, 1].copy()

    @property
    def num_discrete_space(self):
        return int(self.low.shape[0])

    @property
    def shape(self):
        """Number of sub-actions (an int, as the reference has it -- not a tuple)."""
        return self.num_discrete_space

    @property
    def n(self):
        return int(self.high.sum()) + 2

    def sample(self):
        """One uniformly drawn value per sub-action (numpy's global generator, like the reference)."""
        span = self.high - self.low + 1
        return [int(lo + np.floor(u * s)) for lo, s, u in zip(self.low, span, np.random.rand(len(span)))]

    def contains(self, x):
        x = np.asarray(x)
        return x.shape == self.low.shape and bool(np.all((x >= self.low) & (x <= self.high)))

    def __repr__(self):
        return "MultiDiscrete%d" % self.num_discrete_space

    def __eq__(self, other):
        return isinstance(other, MultiDiscrete) and np.array_equal(self.low, other.low) \
            and np.array_equal(self.high, other.high)

    __hash__ = None
