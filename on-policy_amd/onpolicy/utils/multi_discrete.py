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
        self.low = np.array([x[0] for x in array_of_param_array])
        self.high = np.array([x[1] for x in array_of_param_array])
        self.num_discrete_space = self.low.shape[0]
        self.n = np.sum(self.high) + 2

    def sample(self):
        """One uniformly drawn value per sub-action."""
        u = np.random.rand(self.num_discrete_space)
        return [int(x) for x in np.floor((self.high - self.low + 1.0) * u + self.low)]

    def contains(self, x):
        x = np.array(x)
        return len(x) == self.num_discrete_space and bool((x >= self.low).all()) and bool((x <= self.high).all())

    @property
    def shape(self):
        return self.num_discrete_space

    def __repr__(self):
        return "MultiDiscrete" + str(self.num_discrete_space)

    def __eq__(self, other):
        return np.array_equal(self.low, other.low) and np.array_equal(self.high, other.high)

    __hash__ = None
