"""Stand-alone running normaliser that HAPPO uses under ``--use_popart`` (reference
onpolicy/algorithms/utils/popart_hatrpo.py: PopArt :8, forward :37, normalize :64, denormalize :67).
It is the ValueNorm statistics with one twist: ``normalize`` (train=True, the default) first folds the
batch it is given into the running moments -- so every call moves the statistics."""
from onpolicy.utils.valuenorm import ValueNorm


class PopArt(ValueNorm):
    def forward(self, input_vector, train=True):
        x = self._as_tensor(input_vector)
        if train:
            self.update(x.detach())
        return ValueNorm.normalize(self, x)

    def normalize(self, input_vector, train=True):
        return self.forward(input_vector, train)
