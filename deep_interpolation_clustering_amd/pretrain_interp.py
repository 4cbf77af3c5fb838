"""Drop-in for the reference's ``pretrain_interp`` module: the interpolation auto-encoder without
the clustering head (pretrain_interp.py:90-215) on the HIP kernels."""
from ._net_common import AuxFc, DecoderRNN, EncoderRNN, FakeDetFc, FuturePredFc, NetBase   # noqa: F401 (upstream names)


class Net(NetBase):
    clustering = False

    def triplet_loss(self):
        return None                      # pretrain_interp.py:202-203
