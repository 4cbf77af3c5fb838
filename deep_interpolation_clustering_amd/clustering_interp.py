"""Drop-in for the reference's ``clustering_interp`` module: the joint interpolation + DEC network
(clustering_interp.py:89-247) on the HIP kernels.  See ``_net_common`` for the shared body."""
import torch.nn.functional as F

from . import dist, ops
from ._net_common import AuxFc, DecoderRNN, EncoderRNN, FakeDetFc, FuturePredFc, NetBase   # noqa: F401 (upstream names)


class Net(NetBase):
    clustering = True

    def init_cluster_center(self, initial_cluster_centers):
        self.cluster_assignment.init_center(initial_cluster_centers)

    def get_cluster_center(self):
        return self.cluster_assignment.get_center()

    def kl_loss(self, label, pred):
        """F.kl_div(pred.log(), label, 'batchmean') over the global batch (clustering_interp.py:205-207),
        with its gradient wrt pred produced by the same kernel pass."""
        return {'kl': ops.kl_batchmean(label, pred)}

    def triplet_loss(self, anchor, positive, negative, margin):
        triple = F.triplet_margin_loss(anchor, positive, negative, margin=margin, reduction='sum')
        return {'triplet': dist.global_mean(triple, anchor.size(0))}
