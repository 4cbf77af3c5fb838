"""Drop-in for the reference's ``dec`` module (dec.py:13-76) on the HIP kernels (``csrc/dic_latent.hip``)."""
from typing import Optional

import torch
import torch.nn as nn
from torch.nn import Parameter

from . import ops


class ClusterAssignment(nn.Module):
    """Student-t soft assignment between latents and learnable centroids (dec.py:13-63).
    ``cluster_centers`` (K,D) is Xavier-uniform unless given; trained by the optimizer like upstream."""

    def __init__(self, cluster_number: int, embedding_dimension: int, alpha: float = 1.0,
                 cluster_centers: Optional[torch.Tensor] = None) -> None:
        super().__init__()
        self.embedding_dimension = embedding_dimension
        self.cluster_number = cluster_number
        self.alpha = alpha
        if cluster_centers is None:
            cluster_centers = torch.zeros(cluster_number, embedding_dimension, dtype=torch.float)
            nn.init.xavier_uniform_(cluster_centers)
        self.cluster_centers = Parameter(cluster_centers)
        self.last_colsum = None

    def init_center(self, initial_cluster_centers):
        """dec.py:43-44; the values are copied when the parameter is a view of a flat bucket."""
        new = torch.as_tensor(initial_cluster_centers).detach()
        if tuple(new.shape) == tuple(self.cluster_centers.shape):
            with torch.no_grad():
                self.cluster_centers.data.copy_(new.to(self.cluster_centers.device, self.cluster_centers.dtype))
        else:
            self.cluster_centers.data = new

    def get_center(self):
        return self.cluster_centers

    def forward(self, batch: torch.Tensor) -> torch.Tensor:
        """(B,D) -> q (B,K).  The kernel also emits the column sums f_j of this batch; they are kept
        on ``last_colsum`` so ``target_distribution`` need not re-reduce q."""
        q, colsum = ops.dec_soft_assign(batch, self.cluster_centers, self.alpha, return_colsum=True)
        self.last_colsum = (q, colsum)
        return q


def target_distribution(batch: torch.Tensor, colsum: Optional[torch.Tensor] = None) -> torch.Tensor:
    """dec.py:66-76: p_ij = (q_ij^2/f_j) / sum_j(q_ij^2/f_j), f_j = sum_i q_ij over the (global) batch."""
    return ops.dec_target(batch, colsum)
