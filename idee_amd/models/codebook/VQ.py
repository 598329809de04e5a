"""Nearest-neighbour vector quantiser on MI355X — the default path of the reference's models/codebook/VQ.py
(class VQ :737-1112 with EuclideanCodebook :273-555): Euclidean argmin lookup, straight-through gradient,
commitment mse, learnable codebook.  The reference's models/build.py:87 never constructs it (LFQ is hard-wired);
it is provided because the path's brief names it.  Everything outside that default path (EMA updates, k-means
init, dead-code expiry, gumbel / reinmax sampling, cosine similarity, multi-head, affine params, orthogonal
regularisation, masks) raises NotImplementedError.  State-dict keys match the reference
(`_codebook.embed`, `_codebook.initted`, `_codebook.cluster_size`, `_codebook.embed_avg`)."""
from __future__ import annotations

import torch
import torch.nn as nn

from ... import ops


class _Codebook(nn.Module):
    def __init__(self, dim, codebook_size):
        super().__init__()
        embed = torch.empty(1, codebook_size, dim)
        nn.init.kaiming_uniform_(embed)                       # uniform_init, VQ.py:68-71
        self.embed = nn.Parameter(embed)
        self.register_buffer("initted", torch.Tensor([True]))
        self.register_buffer("cluster_size", torch.zeros(1, codebook_size))
        self.register_buffer("embed_avg", embed.clone())


class VQ(nn.Module):
    def __init__(self, dim=16, codebook_size=2, codebook_dim=16, heads=1, commitment_weight=1., **unsupported):
        super().__init__()
        defaults = dict(separate_codebook_per_head=False, decay=0.8, eps=1e-5, freeze_codebook=False, kmeans_init=False,
                        kmeans_iters=10, sync_kmeans=True, use_cosine_sim=False, threshold_ema_dead_code=0, channel_last=True,
                        accept_image_fmap=False, commitment_use_cross_entropy_loss=False, orthogonal_reg_weight=0.,
                        orthogonal_reg_active_codes_only=False, orthogonal_reg_max_codes=None, stochastic_sample_codes=False,
                        sample_codebook_temp=1., straight_through=False, reinmax=False, sync_codebook=None,
                        sync_affine_param=False, ema_update=False, learnable_codebook=True, in_place_codebook_optimizer=None,
                        affine_param=False, affine_param_batch_decay=0.99, affine_param_codebook_decay=0.9, sync_update_v=0.)
        for k, v in unsupported.items():
            if k not in defaults:
                raise TypeError(f"VQ got an unexpected keyword argument {k!r}")
            if v != defaults[k]:
                raise NotImplementedError(f"idee_amd VQ: {k}={v!r} is outside the built path (default {defaults[k]!r})")
        codebook_dim = dim if codebook_dim is None else codebook_dim
        if heads != 1 or codebook_dim != dim:
            raise NotImplementedError("idee_amd VQ: heads != 1 or codebook_dim != dim (projections) is not built")
        self.dim, self.codebook_size, self.commitment_weight = dim, codebook_size, commitment_weight
        self._codebook = _Codebook(dim, codebook_size)

    @property
    def codebook(self):
        return self._codebook.embed[0]

    def indices_to_codes(self, indices):
        return self.codebook[indices]

    def forward(self, x, indices=None, mask=None, sample_codebook_temp=None, freeze_codebook=False):
        """x (b, n, dim) or (b, dim) -> (quantize, embed_ind, loss (1,))   (VQ.forward, VQ.py:897-1112)."""
        if indices is not None or mask is not None or sample_codebook_temp is not None or freeze_codebook:
            raise NotImplementedError("idee_amd VQ.forward: indices / mask / temperature / freeze_codebook are not built")
        return ops.VQNearest.apply(x, self.codebook, self.training, float(self.commitment_weight))
