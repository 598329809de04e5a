"""Vector quantiser on MI355X — the Euclidean-codebook paths of the reference's models/codebook/VQ.py (class VQ :737-1112
with EuclideanCodebook :273-555): nearest-neighbour lookup with straight-through gradient and commitment mse, and for the
codebook either gradient descent (`learnable_codebook=True`, the reference default) or the EMA update
(`ema_update=True`, :519-545) with optional k-means initialisation (`kmeans_init`, :213-253, :354-377) and dead-code expiry
(`threshold_ema_dead_code`, :449-475); under data parallelism the cluster sizes and per-code sums are all-reduced where the
reference calls `all_reduce_fn` (:528, :533, :236, :244; `sync_codebook`).  The reference's models/build.py:87 never
constructs this class (LFQ is hard-wired); it is provided because the path's brief names it.
Outside the built path (they raise NotImplementedError): gumbel / reinmax sampling, cosine similarity, multi-head, affine
parameters, orthogonal regularisation, masks, projections (codebook_dim != dim).  State-dict keys match the reference
(`_codebook.embed`, `_codebook.initted`, `_codebook.cluster_size`, `_codebook.embed_avg`)."""
from __future__ import annotations

import ctypes as C

import torch
import torch.nn as nn

from ... import _lib as L
from ... import ops, parallel
from ..._lib import ptr


class _Codebook(nn.Module):
    def __init__(self, dim, codebook_size, learnable, kmeans_init):
        super().__init__()
        embed = torch.zeros(1, codebook_size, dim)
        if not kmeans_init:
            nn.init.kaiming_uniform_(embed)                   # uniform_init, VQ.py:68-71 (torch.zeros under kmeans_init, :297)
        if learnable:
            self.embed = nn.Parameter(embed)
        else:
            self.register_buffer("embed", embed)              # :326-329
        self.register_buffer("initted", torch.Tensor([not kmeans_init]))
        self.register_buffer("cluster_size", torch.zeros(1, codebook_size))
        self.register_buffer("embed_avg", embed.clone())


def _cluster_stats(x2, idx, K):
    """x2 (M,E), idx int64 (M) -> counts (K), sums (K,E), all-reduced across data-parallel ranks."""
    M, E = x2.shape
    counts = torch.zeros(K, device=x2.device, dtype=torch.float32)
    sums = torch.zeros((K, E), device=x2.device, dtype=torch.float32)
    L.call("idee_vq_cluster_stats", ptr(x2), ptr(idx), ptr(counts), ptr(sums), M, E, K, L.stream())
    return counts, sums


class VQ(nn.Module):
    def __init__(self, dim=16, codebook_size=2, codebook_dim=16, heads=1, commitment_weight=1., decay=0.8, eps=1e-5,
                 kmeans_init=False, kmeans_iters=10, sync_kmeans=True, threshold_ema_dead_code=0, ema_update=False,
                 learnable_codebook=True, sync_codebook=None, **unsupported):
        super().__init__()
        defaults = dict(separate_codebook_per_head=False, freeze_codebook=False, use_cosine_sim=False, channel_last=True,
                        accept_image_fmap=False, commitment_use_cross_entropy_loss=False, orthogonal_reg_weight=0.,
                        orthogonal_reg_active_codes_only=False, orthogonal_reg_max_codes=None, stochastic_sample_codes=False,
                        sample_codebook_temp=1., straight_through=False, reinmax=False, sync_affine_param=False,
                        in_place_codebook_optimizer=None, affine_param=False, affine_param_batch_decay=0.99,
                        affine_param_codebook_decay=0.9, sync_update_v=0.)
        for k, v in unsupported.items():
            if k not in defaults:
                raise TypeError(f"VQ got an unexpected keyword argument {k!r}")
            if v != defaults[k]:
                raise NotImplementedError(f"idee_amd VQ: {k}={v!r} is outside the built path (default {defaults[k]!r})")
        codebook_dim = dim if codebook_dim is None else codebook_dim
        if heads != 1 or codebook_dim != dim:
            raise NotImplementedError("idee_amd VQ: heads != 1 or codebook_dim != dim (projections) is not built")
        assert not (ema_update and learnable_codebook), 'learnable codebook not compatible with EMA update'      # VQ.py:799
        self.dim, self.codebook_size, self.commitment_weight = dim, codebook_size, commitment_weight
        self.decay, self.eps, self.ema_update, self.learnable_codebook = decay, eps, ema_update, learnable_codebook
        self.kmeans_iters, self.sync_kmeans = kmeans_iters, sync_kmeans
        self.threshold_ema_dead_code = threshold_ema_dead_code
        self.reset_cluster_size = threshold_ema_dead_code            # default(reset_cluster_size, threshold), :308
        self.sync_codebook = sync_codebook                            # None: follow torch.distributed (VQ.py:815-816)
        self._codebook = _Codebook(dim, codebook_size, learnable_codebook, kmeans_init)
        # which batch samples seed k-means / replace dead codes: (number of samples, how many) -> int64 indices.  The
        # reference draws torch.randperm(n)[:num] (sample_vectors, :129-137); tests substitute a deterministic choice.
        self.sample_indices = lambda n, num, device: (torch.randperm(n, device=device)[:num] if n >= num
                                                      else torch.randint(0, n, (num,), device=device))

    @property
    def codebook(self):
        return self._codebook.embed[0]

    def indices_to_codes(self, indices):
        return self.codebook[indices]

    def _sync(self):
        on = parallel.world_size() > 1 if self.sync_codebook is None else bool(self.sync_codebook)
        return on and parallel.world_size() > 1

    @torch.no_grad()
    def _kmeans_init(self, x2):
        """init_embed_ (:354-377) with kmeans (:213-253).  x2 (M, E)."""
        cb = self._codebook
        K, E = self.codebook_size, self.dim
        M = x2.shape[0]
        means = x2[self.sample_indices(M, K, x2.device)].contiguous()
        sync = self._sync() and self.sync_kmeans
        if sync:
            torch.distributed.broadcast(means, src=0)
        q = torch.empty_like(x2)
        idx = torch.empty(M, device=x2.device, dtype=torch.int64)
        bins = None
        for _ in range(self.kmeans_iters):
            L.call("idee_vq_nn_fwd", ptr(x2), ptr(means), ptr(q), ptr(idx), None, None, M, E, K, 0, 0.0, L.stream())
            bins, sums = _cluster_stats(x2, idx, K)
            if sync:
                parallel.all_reduce_sum_(bins)
                parallel.all_reduce_sum_(sums)
            L.call("idee_vq_kmeans_update", ptr(means), ptr(bins), ptr(sums), K, E, L.stream())
        cb.embed.data[0].copy_(means)
        cb.embed_avg.data[0].copy_(means * bins.unsqueeze(1))
        cb.cluster_size.data[0].copy_(bins)
        cb.initted.data.copy_(torch.Tensor([True]))

    @torch.no_grad()
    def _ema_step(self, x2, idx):
        """The training-time codebook update of EuclideanCodebook.forward (:519-545) after the lookup."""
        cb = self._codebook
        K, E = self.codebook_size, self.dim
        counts, sums = _cluster_stats(x2, idx.reshape(-1), K)
        if self._sync():
            parallel.all_reduce_sum_(counts)
            parallel.all_reduce_sum_(sums)
        L.call("idee_vq_ema_update", ptr(cb.cluster_size), ptr(cb.embed_avg), ptr(cb.embed), ptr(counts), ptr(sums), K, E,
               float(self.decay), float(self.eps), L.stream())
        if self.threshold_ema_dead_code > 0:                           # expire_codes_, :463-475
            cand = self.sample_indices(x2.shape[0], K, x2.device).to(torch.int64).contiguous()
            L.call("idee_vq_expire_codes", ptr(cb.embed), ptr(cb.cluster_size), ptr(cb.embed_avg), ptr(x2), ptr(cand), K, E,
                   float(self.threshold_ema_dead_code), float(self.reset_cluster_size), None, L.stream())
            if self._sync():                                           # every replica must hold the same replacements
                for t in (cb.embed, cb.cluster_size, cb.embed_avg):
                    torch.distributed.broadcast(t, src=0)

    def forward(self, x, indices=None, mask=None, sample_codebook_temp=None, freeze_codebook=False):
        """x (b, n, dim) or (b, dim) -> (quantize, embed_ind, loss (1,))   (VQ.forward, VQ.py:897-1112)."""
        if indices is not None or mask is not None or sample_codebook_temp is not None:
            raise NotImplementedError("idee_amd VQ.forward: indices / mask / temperature are not built")
        cb = self._codebook
        x2 = ops._f32c(x.detach()).reshape(-1, self.dim)
        if not bool(cb.initted):
            self._kmeans_init(x2)
        embed = cb.embed[0] if (self.learnable_codebook and not freeze_codebook) else cb.embed[0].detach()
        # the lookup (and the value of `quantize`) uses the codebook BEFORE this step's EMA update, as the reference does
        out = ops.VQNearest.apply(x, embed.clone() if self.ema_update else embed, self.training, float(self.commitment_weight))
        if self.training and self.ema_update and not freeze_codebook:
            self._ema_step(x2, out[1])
        return out
