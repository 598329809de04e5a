"""Lookup-free quantisation on MI355X — drop-in for the reference's models/codebook/LFQ.py
(class LFQ, LFQ.py:68-307): sign quantisation of a dim -> log2(K) projection, straight-through
gradient, entropy + commitment auxiliary loss, all in one streaming HIP kernel per direction
(idee_lfq_fwd / idee_lfq_bwd).  Parameter names match the reference (project_in / project_out,
buffer `mask`), so state dicts interchange."""
from __future__ import annotations

from collections import namedtuple
from math import log2

import torch
import torch.nn as nn

from ... import ops

Return = namedtuple("Return", ["quantized", "indices", "entropy_aux_loss"])


class LFQ(nn.Module):
    def __init__(self, *, dim=16, dim_out=16, codebook_size=2, entropy_loss_weight=0.1, commitment_loss_weight=1.5,
                 diversity_gamma=1., straight_through_activation=None, num_codebooks=1, keep_num_codebooks_dim=None,
                 codebook_scale=1., frac_per_sample_entropy=1.):
        super().__init__()
        assert log2(codebook_size).is_integer(), "your codebook size must be a power of 2 for lookup free quantization"
        if codebook_scale != 1. or frac_per_sample_entropy != 1. or \
                (straight_through_activation is not None and not isinstance(straight_through_activation, nn.Identity)):
            raise NotImplementedError("idee_amd LFQ: codebook_scale != 1, frac_per_sample_entropy < 1 and a straight-through "
                                      "activation other than Identity are not built")
        codebook_dim = int(log2(codebook_size))
        codebook_dims = codebook_dim * num_codebooks               # LFQ.py:95-96
        has_projections = dim != codebook_dims                     # LFQ.py:98
        if has_projections and codebook_dims > 4:
            raise NotImplementedError("idee_amd LFQ: more than 4 projected code bits per token (num_codebooks * log2(codebook_size)) "
                                      "is not built")
        if not has_projections and (dim not in (16, 32) or codebook_dim not in (1, 2, 4)):
            raise NotImplementedError("idee_amd LFQ without projections: built for dim 16 / 32 and codebook_size 2 / 4 / 16 "
                                      f"(got dim {dim}, codebook_size {codebook_size})")
        keep_num_codebooks_dim = (num_codebooks > 1) if keep_num_codebooks_dim is None else keep_num_codebooks_dim      # :108
        assert not (num_codebooks > 1 and not keep_num_codebooks_dim)                                                  # :109
        self.dim, self.dim_out, self.codebook_dim, self.codebook_size = dim, dim_out, codebook_dim, codebook_size
        self.num_codebooks, self.keep_num_codebooks_dim, self.has_projections = num_codebooks, keep_num_codebooks_dim, has_projections
        self.project_in = nn.Linear(dim, codebook_dims) if has_projections else nn.Identity()      # :99-100
        self.project_out = nn.Linear(codebook_dims, dim) if has_projections else nn.Identity()
        self.diversity_gamma = diversity_gamma
        self.entropy_loss_weight = entropy_loss_weight
        self.commitment_loss_weight = commitment_loss_weight
        self.codebook_scale = codebook_scale
        self.register_buffer("mask", 2 ** torch.arange(codebook_dim - 1, -1, -1))
        self.register_buffer("zero", torch.tensor(0.), persistent=False)
        bits = ((torch.arange(codebook_size)[..., None].int() & self.mask) != 0).float()
        self.register_buffer("codebook", bits * 2 - 1, persistent=False)

    def split_indices(self, packed):
        """packed (...) int64 as the kernel writes it (the per-codebook indices, most significant codebook first, codebook_dim bits
        each) -> (..., num_codebooks) as LFQ.forward returns them with keep_num_codebooks_dim (LFQ.py:234, :293-298)."""
        c, d = self.num_codebooks, self.codebook_dim
        shifts = torch.arange(c - 1, -1, -1, device=packed.device) * d
        return (packed.unsqueeze(-1) >> shifts) & (self.codebook_size - 1)

    def indices_to_codes(self, indices, project_out=True):
        """LFQ.indices_to_codes (LFQ.py:152-181): e.g. the `indices_to_codes(LongTensor([0]))` call of
        train_synthetic.py:193-194 -> (1, dim); index maps with 3 (+ keep_num_codebooks_dim) or more dims return channels-first codes.
        A table lookup of K codes: plain tensor indexing, not a kernel."""
        is_img_or_video = indices.ndim >= (3 + int(self.keep_num_codebooks_dim))
        if not self.keep_num_codebooks_dim:
            indices = indices.unsqueeze(-1)                                    # '... -> ... 1'
        codes = self.codebook[indices]                                         # (..., c, d)
        codes = codes.reshape(*codes.shape[:-2], -1)                           # '... c d -> ... (c d)'
        if project_out:
            codes = self.project_out(codes)
        if is_img_or_video:      # image / video index maps: codes come back channels-first, 'b ... d -> b d ...' (:177-179)
            codes = codes.permute(0, codes.ndim - 1, *range(1, codes.ndim - 1))
        return codes

    def quantize_tokens(self, z):
        """z (..., dim) channels-last tokens -> (z_q (..., dim), indices int64 (...), aux loss ())."""
        return self.quantize_tokens_signs(z, False)[:3]

    def quantize_tokens_signs(self, z, want_signs=True):
        """-> (z_q, indices, aux loss, signs (..., codebook_dim) | None): `signs` are the +-1 bits before project_out,
        carrying the same straight-through gradient as z_q (for the classifier's rank-d input path)."""
        assert z.shape[-1] == self.dim, f"expected dimension of {self.dim} but received {z.shape[-1]}"
        if not self.has_projections:     # the tokens are the code bits: z_q = sign(z) is its own sign field
            zq, idx, aux = ops.LFQQuantizeNoProj.apply(z, self.codebook_size, self.training, float(self.entropy_loss_weight),
                                                       float(self.diversity_gamma), float(self.commitment_loss_weight))
            return zq, idx, aux, (zq if want_signs else None)
        return ops.LFQQuantize.apply(z, self.project_in.weight, self.project_in.bias, self.project_out.weight,
                                     self.project_out.bias, self.codebook_size, self.training,
                                     float(self.entropy_loss_weight), float(self.diversity_gamma),
                                     float(self.commitment_loss_weight), want_signs, self.num_codebooks)

    def forward(self, x, inv_temperature=100., return_loss_breakdown=False, mask=None):
        """x (b, n, dim) as models/build.py:150-151 passes it (or (b, dim, ...) image/video)."""
        if inv_temperature != 100. or return_loss_breakdown or mask is not None:
            raise NotImplementedError("idee_amd LFQ.forward: non-default inv_temperature / breakdown / mask are not built")
        if x.ndim >= 4:   # (b, d, ...) -> tokens last, and back (LFQ.py:205-207, :289-291)
            perm = (0,) + tuple(range(2, x.ndim)) + (1,)
            zq, idx, aux = self.quantize_tokens(x.permute(*perm).contiguous())
            inv = (0, x.ndim - 1) + tuple(range(1, x.ndim - 1))
            return Return(zq.permute(*inv), self.split_indices(idx) if self.keep_num_codebooks_dim else idx, aux)
        zq, idx, aux = self.quantize_tokens(x.contiguous())
        return Return(zq, self.split_indices(idx) if self.keep_num_codebooks_dim else idx, aux)
