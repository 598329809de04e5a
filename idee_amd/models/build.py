"""Model assembly on MI355X — drop-in for the reference's models/build.py: same `VQ_model(config)`
constructor, `.encoder / .vq / .cls` attributes, `forward(x_d)` return tuple, `import_class`
plugin rule and state-dict keys (models/build.py:17-159 of the reference).

Internally activations stay channels-last tokens (N,V,T,H,W,C) from the encoder through LFQ to
the classifier, so the two (N, V*T*H*W, C) permute copies of build.py:150,153 do not exist; the
returned `z_q` is the permuted VIEW (N,V,C,T,H,W) the reference also returns."""
from __future__ import annotations

import importlib
import os

import torch
import torch.nn as nn


def import_class(en_de, name):
    """Import class `name` from module models.<en_de>.<name> of THIS package (build.py:17-20)."""
    module = importlib.import_module(f"{__package__}.{en_de}.{name}")
    return getattr(module, name)


class VQ_model(nn.Module):
    """Encoder -> LFQ -> classifier.  `config` is the argparse namespace of config.py."""

    def __init__(self, config):
        super().__init__()
        if config.encoder == "CNN_3D":
            self.encoder = import_class('encoder', config.encoder)(
                in_vars=config.in_channels_dynamic, in_channels=config.in_channels, out_channels=config.en_embed_dim,
                drop_path_rate=config.en_drop_path_rate, drop_rate=config.en_drop_rate)
        elif config.encoder == "Swin_3D":
            self.encoder = import_class('encoder', config.encoder)(
                in_vars=config.in_channels_dynamic, in_chans=config.in_channels, embed_dim=config.en_embed_dim,
                window_size=config.en_window_size, depths=config.en_depths, num_heads=config.en_n_heads,
                mlp_ratio=config.en_mlp_ratio, drop_rate=config.en_drop_rate, attn_drop_rate=config.en_attn_drop_rate,
                drop_path_rate=config.en_drop_path_rate, qkv_bias=config.en_qkv_bias, qk_scale=config.en_qk_scale,
                patch_size=config.en_patch_size, patch_norm=config.en_patch_norm, use_checkpoint=config.en_use_checkpoint)
        elif config.encoder == "Mamba":
            raise NotImplementedError("Encoder Mamba is outside the MI355X hot path (its arithmetic lives in the CUDA-only "
                                      "mamba_ssm package); use encoder='Swin_3D' or 'CNN_3D'")
        else:
            raise NotImplementedError(f"Encoder {config.encoder} not implemented")

        self.cls = import_class('classifier', 'CNN_3D')(in_var=config.in_channels_dynamic, embed_dim=config.codebook_dim,
                                                        dim=config.cls_dim, drop_rate=config.cls_drop_rate)
        # The reference hard-wires LFQ here (build.py:87; its `codebook` flag is never read).  `config.codebook_impl = "VQ"` (an
        # attribute the caller sets; not a flag of config.py) puts the nearest-neighbour codebook of models/codebook/VQ.py in its
        # place — BASELINE config #4's "+ VQ codebook" variant: dim = codebook_dim, codebook_size codes, learnable codebook,
        # commitment weight lambda_commitment.
        self.codebook_impl = getattr(config, "codebook_impl", "LFQ")
        if self.codebook_impl == "VQ":
            self.vq = import_class('codebook', 'VQ')(dim=config.codebook_dim, codebook_size=config.codebook_size,
                                                     codebook_dim=config.codebook_dim, commitment_weight=config.lambda_commitment)
        elif self.codebook_impl == "LFQ":
            self.vq = import_class('codebook', 'LFQ')(dim=config.codebook_dim, codebook_size=config.codebook_size,
                                                      entropy_loss_weight=config.lambda_entropy,
                                                      diversity_gamma=config.diversity_gamma,
                                                      commitment_loss_weight=config.lambda_commitment)
        else:
            raise NotImplementedError(f"codebook_impl {self.codebook_impl!r}: 'LFQ' (the reference's wiring) or 'VQ'")
        self.pretrained = config.en_de_pretrained
        # `config.act_dtype = "bf16"` (an attribute the caller sets; default f32 = the reference's arithmetic): the token tensors
        # between the fused units (patch embedding -> Swin blocks -> 3^3 convs -> classifier) and their gradients are stored as
        # bf16 and every GEMM-shaped kernel runs on the bf16 MFMA with f32 accumulation; parameters, LFQ's projections, logits,
        # losses and the optimizer stay f32 (BASELINE config #4)
        ad = getattr(config, "act_dtype", "f32")
        if ad not in ("f32", "bf16"):
            raise ValueError(f"config.act_dtype must be 'f32' or 'bf16', got {ad!r}")
        self.act_dtype = torch.bfloat16 if ad == "bf16" else torch.float32
        # bf16 storage with the CNN_3D encoder: its LayerNorm / ReLU / residual units (idee_ln_relu_res_*) are f32-storage kernels, so
        # the encoder keeps f32 tokens INSIDE and hands bf16 tokens on (codebook, classifier and their gradients in bf16 storage);
        # element-wise dropout inside a Swin block (en_drop_rate / en_attn_drop_rate > 0) is built in the exact-f32 block kernels: under
        # bf16 storage such a block widens its tokens to f32 for the block and narrows them again (models/encoder/Swin_3D.py)
        self.encoder.act_dtype = self.act_dtype if config.encoder == "Swin_3D" else torch.float32
        # IDEE_SIGN_FIELD=0 runs the classifier on z_q itself, convolution for convolution as build.py:157
        self.sign_field_classifier = os.environ.get("IDEE_SIGN_FIELD", "1") != "0" and self.codebook_impl == "LFQ"
        self._init_weights()

    def _init_weights(self, init_type='normal', gain=.02):
        """Conv/Linear weights ~ N(0.02, gain), biases 0, LayerNorm affine (1, 0); relative-position
        tables keep the encoder's trunc-normal (build.py:96-118)."""
        if init_type != 'normal':
            raise NotImplementedError('initialization method [%s] is not implemented' % init_type)
        with torch.no_grad():
            for mod in (self.encoder, self.cls):
                ws, bs, _tables = mod.weight_names()
                for n in ws:
                    getattr(mod, n).normal_(0.02, gain)
                for n in bs:
                    getattr(mod, n).zero_()
                norm_w, norm_b = getattr(mod, "norm_names", lambda: ([], []))()
                for n in norm_w:
                    getattr(mod, n).fill_(1.0)
                for n in norm_b:
                    getattr(mod, n).zero_()
            if self.codebook_impl == "LFQ":
                for lin in (self.vq.project_in, self.vq.project_out):
                    lin.weight.normal_(0.02, gain)
                    lin.bias.zero_()
        if self.pretrained:
            print('initialize weights from pretrained model {} ...'.format(self.pretrained))
            checkpoint = torch.load(self.pretrained, map_location='cpu')
            state_dict = {k.replace('module.', ''): v for k, v in checkpoint['model_state_dict'].items()}
            self.load_state_dict(state_dict, strict=True)

    def forward(self, x_d):
        """x_d (N,V,C,T,H,W) -> z (N,1,H,W), y [V x (N,1,H,W)], anomaly int64 (N,V,T,H,W),
        z_q (N,V,C,T,H,W), loss_z_q (1,)   — build.py:130-159."""
        z = self.encoder.forward_tokens(x_d)                 # (N,V,T,H,W,E)
        if z.dtype != self.act_dtype:                        # (CNN_3D under bf16 storage: f32 inside the encoder, see __init__)
            z = z.to(self.act_dtype)
        if self.sign_field_classifier and self.cls.dim // self.cls.in_var % 16 == 0:
            # z_q has rank codebook_dim per variable: the classifier's first convolutions read the +-1 code bits
            # with weights folded through project_out (exact refactoring, DESIGN.md "rank-d classifier input")
            z_q, anomaly, loss_z_q, signs = self.vq.quantize_tokens_signs(z)
            zc, y = self.cls.forward_signs(signs, self.vq.project_out, self.act_dtype)
            self.last_signs = signs if self.training else None   # the train step's anomaly loss reads the bits too (train.py)
        elif self.codebook_impl == "VQ":
            # nearest-neighbour codebook variant: an f32 lookup (like LFQ's f32 projection) on the widened tokens; `anomaly` is
            # the code index, z_q the selected code vectors with the straight-through gradient (VQ.py:897-1112)
            N, V, T, H, W, E = z.shape
            z_q, anomaly, loss_z_q = self.vq(z.float().reshape(N * V, T * H * W, E))
            z_q, anomaly, loss_z_q = z_q.reshape(N, V, T, H, W, E), anomaly.reshape(N, V, T, H, W), loss_z_q.reshape(())
            zc, y = self.cls.forward_tokens(z_q.to(self.act_dtype))
            self.last_signs = None
        else:
            z_q, anomaly, loss_z_q = self.vq.quantize_tokens(z)
            zc, y = self.cls.forward_tokens(z_q)
            self.last_signs = None
        return zc, y, anomaly, z_q.permute(0, 1, 5, 2, 3, 4), loss_z_q.unsqueeze(0)
