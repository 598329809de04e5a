import sys; sys.path.insert(0,'.'); sys.path.insert(0,'tests')
import torch
from oracle import idee_oracle as O
from idee_amd.config import default_config
from idee_amd.models.build import VQ_model
torch.manual_seed(21)
kw = dict(encoder="Swin_3D", in_channels=2, in_channels_dynamic=2, en_embed_dim=[32, 32], en_depths=[2, 2], codebook_dim=32, cls_dim=32)
cfg = default_config(**kw); m = VQ_model(cfg)
g = torch.Generator().manual_seed(22)
with torch.no_grad():
    for n, p in m.named_parameters():
        p.copy_(torch.randn(p.shape, generator=g) * (0.5 / max(1.0, float(p[0].numel() if p.ndim > 1 else 1)) ** 0.5) + (0.0 if p.ndim > 1 else 0.05))
sd = {k: v.detach().clone() for k, v in m.state_dict().items()}
ocfg = O.default_cfg(**kw)
x = torch.randn((1, 2, 2, 8, 12, 16), generator=g)
with torch.no_grad():
    enc_o = O.swin_encoder(x, sd, ocfg)
    zo, yo, ano, zqo, lzo = O.vq_model_forward(x, sd, ocfg, training=False)
    m = m.cuda().eval()
    enc = m.encoder(x.cuda()).cpu()
    print('enc', (enc-enc_o).abs().max().item(), enc_o.abs().max().item())
    # stage by stage encoder
    from idee_amd import ops
    t = ops.PatchEmbedLN.apply(x.cuda(), m.encoder.s0_ds_w, m.encoder.s0_ds_b)
    to = O.patch_embed(x[:,0], sd, 'encoder.layers_var.0.0.downsample.').permute(0,2,3,4,1)
    print('pe', (t[:,0].cpu()-to).abs().max().item())
    z, y, an, zq, lz = m(x.cuda())
    print('zq', (zq.cpu()-zqo).abs().max().item(), 'an', (an.cpu()!=ano).sum().item())
    zc, yc = m.cls(zqo.cuda())
    print('cls on oracle zq: z', (zc.cpu()-zo).abs().max().item(), 'y', (torch.stack(yc).cpu()-torch.stack(yo)).abs().max().item())
