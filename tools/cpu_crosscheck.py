"""in the AUTHORING container, time the true reference and the oracle (CPU restatement) under the
same protocol — one clip V=6, T=8, 200x200, full training step (forward + reference losses + backward + Adam),
1 warm-up + 2 timed steps each, torch.set_num_threads(cores) — to show that the restatement bench.py times as
`cpu_baseline` on the GPU box is a fair stand-in for the reference's CPU path.
Reads /root/reference: runs only here, never on the GPU box (nothing imports this file)."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import torch  # noqa: E402
import make_fixtures as MF  # noqa: E402  (installs the in-memory timm stub and imports the reference)
from oracle import idee_oracle as O  # noqa: E402

cores = len(os.sched_getaffinity(0))
torch.set_num_threads(cores)
V, T, H, W = 6, 8, 200, 200
cfg = MF.ref_cfg(encoder="Swin_3D", in_channels=1, in_channels_dynamic=V)
torch.manual_seed(0)
model = MF.VQ_model(cfg).train()
g = torch.Generator().manual_seed(1234)
x = torch.randn((1, V, 1, T, H, W), generator=g).clamp_(-10, 10)
me = (torch.rand((1, H, W), generator=g) < 0.10).float()
ml = (torch.rand((1, H, W), generator=g) < 0.20).float()
sd0 = {k: v.detach().clone() for k, v in model.state_dict().items()}

# ---- the reference ----
crit = MF.ref_losses.BCE_loss_synthetic()
crit_an = MF.ref_losses.Anomaly_L1_loss_synthetic(n_dynamic=V, delta_t=cfg.delta_t, dim=cfg.en_embed_dim[-1])
opt = torch.optim.Adam(model.parameters(), lr=cfg.lr, weight_decay=cfg.weight_decay, betas=(cfg.beta1, cfg.beta2))
t_ref = []
for step in range(3):
    t0 = time.perf_counter()
    opt.zero_grad(set_to_none=True)
    pred, pred_y, anomaly, z_q, loss_z_q = model(x)
    tgt = me.unsqueeze(1)
    loss = crit(pred, tgt) + crit_an(z_q, ml, model.vq.indices_to_codes(torch.Tensor([0]).long()).clone().detach()) * cfg.lambda_anomaly \
        + sum(crit(p, tgt) for p in pred_y) + loss_z_q
    loss.backward()
    opt.step()
    t_ref.append(time.perf_counter() - t0)
ref_loss = float(loss)

# ---- the oracle, same initial state ----
ocfg = O.default_cfg(encoder="Swin_3D", in_channels_dynamic=V, in_channels=1)
sd = {k: v.clone() for k, v in sd0.items()}
names = [k for k, v in sd.items() if v.is_floating_point()]
state = {k: (torch.zeros_like(sd[k]), torch.zeros_like(sd[k])) for k in names}
t_or = []
for step in range(1, 4):
    t0 = time.perf_counter()
    for k in names:
        sd[k] = sd[k].detach().requires_grad_()
    outs = O.vq_model_forward(x, sd, ocfg, training=True)
    total, _ = O.train_losses_synthetic(outs, me, ml, sd, ocfg)
    total.backward()
    with torch.no_grad():
        for k in names:
            p, m, v = O.adam_step(sd[k].detach(), sd[k].grad, state[k][0], state[k][1], step, cfg.lr, cfg.beta1, cfg.beta2, 1e-8,
                                  cfg.weight_decay)
            state[k] = (m, v)
            sd[k] = p
    t_or.append(time.perf_counter() - t0)
print(f"cores {cores}  reference step {sum(t_ref[1:]) / 2:.3f} s ({2 / sum(t_ref[1:]):.3f} samples/s)  "
      f"oracle step {sum(t_or[1:]) / 2:.3f} s ({2 / sum(t_or[1:]):.3f} samples/s)  third-step loss: reference {ref_loss:.6f} oracle {float(total):.6f}")
