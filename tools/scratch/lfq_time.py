import torch, time
from idee_amd import ops, _lib as L
torch.manual_seed(0)
dev = "cuda"
N, V, T, H, W, E, K = 4, 6, 8, 200, 200, 16, 2
z = torch.randn(N, V, T, H, W, E, device=dev)
w_in = torch.randn(1, E, device=dev) * 0.3; b_in = torch.zeros(1, device=dev)
w_out = torch.randn(E, 1, device=dev) * 0.3; b_out = torch.zeros(E, device=dev)
zq, idx, loss, signs = ops.LFQQuantize.apply(z.requires_grad_(), w_in.requires_grad_(), b_in.requires_grad_(), w_out.requires_grad_(), b_out.requires_grad_(), K, True, 0.1, 0.1, 3.0, True)
dzq = torch.randn_like(zq); dsg = torch.randn_like(signs)
def run(use_dzq, use_ds, n=20):
    gs = []
    for _ in range(3):
        torch.autograd.grad([zq, loss, signs] if use_ds else [zq, loss], [z], [dzq if use_dzq else torch.zeros_like(zq), torch.ones_like(loss)] + ([dsg] if use_ds else []), retain_graph=True)
    torch.cuda.synchronize()
    log = L.set_event_log(["idee_lfq_bwd"])
    for _ in range(n):
        torch.autograd.grad([zq, loss, signs] if use_ds else [zq, loss], [z], [dzq if use_dzq else torch.zeros_like(zq), torch.ones_like(loss)] + ([dsg] if use_ds else []), retain_graph=True)
    torch.cuda.synchronize()
    ts = [a.elapsed_time(b) for a, b, _ in log["idee_lfq_bwd"]]
    L.set_event_log(None)
    return sum(ts) / len(ts)
print("dense dzq + dsigns", run(True, True))
# None path: call backward with zq grad None
def run_none(n=20):
    for _ in range(3):
        torch.autograd.grad([loss, signs], [z], [torch.ones_like(loss), dsg], retain_graph=True)
    torch.cuda.synchronize()
    log = L.set_event_log(["idee_lfq_bwd"])
    for _ in range(n):
        torch.autograd.grad([loss, signs], [z], [torch.ones_like(loss), dsg], retain_graph=True)
    torch.cuda.synchronize()
    ts = [a.elapsed_time(b) for a, b, _ in log["idee_lfq_bwd"]]
    L.set_event_log(None)
    return sum(ts) / len(ts)
print("no dzq + dsigns", run_none())
