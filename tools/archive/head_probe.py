"""Time the fused head kernels at the training shape (B = 98,304), HIP events on the launch stream."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from booster_gym_amd.utils.utils import actor_head_forward, actor_head_loss_backward, critic_head_backward, critic_head_forward, head_scratch
B, A, dev = 98304, 12, "cuda:0"
g = torch.Generator(device="cpu").manual_seed(0)
h = torch.nn.functional.elu(torch.randn(B, 128, generator=g)).to(dev)
W = (torch.randn(A, 128, generator=g) * 0.1).to(dev); b = torch.zeros(A, device=dev)
logstd = torch.full((A,), -2.0, device=dev); mu0 = h @ W.t()
actions = mu0 + 0.135 * torch.randn(B, A, generator=g).to(dev)
old_logp = (-0.5 * ((actions - mu0) / logstd.exp()) ** 2 - logstd - 0.9189385332046727).sum(-1)
adv = torch.randn(B, generator=g).to(dev)
st3 = torch.stack([adv.double().sum(), (adv.double() ** 2).sum(), torch.tensor(float(B), dtype=torch.float64, device=dev)])
gh = torch.empty(B, 128, device=dev); dW = torch.empty(A, 128, device=dev); db = torch.empty(A, device=dev); dbh = torch.empty(128, device=dev)
gls = torch.zeros(A, dtype=torch.float64, device=dev); st = torch.zeros(5, dtype=torch.float64, device=dev); sc = head_scratch(dev)
vals, rets = torch.randn(B, device=dev), torch.randn(B, device=dev); mu = torch.empty(B, A, device=dev); v = torch.empty(B, device=dev)
def timeit(f, n=30):
    for _ in range(3): f()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
print("actor head forward only      %7.1f us" % timeit(lambda: actor_head_forward(h, W, b, mu)))
print("actor head fwd+loss+bwd      %7.1f us" % timeit(lambda: actor_head_loss_backward(h, W, b, logstd, actions, mu0, logstd, old_logp, adv, st3, 0.2, 1.0, -0.01, gh, dW, db, dbh, gls, st, sc)))
print("critic head forward          %7.1f us" % timeit(lambda: critic_head_forward(h, W[:1].contiguous(), b[:1].contiguous(), v)))
print("critic head backward         %7.1f us" % timeit(lambda: critic_head_backward(h, W[:1].contiguous(), vals, rets, gh, dW[:1], db[:1], dbh, st, sc)))
