"""The actor's output layer + loss + backward (bg_actor_head_partial) and its forward-only form alone on the GPU, SUSTAINED rate (100 untimed launches first),
beside a 50 MB -> 50 MB copy: is the head kernel itself slow, or only where it runs (beside the critic's backward GEMMs)?  python tools/head_time.py"""
import os, sys
sys.path.insert(0, os.getcwd())
import torch
from booster_gym_amd import _lib
from booster_gym_amd.utils.utils import actor_head_forward, actor_head_loss_backward, head_scratch
dev = "cuda:0"
B = 98304
torch.manual_seed(0)
h = torch.randn(B, 128, device=dev); W = torch.randn(12, 128, device=dev) * 0.1; b = torch.zeros(12, device=dev)
mu = torch.zeros(B, 12, device=dev)
logstd = torch.full((12,), -2.0, device=dev); act = torch.randn(B, 12, device=dev) * 0.1
old_logp = torch.zeros(B, device=dev); adv = torch.randn(B, device=dev)
adv_sums = torch.tensor([float(adv.sum()), float((adv * adv).sum()), float(B)], dtype=torch.float64, device=dev)
g = torch.zeros(B, 128, device=dev); gW = torch.zeros(12, 128, device=dev); gb = torch.zeros(12, device=dev); gbh = torch.zeros(128, device=dev)
gls = torch.zeros(12, dtype=torch.float64, device=dev); st = torch.zeros(5, dtype=torch.float64, device=dev); sc = head_scratch(dev)
def t(fn, n=200):
    for _ in range(100): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
fwd = lambda: actor_head_forward(h, W, b, mu)
fin = _lib.ReduceProblem()
full = lambda: actor_head_loss_backward(h, W, b, logstd, act, mu, logstd, old_logp, adv, adv_sums, 0.2, 0.0, -0.01, g, gW, gb, gbh, gls, st, sc, finish=fin)
print("actor head forward only (mode 0): %.1f us" % t(fwd))
print("actor head forward + loss + backward (partial form): %.1f us" % t(full))
cp = lambda: g.copy_(h)
print("copy 50 MB -> 50 MB (torch): %.1f us" % t(cp))
