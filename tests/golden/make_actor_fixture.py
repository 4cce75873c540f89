"""Dump the weight tensors of the reference's trained TorchScript actor to a plain .npz.

Source: /root/reference/deploy/models/T1.pt (47->256->128->128->12, ELU; utils/model.py:18-26).
Only numbers are stored; the TorchScript archive (which embeds serialized code) is not copied.
Run in the build container:  python tests/golden/make_actor_fixture.py
"""
import numpy as np
import torch

m = torch.jit.load("/root/reference/deploy/models/T1.pt", map_location="cpu")
sd = {k: v.detach().numpy() for k, v in m.state_dict().items()}
for k, v in sd.items():
    print(k, v.shape)
np.savez_compressed("tests/golden/t1_actor.npz", **sd)
