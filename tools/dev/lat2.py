import os, sys, time, importlib
import numpy as np, torch
sys.path.insert(0, "/root/repo/tests"); sys.path.insert(0, "/root/repo")
import helpers as H
zkp = H.zkp
ctx = zkp.Context(0); dev = torch.device("cuda", 0)
def rnd(shape, seed):
    g = torch.Generator(device=dev); g.manual_seed(seed)
    return torch.randint(-2**31, 2**31 - 1, shape, dtype=torch.int32, device=dev, generator=g)
B = 16
N = rnd((B, 64), 2); N[:, 0] |= 1; N[:, -1] |= -2**31
g_ = rnd((B, 64), 3); g_[:, -1] &= 0x3FFFFFFF
ni = rnd((B, 64), 4); ni[:, -1] &= 0x3FFFFFFF
x = rnd((B, 64), 5); x[:, -1] &= 0x3FFFFFFF
y = rnd((B, 24), 6); y[:, 17:] = 0
sg = rnd((B, 11, 64), 7); sg[:, :, -1] &= 0x3FFFFFFF
v = torch.zeros(B, dtype=torch.uint8, device=dev)
for rep in range(5):
    ctx.dlog_verify(2048, 768, B, N, g_, ni, x, y, v); ctx.synchronize()
    ctx.correct_key_ni_verify(2048, B, N, sg, b"KZen", v); ctx.synchronize()
print(ctx.last_geometry())
