"""CPU study: how much error does Winograd F(2x2,3x3) with fp16 operands (input transform rounded to
fp16 after each of its two stages, transformed weights rounded to fp16, float32 accumulation and
output transform) add to the conv stack, against the direct convolution on the same fp16-rounded
weights and activations?  Prints per-layer max error / layer max and the descriptor cosine."""
import sys, os
import numpy as np
import torch
import torch.nn.functional as F
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "face-recognition_amd")]
from frhot import synth
from frhot.weights import synthetic_weights
from oracle import preprocess as op
from oracle.vggface import CONV_LAYERS

h = lambda t: t.half().float()
BT = torch.tensor([[1, 0, -1, 0], [0, 1, 1, 0], [0, -1, 1, 0], [0, 1, 0, -1]], dtype=torch.float32)
G = torch.tensor([[1, 0, 0], [.5, .5, .5], [.5, -.5, .5], [0, 0, 1]], dtype=torch.float32)
AT = torch.tensor([[1, 1, 1, 0], [0, 1, -1, -1]], dtype=torch.float32)


def wino(x, w, b):
    """x [B,C,H,W] fp16-valued float32, w [K,C,3,3] fp16-valued -> [B,K,H,W] float32 (before ReLU)"""
    B, C, H, W = x.shape
    xp = F.pad(x, (1, 1, 1, 1))
    d = xp.unfold(2, 4, 2).unfold(3, 4, 2)              # [B,C,H/2,W/2,4,4]
    t = h(torch.einsum("ij,bcyxjk->bcyxik", BT, d))     # rows, rounded
    V = h(torch.einsum("bcyxik,lk->bcyxil", t, BT))     # cols, rounded
    U = h(torch.einsum("ij,kcjl,ml->kcim", G, w, G))    # [K,C,4,4]
    M = torch.einsum("bcyxim,kcim->bkyxim", V, U)
    Y = torch.einsum("ij,bkyxjm,lm->bkyxil", AT, M, AT)  # [B,K,H/2,W/2,2,2]
    Y = Y.permute(0, 1, 2, 4, 3, 5).reshape(B, -1, H, W)
    return Y + b[None, :, None, None], V.abs().max().item(), U.abs().max().item()


def main():
    s = 1.0 / 16.0
    torch.set_num_threads(8)
    wts = synthetic_weights(seed=1)
    wq = {k: (v.astype(np.float16).astype(np.float32) if k.endswith(".w") else v) for k, v in wts.items()}
    imgs = synth.crops(seed=21, n=2, lo=112, hi=448)
    x0 = torch.from_numpy(np.stack([op.preprocess(im, apply_gamma=True) for im in imgs])).permute(0, 3, 1, 2)
    xd = h(x0 * s) ; xw = xd.clone()                     # scaled fp16 activations, as the device stores them
    with torch.no_grad():
        for li, (name, cin, cout, pool) in enumerate(CONV_LAYERS):
            w = torch.from_numpy(wq[name + ".w"]).permute(0, 3, 1, 2).contiguous()
            b = torch.from_numpy(wq[name + ".b"]) * s
            yd = F.conv2d(xd, w, b, padding=1)
            if li == 0:
                yw, vm, um = F.conv2d(xw, w, b, padding=1), 0, 0
            else:
                yw, vm, um = wino(xw, w, b)
            yd, yw = F.relu(yd), F.relu(yw)
            if pool:
                yd, yw = F.max_pool2d(yd, 2, 2), F.max_pool2d(yw, 2, 2)
            xd, xw = h(yd), h(yw)
            e = (xd - xw).abs().max().item() / xd.abs().max().item()
            print(f"{name}: rel-to-max err {e:.2e}  act max {xd.abs().max().item():.1f}  V max {vm:.1f}  U max {um:.3f}", flush=True)
        fd, fw = xd.permute(0, 2, 3, 1).flatten(1), xw.permute(0, 2, 3, 1).flatten(1)
        w6 = torch.from_numpy(wq["fc6.w"]).flatten(1); w7 = torch.from_numpy(wq["fc7.w"])
        o = []
        for f in (fd, fw):
            a = h(F.relu(f @ w6.T + torch.from_numpy(wq["fc6.b"]) * s))
            o.append(F.relu(a @ w7.T + torch.from_numpy(wq["fc7.b"]) * s))
        cos = F.cosine_similarity(o[0].double(), o[1].double(), dim=1)
        print("descriptor cosine direct vs winograd:", cos.numpy())


main()
