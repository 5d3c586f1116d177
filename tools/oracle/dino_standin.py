"""nn.Module with the surface the reference touches on a DINOv2 hub model (container-only tooling).

The reference obtains its backbone from ``torch.hub.load('facebookresearch/dinov2', name)``
(P/src/model.py:342-343; third-party, needs network).  ``gen_golden.py`` monkey-patches
``torch.hub.load`` to return this stand-in so that the reference's own ``Patchioner`` glue runs
end-to-end here.  Surface used by the reference: ``.blocks[-1].attn.qkv`` (an ``nn.Linear`` that
receives the forward hook, model.py:589-590), ``.norm``, ``.patch_size``, ``.eval()``,
``forward(imgs, is_training=True) -> dict``.
"""
import math

import torch
import torch.nn as nn
import torch.nn.functional as F


class _Attn(nn.Module):
    def __init__(self, D, heads):
        super().__init__()
        self.num_heads = heads
        self.qkv = nn.Linear(D, 3 * D)
        self.proj = nn.Linear(D, D)

    def forward(self, x):
        B, T, D = x.shape
        h = self.num_heads
        q, k, v = self.qkv(x).reshape(B, T, 3, h, D // h).permute(2, 0, 3, 1, 4)
        a = ((q * (D // h) ** -0.5) @ k.transpose(-2, -1)).softmax(-1)
        return self.proj((a @ v).transpose(1, 2).reshape(B, T, D))


class _LS(nn.Module):
    def __init__(self, D):
        super().__init__()
        self.gamma = nn.Parameter(torch.ones(D))

    def forward(self, x):
        return x * self.gamma


class _Mlp(nn.Module):
    def __init__(self, D):
        super().__init__()
        self.fc1 = nn.Linear(D, 4 * D)
        self.fc2 = nn.Linear(4 * D, D)

    def forward(self, x):
        return self.fc2(F.gelu(self.fc1(x)))


class _Block(nn.Module):
    def __init__(self, D, heads):
        super().__init__()
        self.norm1 = nn.LayerNorm(D, eps=1e-6)
        self.attn = _Attn(D, heads)
        self.ls1 = _LS(D)
        self.norm2 = nn.LayerNorm(D, eps=1e-6)
        self.mlp = _Mlp(D)
        self.ls2 = _LS(D)

    def forward(self, x):
        x = x + self.ls1(self.attn(self.norm1(x)))
        return x + self.ls2(self.mlp(self.norm2(x)))


class _PatchEmbed(nn.Module):
    def __init__(self, D, p):
        super().__init__()
        self.proj = nn.Conv2d(3, D, p, p)


class DinoStandIn(nn.Module):
    def __init__(self, D, depth, heads, registers=4, grid=37, patch_size=14):
        super().__init__()
        self.patch_size = patch_size
        self.cls_token = nn.Parameter(torch.zeros(1, 1, D))
        self.pos_embed = nn.Parameter(torch.zeros(1, 1 + grid * grid, D))
        self.register_tokens = nn.Parameter(torch.zeros(1, registers, D)) if registers else None
        self.mask_token = nn.Parameter(torch.zeros(1, D))
        self.patch_embed = _PatchEmbed(D, patch_size)
        self.blocks = nn.ModuleList([_Block(D, heads) for _ in range(depth)])
        self.norm = nn.LayerNorm(D, eps=1e-6)
        self.R = registers

    def _pos(self, n_h, n_w):
        pe = self.pos_embed
        N = pe.shape[1] - 1
        M = int(math.sqrt(N))
        if n_h * n_w == N and n_h == n_w:
            return pe
        D = pe.shape[-1]
        pp = F.interpolate(pe[:, 1:].reshape(1, M, M, D).permute(0, 3, 1, 2), size=(n_h, n_w),
                           mode="bicubic", antialias=True)
        return torch.cat([pe[:, :1], pp.permute(0, 2, 3, 1).reshape(1, -1, D)], dim=1)

    def forward(self, imgs, is_training=False):
        B = imgs.shape[0]
        x = self.patch_embed.proj(imgs)
        n_h, n_w = x.shape[-2:]
        x = x.flatten(2).transpose(1, 2)
        x = torch.cat([self.cls_token.expand(B, -1, -1), x], 1) + self._pos(n_h, n_w)
        if self.R:
            x = torch.cat([x[:, :1], self.register_tokens.expand(B, -1, -1), x[:, 1:]], 1)
        for b in self.blocks:
            x = b(x)
        xn = self.norm(x)
        out = {"x_norm_clstoken": xn[:, 0], "x_norm_regtokens": xn[:, 1:self.R + 1],
               "x_norm_patchtokens": xn[:, self.R + 1:], "x_prenorm": x, "masks": None}
        return out if is_training else out["x_norm_clstoken"]
