"""CPU ORACLE for the frozen CLIP text tower.

*** TEST INFRASTRUCTURE ONLY. ***  Only `tests/`, `__graft_entry__.smoke()` and
`bench.py`'s cpu_baseline leg may import this file; the product never does.

Restates, in plain PyTorch CPU tensor algebra, what the reference
`CLIP_TEncoder.forward` executes (/root/reference/src/models/encoders.py:30-37):
token + positional embedding, `Transformer` of `ResidualAttentionBlock`s with the
causal additive mask (/root/reference/src/models/base/clip_base.py:186-220,
222-254, 390-396), `ln_final`, then the EOT-token row times `text_projection`.
`nn.MultiheadAttention` / `nn.LayerNorm` / `nn.Embedding` live in PyTorch
(third-party, torch==2.4.0, reference pyproject.toml:26) and are restated from
their documented math.

Pinning: golden vectors produced by running the reference's own
`clip_base.CLIP` text modules in the build container
(tools/make_golden.py -> tests/golden/text_*.npz), checked in
tests/test_oracle_golden.py.

Weights: mapping `P[name] -> Tensor` with the reference state_dict names of
`CLIP_TEncoder` (token_embedding.weight, positional_embedding,
transformer.resblocks.N.{attn.in_proj_weight,...}, ln_final.*, text_projection).
"""
from __future__ import annotations

import math
from typing import Mapping, Tuple

import torch
from torch import Tensor


def _layer_norm(x: Tensor, w: Tensor, b: Tensor, eps: float = 1e-5) -> Tensor:
    mu = x.mean(-1, keepdim=True)
    var = ((x - mu) ** 2).mean(-1, keepdim=True)
    return (x - mu) / torch.sqrt(var + eps) * w + b


def _quick_gelu(x: Tensor) -> Tensor:
    """clip_base.py:166-168."""
    return x * torch.sigmoid(1.702 * x)


def residual_attention_block(P: Mapping[str, Tensor], pre: str, x: Tensor, H: int) -> Tensor:
    """clip_base.py:211-220 on batch-first x [B,L,W]."""
    B, L, W = x.shape
    dh = W // H
    h = _layer_norm(x, P[pre + 'ln_1.weight'], P[pre + 'ln_1.bias'])
    qkv = h @ P[pre + 'attn.in_proj_weight'].t() + P[pre + 'attn.in_proj_bias']
    q, k, v = qkv.split(W, dim=-1)
    q = (q * (1.0 / math.sqrt(dh))).reshape(B, L, H, dh).transpose(1, 2)
    k = k.reshape(B, L, H, dh).transpose(1, 2)
    v = v.reshape(B, L, H, dh).transpose(1, 2)
    s = q @ k.transpose(-1, -2)
    mask = torch.full((L, L), float('-inf'), dtype=x.dtype).triu(1)          # clip_base.py:390-396
    a = torch.softmax(s + mask, dim=-1)
    o = (a @ v).transpose(1, 2).reshape(B, L, W)
    x = x + (o @ P[pre + 'attn.out_proj.weight'].t() + P[pre + 'attn.out_proj.bias'])
    h = _layer_norm(x, P[pre + 'ln_2.weight'], P[pre + 'ln_2.bias'])
    f = _quick_gelu(h @ P[pre + 'mlp.c_fc.weight'].t() + P[pre + 'mlp.c_fc.bias'])
    return x + (f @ P[pre + 'mlp.c_proj.weight'].t() + P[pre + 'mlp.c_proj.bias'])


def clip_text_forward(P: Mapping[str, Tensor], text: Tensor, heads: int) -> Tuple[Tensor, Tensor]:
    """encoders.py:30-37: returns (quest [B,E], words [B,L,W])."""
    text = text.long()
    B, L = text.shape
    x = P['token_embedding.weight'][text] + P['positional_embedding'][:L]
    n_layers = 1 + max(int(k.split('.')[2]) for k in P if k.startswith('transformer.resblocks.'))
    for i in range(n_layers):
        x = residual_attention_block(P, f'transformer.resblocks.{i}.', x, heads)
    x = _layer_norm(x, P['ln_final.weight'], P['ln_final.bias'])
    eot = text.argmax(dim=-1)
    return x[torch.arange(B), eot] @ P['text_projection'], x
