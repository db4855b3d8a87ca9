"""CPU ORACLE for the TSPM baseline model.

*** TEST INFRASTRUCTURE ONLY. ***  Only `tests/`, `__graft_entry__.smoke()` and
`bench.py`'s cpu_baseline leg may import this file; the product never does.

Restates, in plain PyTorch CPU tensor algebra, what the reference's second model
`TSPM.forward` executes (/root/reference/src/models/tspm.py:347-400) with float
question / prompt features (the branch at :373-376): five input Linears, the
shared-weight audio-visual `AVHanLayer` applied to the ORIGINAL inputs in both
directions (:61-69, :35-47), question-prompt attention over the video clips whose
head-averaged weights pick the top-K time steps in temporal order (:96-143),
patch-token self-attention + audio-guided patch attention on those steps
(:244-306, :209-222), question grounding on the selected audio / spatial features
(:160-187), and the tanh fusion head (:390-400).  `nn.MultiheadAttention`
(third-party torch) is restated from its math path, including the averaged
attention weights it returns by default.  Eval mode (no dropout).

Pinning: `tests/golden/tspm_*.npz`, produced by `tools/make_golden.py --tspm`
from the reference's own `TSPM` class; checked in tests/test_oracle_golden.py.
"""
from __future__ import annotations

import math
from typing import Dict, Mapping, Optional, Tuple

import torch
from torch import Tensor


def _ln(P, name: str, x: Tensor, eps: float = 1e-5) -> Tensor:
    mu = x.mean(-1, keepdim=True)
    var = ((x - mu) ** 2).mean(-1, keepdim=True)
    return (x - mu) / torch.sqrt(var + eps) * P[name + '.weight'] + P[name + '.bias']


def _lin(P, name: str, x: Tensor) -> Tensor:
    return x @ P[name + '.weight'].t() + P[name + '.bias']


def mha(P, name: str, q: Tensor, kv: Tensor, H: int) -> Tuple[Tensor, Tensor]:
    """Batch-first nn.MultiheadAttention(q, kv, kv): returns (output [N,Lq,E], head-averaged weights [N,Lq,Lk])."""
    E = q.shape[-1]
    dh = E // H
    W, b = P[name + '.in_proj_weight'], P[name + '.in_proj_bias']
    N, Lq, Lk = q.shape[0], q.shape[1], kv.shape[1]
    qq = ((q @ W[:E].t() + b[:E]) * (1.0 / math.sqrt(dh))).reshape(N, Lq, H, dh).transpose(1, 2)
    kk = (kv @ W[E:2 * E].t() + b[E:2 * E]).reshape(N, Lk, H, dh).transpose(1, 2)
    vv = (kv @ W[2 * E:].t() + b[2 * E:]).reshape(N, Lk, H, dh).transpose(1, 2)
    a = torch.softmax(qq @ kk.transpose(-1, -2), dim=-1)
    o = (a @ vv).transpose(1, 2).reshape(N, Lq, E)
    return o @ P[name + '.out_proj.weight'].t() + P[name + '.out_proj.bias'], a.mean(dim=1)


def av_han_layer(P, pre: str, src_q: Tensor, src_v: Tensor) -> Tensor:
    """tspm.py:35-47 (nhead = 1)."""
    s1, _ = mha(P, pre + '.cm_attn', src_q, src_v, 1)
    s2, _ = mha(P, pre + '.self_attn', src_q, src_q, 1)
    x = _ln(P, pre + '.norm1', src_q + s1 + s2)
    f = _lin(P, pre + '.linear2', torch.relu(_lin(P, pre + '.linear1', x)))
    return _ln(P, pre + '.norm2', x + f)


def qst_query_block(P, pre: str, q: Tensor, kv: Tensor) -> Tuple[Tensor, Tensor]:
    """The `attn -> linear1 -> relu -> linear2 -> + -> LayerNorm` block shared by TemporalPerception (:96-116),
    SpatioPerception (:277-289) and QstTemporalGrounding (:160-179); 4 heads."""
    a, w = mha(P, pre + '.attn_qst_query', q, kv, 4)
    s = _lin(P, pre + '.qst_query_linear2', torch.relu(_lin(P, pre + '.qst_query_linear1', a)))
    return _ln(P, pre + '.qst_query_visual_norm', a + s), w


def tokens_self_attn(P, pre: str, x: Tensor) -> Tensor:
    """tspm.py:209-222 (nhead = 1)."""
    s, _ = mha(P, pre + '.self_attn', x, x, 1)
    x = _ln(P, pre + '.norm1', x + s)
    f = _lin(P, pre + '.linear2', torch.relu(_lin(P, pre + '.linear1', x)))
    return _ln(P, pre + '.norm2', x + f)


def tspm_forward(P: Mapping[str, Tensor], inp: Mapping[str, Tensor], topK: int,
                 taps: Optional[Dict[str, Tensor]] = None) -> Tensor:
    audio, video, patch = inp['audio'], inp['video'], inp['patch']
    quest, prompt = inp['quest'].squeeze(1), inp['prompt'].squeeze(1)
    B, T = audio.shape[:2]
    a = _lin(P, 'input_a', audio)
    v = _lin(P, 'input_v', video)
    pt = _lin(P, 'input_v_patch', patch)                             # [B,T,P,C]
    qf = _lin(P, 'input_qst', quest)
    qp = _lin(P, 'input_qst_prompt', prompt)
    a_av = av_han_layer(P, 'AV_Attn.layers.0', a, v)                 # both directions read the ORIGINAL a, v (:66-67)
    v_av = av_han_layer(P, 'AV_Attn.layers.0', v, a)
    # TemporalPerception: the prompt attends to the video clips; its averaged weights pick the top-K steps (:118-143)
    _, w = qst_query_block(P, 'TemporalPerception', qp[:, None], v)  # w [B,1,T]
    idx = torch.argsort(w, dim=-1)[:, :, -topK:]
    idx, _ = torch.sort(idx)                                         # temporal order
    sel = idx[:, 0]                                                  # [B,topK]
    ar = torch.arange(B)[:, None]
    a_ts, v_ts = a[ar, sel], v[ar, sel]                              # [B,topK,C]
    p_sel = pt[ar, sel]                                              # [B,topK,P,C]  (:257-272)
    C = a.shape[-1]
    tok = tokens_self_attn(P, 'SpatioPerception.TokensAttn', p_sel.reshape(B * topK, -1, C))
    v_sp, _ = qst_query_block(P, 'SpatioPerception', a_ts.reshape(B * topK, 1, C), tok)
    v_sp = v_sp.reshape(B, topK, C)
    a_q, _ = qst_query_block(P, 'QstTempGrd_Module', qf[:, None], a_ts)
    v_q, _ = qst_query_block(P, 'QstTempGrd_Module', qf[:, None], v_sp)
    feat = torch.cat([a_q[:, 0], a_av.mean(1), a_ts.mean(1), v_q[:, 0], v_av.mean(1), v_sp.mean(1)], dim=-1)
    av = _lin(P, 'av_fusion_fc', torch.tanh(feat))
    out = _lin(P, 'answer_pred_fc', torch.tanh(av * qf))
    if taps is not None:
        taps.update(temp_weights=w, top_k_index_sort=idx, audio_AVattn=a_av, visual_AVattn=v_av, audio_tssm=a_ts,
                    visual_sp=v_sp, audio_qtgm=a_q[:, 0], visual_qtgm=v_q[:, 0], av_feat=av)
    return out


def forward_backward(P: Mapping[str, Tensor], inp: Mapping[str, Tensor], topK: int, taps=None):
    Pg = {k: v.clone().requires_grad_(True) for k, v in P.items()}
    logits = tspm_forward(Pg, inp, topK, taps)
    loss = torch.nn.functional.cross_entropy(logits, inp['label'])
    loss.backward()
    grads = {k: (v.grad if v.grad is not None else torch.zeros_like(v)) for k, v in Pg.items()}
    return loss.detach(), logits.detach(), grads
