"""MI355X-native fusion blocks behind the reference's module surface.

Same class names, constructor signatures, parameter names/shapes and call
conventions as the reference's `src/models/modules.py`
(`Projection` 19-43, `AVQCrossAttn` 159-265, `QstGrounding` 268-347,
`TempMoE` 350-612, `PatchSelecter` 615-716), so reference checkpoints load with
`load_state_dict` and `QA_TIGER` drops into `src/train.py` / `src/test.py`.

The torch.nn members (`nn.Linear`, `nn.LayerNorm`, `nn.MultiheadAttention`) are
kept ONLY as parameter containers (identical names and default initialisers);
their `forward` is never called.  Every forward/backward FLOP goes through the
hand-written HIP kernels of libqatiger_hip.so via `qa-tiger_amd/ops.py`.  The
execution plan is not the reference's:
  * both directions of AVQCrossAttn run as ONE stacked batch [audio; video]
    (shared weights, reference modules.py:257-259), the words K/V projection is
    computed once;
  * q/k/v are consumed in place from the packed in-projection output (no
    head-split permutes), batch-first throughout (the reference permutes to
    (L,N,E) at every call);
  * residual + dropout + LayerNorm are one kernel;
  * attentions with one or two queries (TempMoE.qst_attn, PatchSelecter.crs_attn) never project
    K and V: W_k is folded into the query, W_v is applied after the attention-weighted sum of the raw
    rows (csrc/xattn.hip);
  * TempMoE runs all experts' first layers as one GEMM, reduces over time with
    the quirk-exact coefficient table, and applies the second layers after the
    reduction (linear), never materialising the (T,B,K,D) stack
    (reference modules.py:590-599).
Not implemented (never used by `QA_TIGER`): `AVCrossAttn`, `TSPM_topKSelection`.
"""
from __future__ import annotations

import importlib
from typing import List, Optional, Union

import torch
import torch.nn as nn
from torch import Tensor

_qt = importlib.import_module('qa-tiger_amd')      # repo root must be on sys.path
ops = _qt.ops
AttnSpec = ops.AttnSpec


def _kaiming_(m: nn.Module):
    # reference modules.py:189-192, 296-302, 413-418, 649-654
    if isinstance(m, nn.Linear):
        nn.init.kaiming_normal_(m.weight)
        if m.bias is not None:
            nn.init.constant_(m.bias, 0)


def _mha_proj(mha: nn.MultiheadAttention, which: str):
    """Row blocks of the packed in-projection (q | k | v), reference Appendix C of SURVEY.md."""
    E = mha.embed_dim
    W, b = mha.in_proj_weight, mha.in_proj_bias
    if which == 'q':
        return W[:E], b[:E]
    if which == 'kv':
        return W[E:], b[E:]
    return W, b


class Projection(nn.Module):
    """reference modules.py:19-43"""

    def __init__(self, inp_dim: int = 512, d_model: int = 512):
        super().__init__()
        self.proj = nn.Linear(inp_dim, d_model)

    def forward(self, inp: Tensor) -> Tensor:
        # `emit_image` (set by QA_TIGER on patch_proj): the result's first consumer is a large Linear layer (PatchSelecter's packed
        # in-projection), so the GEMM epilogue writes that layer's operand image along with the fp32 rows
        return ops.linear(inp, self.proj.weight, self.proj.bias, emit_image=getattr(self, 'emit_image', False))


class AVQCrossAttn(nn.Module):
    """reference modules.py:159-265"""

    def __init__(self, d_model: int = 512, nhead: int = 8, dropout: float = 0.1):
        super().__init__()
        self.qst_attn = nn.MultiheadAttention(d_model, nhead, dropout=dropout)
        self.crs_attn = nn.MultiheadAttention(d_model, nhead, dropout=dropout)
        self.slf_attn = nn.MultiheadAttention(d_model, nhead, dropout=dropout)
        self.linear1 = nn.Linear(d_model, d_model)
        self.linear2 = nn.Linear(d_model, d_model)
        self.dropout = nn.Dropout(dropout)
        self.norm1 = nn.LayerNorm(d_model)
        self.norm2 = nn.LayerNorm(d_model)
        self.nhead = nhead
        _kaiming_(self.linear1)
        _kaiming_(self.linear2)

    def forward(self, src_q: Tensor, src_v: Tensor, query: Tensor, visualize: bool = False) -> List[Tensor]:
        """`visualize=True` also returns `[a_weight, v_weight]`, the head-averaged question-attention maps [B,T,L] of the
        two directions (what `nn.MultiheadAttention` returns as its second value, post-dropout in train mode; reference
        modules.py:217, 261-263), computed by `qt_attn_avg_weights` with the forward's own dropout seed."""
        B, T, D = src_q.shape
        L = query.shape[1]
        H = self.nhead
        p = self.dropout.p if self.training else 0.0
        nxt = ops.seeds.next
        # [2B,T,D]: items 0..B-1 audio, B..2B-1 video; its gradient (three in-projection dX GEMMs + the residual of norm1) meets in one buffer
        x = ops.share_grad(torch.cat([src_q, src_v], dim=0))
        # self-attention (packed QKV, each item attends to itself)
        qkv = ops.linear(x, *_mha_proj(self.slf_attn, 'qkv'))
        fl = (D, D)                                                            # each attention output feeds only its out-projection
        o_s = ops.attention(qkv, None, AttnSpec(2 * B, 2 * B, 0, H, T, T, D, 0, D, 2 * D), p, nxt(), for_linear=fl)
        # cross-modal attention: audio item b reads video item b and vice versa (shift B)
        qkv = ops.linear(x, *_mha_proj(self.crs_attn, 'qkv'))
        o_c = ops.attention(qkv, None, AttnSpec(2 * B, 2 * B, B, H, T, T, D, 0, D, 2 * D), p, nxt(), for_linear=fl)
        # question attention: both directions share one words K/V projection
        q = ops.linear(x, *_mha_proj(self.qst_attn, 'q'))
        kv = ops.linear(query, *_mha_proj(self.qst_attn, 'kv'))
        qst_spec, qst_seed = AttnSpec(2 * B, B, 0, H, T, L, D, 0, 0, D), nxt()
        o_q = ops.attention(q, kv, qst_spec, p, qst_seed, for_linear=fl)
        # (kept as [2B*T, D]: add_layer_norm takes any shapes with the same rows; a reshape node between the two would strip the
        # max|dY| tag the LayerNorm backward puts on these layers' gradients)
        a_s = ops.linear(o_s, self.slf_attn.out_proj.weight, self.slf_attn.out_proj.bias)
        a_c = ops.linear(o_c, self.crs_attn.out_proj.weight, self.crs_attn.out_proj.bias)
        a_q = ops.linear(o_q, self.qst_attn.out_proj.weight, self.qst_attn.out_proj.bias)
        x = ops.share_grad(ops.add_layer_norm([x, a_s, a_c, a_q], self.norm1.weight, self.norm1.bias,
                                              drops=[False, True, True, True], drop_p=p))      # feeds linear1 and the residual of norm2
        h = ops.linear(x, self.linear1.weight, self.linear1.bias, relu=True, drop_p=p, seed=nxt(), emit_image=True)    # h feeds linear2
        f = ops.linear(h, self.linear2.weight, self.linear2.bias)
        x = ops.add_layer_norm([x, f], self.norm2.weight, self.norm2.bias, drops=[False, True], drop_p=p)
        if visualize:
            w = ops.attention_avg_weights(q, kv, qst_spec, p, qst_seed)       # [2B,T,L]: audio direction, then video
            xa, xv = ops.split_rows(x, B)
            return xa, xv, [w[:B], w[B:]]
        return list(ops.split_rows(x, B))          # (the two halves' gradients meet in one buffer: ops._SplitRowsFn)


class QstGrounding(nn.Module):
    """reference modules.py:268-347"""

    def __init__(self, d_model: int = 512, nhead: int = 8, dropout: float = 0.1):
        super().__init__()
        self.act = nn.ReLU()
        self.norm = nn.LayerNorm(d_model)
        self.attn = nn.MultiheadAttention(d_model, nhead, dropout=0.1)
        self.mlp = nn.Sequential(nn.Linear(d_model, d_model // 2), nn.ReLU(), nn.Linear(d_model // 2, d_model))
        self.dropout = nn.Dropout(dropout)
        self.nhead = nhead
        self.mlp.apply(_kaiming_)

    def forward(self, qst: Tensor, data: Union[Tensor, List[Tensor]], relu_out: bool = False) -> Tensor:
        """`relu_out` (not in the reference signature) fuses the head's ReLU
        (reference net.py:141) into the LayerNorm kernel."""
        d = torch.cat(list(data), dim=1) if isinstance(data, (list, tuple)) else data   # [B,S,D]
        B, S, D = d.shape
        p = self.dropout.p if self.training else 0.0
        q = ops.linear(qst, *_mha_proj(self.attn, 'q'))
        kv = ops.linear(d, *_mha_proj(self.attn, 'kv'))
        o = ops.attention(q, kv, AttnSpec(B, B, 0, self.nhead, 1, S, D, 0, 0, D), self.attn.dropout if self.training else 0.0,
                          ops.seeds.next())
        a = ops.linear(o, self.attn.out_proj.weight, self.attn.out_proj.bias)
        m = ops.linear(ops.linear(a, self.mlp[0].weight, self.mlp[0].bias, relu=True), self.mlp[2].weight, self.mlp[2].bias)
        # LayerNorm(mean_S(data) + dropout(mlp(attn)))  — the mean is S addends scaled 1/S (the path uses S = 2, net.py:136-138);
        # the fused kernel takes 4 addends, so a longer sequence (the reference is generic, modules.py:304-347) is summed first
        if S <= 3:
            xs = [d[:, s, :] for s in range(S)] + [m]
            return ops.add_layer_norm(xs, self.norm.weight, self.norm.bias, scales=[1.0 / S] * S + [1.0],
                                      drops=[False] * S + [True], drop_p=p, relu_out=relu_out)
        return ops.add_layer_norm([_SumSeqFn.apply(d), m], self.norm.weight, self.norm.bias, scales=[1.0 / S, 1.0],
                                  drops=[False, True], drop_p=p, relu_out=relu_out)


class _SumSeqFn(torch.autograd.Function):
    """[B,S,D] -> [B,D]: sum over the sequence axis with the row-add kernel (QstGrounding over more than 3 tokens)."""

    @staticmethod
    def forward(ctx, d: Tensor):
        ctx.S = d.shape[1]
        acc = ops.add_rows(d[:, 0, :], d[:, 1, :])
        for s in range(2, d.shape[1]):
            acc = ops.add_rows(acc, d[:, s, :])
        return acc

    @staticmethod
    def backward(ctx, g: Tensor):
        return g.unsqueeze(1).expand(g.shape[0], ctx.S, g.shape[1])


class _AddRowsFn(torch.autograd.Function):
    """out[s*R + r] = a[r] + b[s*R + r]   (a broadcast over the S stacked streams)."""

    @staticmethod
    def forward(ctx, a: Tensor, b: Tensor):
        ctx.ra = a.shape[0]
        return ops.add_rows(a, b)

    @staticmethod
    def backward(ctx, g: Tensor):
        g = ops.as2d(g)
        ra = ctx.ra
        S = g.shape[0] // ra
        da = g
        if S == 2:
            da = ops.add_rows(g[:ra], g[ra:])
        elif S > 2:
            da = g[:ra]
            for s in range(1, S):
                da = ops.add_rows(da, g[s * ra:(s + 1) * ra])
        return da, g


class TempMoE(nn.Module):
    """reference modules.py:350-612"""

    def __init__(self, d_model: int = 512, nhead: int = 8, topK: int = 5, n_experts: int = 10, sigma: int = 9,
                 dropout: float = 0.1, vis_branch: bool = False):
        super().__init__()
        if not sigma > 0:
            raise ValueError(f'TempMoE: sigma must be positive, got {sigma}')
        self.sigma, self.topK, self.n_experts = sigma, topK, n_experts
        if vis_branch:
            self.anorm = nn.LayerNorm(d_model)
            self.vnorm = nn.LayerNorm(d_model)
        else:
            self.norm = nn.LayerNorm(d_model)
        self.dropout = nn.Dropout(dropout)          # defined but never applied (reference modules.py:380)
        self.qst_attn = nn.MultiheadAttention(d_model, nhead, dropout=0.1)
        self.gauss_pred = nn.Sequential(nn.Linear(d_model, 2 * n_experts))
        self.router = nn.Sequential(nn.Linear(d_model, n_experts))
        self.experts = nn.ModuleList([
            nn.Sequential(nn.Linear(d_model, d_model // 2), nn.ReLU(), nn.Linear(d_model // 2, d_model))
            for _ in range(n_experts)])
        self.experts.apply(_kaiming_)
        self.margin = 1 / (n_experts * 2)
        # plain attribute, not a buffer, exactly like the reference (modules.py:405-408, quirk Q3);
        # the kernels recompute it from K
        self.center = torch.linspace(self.margin, 1 - self.margin, n_experts)
        self.router.apply(_kaiming_)
        self.gauss_pred.apply(_kaiming_)
        self.nhead = nhead
        self.last = {}                                 # diagnostics of the last call (detached)

    def forward(self, qst: Tensor, data: Tensor, sub_data: Optional[List[Tensor]] = None):
        B, T, D = data.shape
        K, k, H = self.n_experts, self.topK, self.nhead
        pa = self.qst_attn.dropout if self.training else 0.0
        # question -> time attention (Lq = 1)
        q = ops.linear(qst, *_mha_proj(self.qst_attn, 'q'))
        if ops.fold_kv:     # one query per sample: fold W_k into it, apply W_v after the weighted sum (no K/V GEMM over B*T rows)
            o = ops.folded_attention(q, data.reshape(B * T, D), self.qst_attn, B, 1, T, 1, 0, pa, ops.seeds.next())
        else:
            kv = ops.linear(data, *_mha_proj(self.qst_attn, 'kv'))
            o = ops.attention(q, kv, AttnSpec(B, B, 0, H, 1, T, D, 0, 0, D), pa, ops.seeds.next())
        temp_w = ops.linear(o, self.qst_attn.out_proj.weight, self.qst_attn.out_proj.bias)          # [B,D]
        logits = ops.linear(temp_w, self.router[0].weight, self.router[0].bias)                      # [B,K]
        raw = ops.linear(temp_w, self.gauss_pred[0].weight, self.gauss_pred[0].bias)                 # [B,2K]
        coef, csum, ti, tp, gauss = ops.MoECoefFn.apply(logits, raw, T, K, k, float(self.sigma))
        self.last = dict(temp_w=temp_w.detach(), router_logits=logits.detach(), topk_inds=ti, topk_probs=tp, gauss=gauss)
        # all experts, first layer, as one GEMM over the stacked streams
        w1 = ops.stack_params('rows', [e[0].weight for e in self.experts])        # [K*Dh, D]
        b1 = ops.stack_params('rows', [e[0].bias for e in self.experts])
        w2 = ops.stack_params('cols', [e[2].weight for e in self.experts])        # [D, K*Dh]
        b2t = ops.stack_params('stack1', [e[2].bias for e in self.experts])       # [D, K]
        d2 = data.reshape(B * T, D)
        if sub_data is not None:
            S = len(sub_data)
            sub2 = [s.reshape(B * T, D) for s in sub_data]
            stacked = ops.join_rows(sub2[0], sub2[1]) if S == 2 else torch.cat(sub2, dim=0)
            xe = _AddRowsFn.apply(d2, stacked)                                                        # [S*BT, D]
        else:
            S, xe = 1, d2
        hh = ops.linear(xe, w1, b1, relu=True, grad_premasked=True)       # [S*BT, K*Dh]
        r = ops.MoEReduceFn.apply(hh, coef, S, K)                          # [S*B, K*Dh]
        bias_term = ops.linear(csum.repeat(S, 1) if S > 1 else csum, b2t)  # Σ_e csum[b,e]·b2_e
        out = ops.linear(r, w2, None, residual=bias_term)                  # [S*B, D]
        if sub_data is not None:
            outs = [ops.add_layer_norm([out[s * B:(s + 1) * B]], n.weight, n.bias).reshape(B, 1, D)
                    for s, n in zip(range(S), (self.anorm, self.vnorm))]
            return outs[0], outs[1]
        return ops.add_layer_norm([out], self.norm.weight, self.norm.bias).reshape(B, 1, D)


class PatchSelecter(nn.Module):
    """reference modules.py:615-716"""

    def __init__(self, d_model: int = 512, nhead: int = 8, dropout: float = 0.1):
        super().__init__()
        self.d_model, self.nhead = d_model, nhead
        self.vnorm = nn.LayerNorm(d_model)
        self.anorm = nn.LayerNorm(d_model)
        self.dropout = nn.Dropout(dropout)
        self.slf_attn = nn.MultiheadAttention(d_model, nhead, dropout=dropout)
        self.crs_attn = nn.MultiheadAttention(d_model, nhead, dropout=dropout)
        self.mlp = nn.Sequential(nn.Linear(d_model, d_model // 2), nn.ReLU(), nn.Linear(d_model // 2, d_model))
        self.mlp.apply(_kaiming_)

    def forward(self, patch: Tensor, audio: Tensor, video: Tensor) -> List[Tensor]:
        B, T, P, D = patch.shape
        BT, H = B * T, self.nhead
        p = self.dropout.p if self.training else 0.0
        nxt = ops.seeds.next
        pp = ops.reshape(patch, (BT * P, D))                               # (keeps the max|d_pp| tag of the linked dX GEMM for patch_proj's backward)
        # per-frame self-attention over the P patch tokens, residual, no norm (modules.py:688)
        link = ops.GradLink()                                              # d(pp) = residual gradient + in-projection dX, summed inside the dX GEMM
        qkv = ops.linear(pp, *_mha_proj(self.slf_attn, 'qkv'), link=link)
        o = ops.attention(qkv, None, AttnSpec(BT, BT, 0, H, P, P, D, 0, D, 2 * D), p, nxt(), for_linear=(D, D))
        psa = ops.linear(o, self.slf_attn.out_proj.weight, self.slf_attn.out_proj.bias, residual=pp, link=link)
        # two queries per frame — row 0 video, row 1 audio (modules.py:692) — stacked as [video; audio]
        y2 = torch.cat([video.reshape(BT, D), audio.reshape(BT, D)], dim=0)
        qc = ops.linear(y2, *_mha_proj(self.crs_attn, 'q'))
        if ops.fold_kv:     # 2 queries against P patch rows: the K/V projection of all B*T*P rows (17 % of the path's GEMM FLOPs) is folded away
            o_c = ops.folded_attention(qc, psa, self.crs_attn, BT, 2, P, 1, BT, p, nxt())
        else:
            kv = ops.linear(psa, *_mha_proj(self.crs_attn, 'kv'))
            o_c = ops.attention(qc, kv, AttnSpec(BT, BT, 0, H, 2, P, D, 0, 0, D, q_is=1, q_rs=BT, o_rows=2 * BT), p, nxt())
        a_c = ops.linear(o_c, self.crs_attn.out_proj.weight, self.crs_attn.out_proj.bias, drop_p=p, seed=nxt(), emit_image=True)
        m = ops.linear(ops.linear(a_c, self.mlp[0].weight, self.mlp[0].bias, relu=True, emit_image=True), self.mlp[2].weight, self.mlp[2].bias)
        v_rel, a_rel = ops.split_rows(m, BT)
        # return order is [audio, video] (modules.py:713-716, quirk Q5)
        # (the two results go into the halves of one buffer: vt_aggregator reads them stacked, ops.join_rows then costs nothing)
        both = torch.empty((2, BT, D), device=m.device, dtype=torch.float32)
        return [ops.add_layer_norm([a_rel], self.anorm.weight, self.anorm.bias, out=both[0]).reshape(B, T, D),
                ops.add_layer_norm([v_rel], self.vnorm.weight, self.vnorm.bias, out=both[1]).reshape(B, T, D)]
