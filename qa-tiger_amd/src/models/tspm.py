"""TSPM — the reference's second model (reference src/models/tspm.py:309-400), on the HIP kernels.

Same class / attribute / parameter names as the reference, so `TSPM(**cfg.hyper_params.model)`,
`load_state_dict` of a reference checkpoint and `model(reshaped_data) -> {'out': [B,42]}` work unchanged
(`get_model`, src/trainutils.py:88-89).  Question and prompt arrive as float features `[B,1,qst_dim]`
(the branch at tspm.py:373-376); the token branch needs the text tower (`encoders.CLIP_TEncoder`).

What differs from the reference's execution plan (same math):
* audio and video go through the shared-weight `AVHanLayer` as one stacked batch; the cross-modal attention reads the
  other half through the kernel's kv-item shift (as in AVQCrossAttn);
* `nn.MultiheadAttention(512, nhead=1)` has a 512-wide head: it runs as batched GEMMs + a row-softmax kernel
  (`ops.bmm_attention`); the 4-head question attentions run on the fused attention kernels;
* the top-K time steps are gathered BEFORE the patch projection (`input_v_patch` is row-wise, so
  gather(linear(x)) = linear(gather(x))): T/topK = 6x fewer rows through the largest GEMM;
* the Python double loops of `SelectTopK` / `TopKSegs` (tspm.py:127-132, 248-257) are one device gather;
* `TemporalPerception`'s feed-forward block output is discarded by the reference (tspm.py:140-143), so it is not
  computed; its parameters receive zero gradients in both.
"""
from __future__ import annotations

import importlib
from typing import Dict

import torch
import torch.nn as nn
from torch import Tensor

_pkg = __name__.rsplit('.', 3)[0]
ops = importlib.import_module(_pkg + '.ops')
_mods = importlib.import_module(_pkg + '.src.models.modules')
_mha_proj = _mods._mha_proj
AttnSpec = ops.AttnSpec
C_HID = 512


def _p(mod: nn.Module) -> float:
    return 0.1 if mod.training else 0.0


def _single_head_attention(qkv: Tensor, n: int, L: int, C: int, shift: int, p: float, seed: int) -> Tensor:
    """Packed-QKV self/cross attention with ONE head of width C.  Short sequences (the 14 patch tokens) fit the generic
    attention kernel's LDS tiles even at d_h = 512; longer ones (T = 60) go through batched GEMMs + row softmax."""
    if 4 * (4 * L * (C + 1) + 2 * L * (L + 1)) <= 160 * 1024:
        return ops.attention(qkv, None, AttnSpec(n, n, shift, 1, L, L, C, 0, C, 2 * C), p, seed)
    return ops.bmm_attention(qkv, None, n, L, L, C, 0, C, 2 * C, kv_shift=shift, drop_p=p, seed=seed)


class AVHanLayer(nn.Module):
    """reference tspm.py:15-47 (nhead = 1)"""

    def __init__(self, d_model, nhead, dim_feedforward=512, dropout=0.1):
        super().__init__()
        self.self_attn = nn.MultiheadAttention(d_model, nhead, dropout=dropout)
        self.cm_attn = nn.MultiheadAttention(d_model, nhead, dropout=dropout)
        self.linear1 = nn.Linear(d_model, dim_feedforward)
        self.linear2 = nn.Linear(dim_feedforward, d_model)
        self.norm1 = nn.LayerNorm(d_model)
        self.norm2 = nn.LayerNorm(d_model)
        self.d_model = d_model

    def forward_stacked(self, x: Tensor, B: int, T: int) -> Tensor:
        """x [2B,T,C] = [audio; video]; item n cross-attends to item (n+B) mod 2B."""
        C, p, nxt = self.d_model, _p(self), ops.seeds.next
        qkv = ops.linear(x, *_mha_proj(self.cm_attn, 'qkv'))
        o1 = _single_head_attention(qkv, 2 * B, T, C, B, p, nxt())
        qkv = ops.linear(x, *_mha_proj(self.self_attn, 'qkv'))
        o2 = _single_head_attention(qkv, 2 * B, T, C, 0, p, nxt())
        s1 = ops.linear(o1, self.cm_attn.out_proj.weight, self.cm_attn.out_proj.bias).reshape(2 * B, T, C)
        s2 = ops.linear(o2, self.self_attn.out_proj.weight, self.self_attn.out_proj.bias).reshape(2 * B, T, C)
        x = ops.add_layer_norm([x, s1, s2], self.norm1.weight, self.norm1.bias, drops=[False, True, True], drop_p=p)
        f = ops.linear(ops.linear(x, self.linear1.weight, self.linear1.bias, relu=True, drop_p=p, seed=nxt()),
                       self.linear2.weight, self.linear2.bias)
        return ops.add_layer_norm([x, f], self.norm2.weight, self.norm2.bias, drops=[False, True], drop_p=p)


class AV_Attn(nn.Module):
    """reference tspm.py:50-75 (num_layers = 1, norm = None: `norm1`/`norm2` exist but are never applied)"""

    def __init__(self, encoder_layer, num_layers, norm=None):
        super().__init__()
        self.layers = nn.ModuleList([encoder_layer])
        self.num_layers = num_layers
        self.norm1 = nn.LayerNorm(C_HID)
        self.norm2 = nn.LayerNorm(C_HID)
        self.norm = norm

    def forward(self, src_a: Tensor, src_v: Tensor):
        B, T, _ = src_a.shape
        x = self.layers[0].forward_stacked(torch.cat([src_a, src_v], dim=0), B, T)   # both directions read the ORIGINAL inputs (:66-67)
        a, v = x[:B], x[B:]
        if self.norm:
            a = ops.add_layer_norm([a], self.norm1.weight, self.norm1.bias)
            v = ops.add_layer_norm([v], self.norm2.weight, self.norm2.bias)
        return a, v


class _QstQuery(nn.Module):
    """attn(4 heads) -> linear1 -> ReLU -> linear2 -> + -> LayerNorm: tspm.py:96-116, 160-179, 277-289."""

    def __init__(self):
        super().__init__()
        self.attn_qst_query = nn.MultiheadAttention(C_HID, 4, dropout=0.1)
        self.qst_query_linear1 = nn.Linear(C_HID, C_HID)
        self.qst_query_linear2 = nn.Linear(C_HID, C_HID)
        self.qst_query_visual_norm = nn.LayerNorm(C_HID)

    def _attend(self, q_in: Tensor, kv_in: Tensor, n: int, Lk: int, need_weights: bool = False):
        """one query row per item: q_in [n,C], kv_in [n*Lk,C] -> attention output [n,C] (before out_proj), weights [n,1,Lk]"""
        p = _p(self)
        q = ops.linear(q_in, *_mha_proj(self.attn_qst_query, 'q'))
        kv = ops.linear(kv_in, *_mha_proj(self.attn_qst_query, 'kv'))
        spec = AttnSpec(n, n, 0, 4, 1, Lk, C_HID, 0, 0, C_HID)
        seed = ops.seeds.next()
        o = ops.attention(q, kv, spec, p, seed)
        w = ops.attention_avg_weights(q, kv, spec, p, seed) if need_weights else None
        return o, w

    def _block(self, q_in: Tensor, kv_in: Tensor, n: int, Lk: int) -> Tensor:
        p = _p(self)
        o, _ = self._attend(q_in, kv_in, n, Lk)
        a = ops.linear(o, self.attn_qst_query.out_proj.weight, self.attn_qst_query.out_proj.bias)
        s = ops.linear(ops.linear(a, self.qst_query_linear1.weight, self.qst_query_linear1.bias, relu=True, drop_p=p, seed=ops.seeds.next()),
                       self.qst_query_linear2.weight, self.qst_query_linear2.bias)
        return ops.add_layer_norm([a, s], self.qst_query_visual_norm.weight, self.qst_query_visual_norm.bias,
                                  drops=[False, True], drop_p=p)


class TemporalPerception(_QstQuery):
    """reference tspm.py:77-143"""

    def __init__(self, topK=10):
        super().__init__()
        self.topK = topK

    def forward(self, audio_input: Tensor, visual_input: Tensor, qst_input: Tensor):
        B, T, C = audio_input.shape
        _, w = self._attend(qst_input, visual_input.reshape(B * T, C), B, T, need_weights=True)   # [B,1,T]
        idx = torch.argsort(w, dim=-1)[:, :, -self.topK:]
        idx, _ = torch.sort(idx)                                     # temporal order (:122)
        sel = idx[:, 0]
        ar = torch.arange(B, device=sel.device)[:, None]
        return audio_input[ar, sel], visual_input[ar, sel], idx


class QstTemporalGrounding(_QstQuery):
    """reference tspm.py:146-187"""

    def forward(self, qst_input: Tensor, audio_input: Tensor, visual_input: Tensor):
        B, K, C = audio_input.shape
        return (self._block(qst_input, audio_input.reshape(B * K, C), B, K),
                self._block(qst_input, visual_input.reshape(B * K, C), B, K))


class TokensSelfAttn(nn.Module):
    """reference tspm.py:189-222 (nhead = 1)"""

    def __init__(self, d_model, nhead, dim_feedforward=512, dropout=0.1):
        super().__init__()
        self.self_attn = nn.MultiheadAttention(d_model, nhead, dropout=dropout)
        self.linear1 = nn.Linear(d_model, dim_feedforward)
        self.linear2 = nn.Linear(dim_feedforward, d_model)
        self.norm1 = nn.LayerNorm(d_model)
        self.norm2 = nn.LayerNorm(d_model)

    def forward(self, x: Tensor) -> Tensor:                          # [N,P,C]
        N, P, C = x.shape
        p, nxt = _p(self), ops.seeds.next
        qkv = ops.linear(x, *_mha_proj(self.self_attn, 'qkv'))
        o = _single_head_attention(qkv, N, P, C, 0, p, nxt())
        s = ops.linear(o, self.self_attn.out_proj.weight, self.self_attn.out_proj.bias).reshape(N, P, C)
        x = ops.add_layer_norm([x, s], self.norm1.weight, self.norm1.bias, drops=[False, True], drop_p=p)
        f = ops.linear(ops.linear(x, self.linear1.weight, self.linear1.bias, relu=True, drop_p=p, seed=nxt()),
                       self.linear2.weight, self.linear2.bias)
        return ops.add_layer_norm([x, f], self.norm2.weight, self.norm2.bias, drops=[False, True], drop_p=p)


class SpatioPerceptionModule(_QstQuery):
    """reference tspm.py:225-306"""

    def __init__(self, topK=10):
        super().__init__()
        self.topK = topK
        self.TokensAttn = TokensSelfAttn(d_model=C_HID, nhead=1)

    def forward(self, audios_feat: Tensor, visual_patch_top_k: Tensor) -> Tensor:
        """audios_feat [B,K,C]; visual_patch_top_k [B,K,P,C] (already gathered and projected)"""
        B, K, P, C = visual_patch_top_k.shape
        tok = self.TokensAttn(visual_patch_top_k.reshape(B * K, P, C))
        return self._block(audios_feat.reshape(B * K, C), tok.reshape(B * K * P, C), B * K, P).reshape(B, K, C)


class TSPM(nn.Module):
    def __init__(self, topK=10, avq_cross_attn: bool = False, audio_dim: int = 128, vis_dim: int = 768, patch_dim: int = 1024,
                 qst_dim: int = 768, hidden_size=512, **kwargs):
        super().__init__()
        if hidden_size != C_HID:
            raise ValueError('TSPM sub-modules are 512 wide (hard-coded in the reference, tspm.py:84-90)')
        if avq_cross_attn:
            raise NotImplementedError('avq_cross_attn=True leaves avq_feat undefined in the reference (tspm.py:395-397)')
        self.input_a = nn.Linear(audio_dim, hidden_size)
        self.input_v = nn.Linear(vis_dim, hidden_size)
        self.input_v_patch = nn.Linear(patch_dim, hidden_size)
        self.input_qst = nn.Linear(qst_dim, hidden_size)
        self.input_qst_prompt = nn.Linear(qst_dim, hidden_size)
        self.avq_cross_attn = avq_cross_attn
        self.AV_Attn = AV_Attn(AVHanLayer(d_model=512, nhead=1, dim_feedforward=512), num_layers=1)
        self.TemporalPerception = TemporalPerception(topK)
        self.SpatioPerception = SpatioPerceptionModule(topK)
        self.QstTempGrd_Module = QstTemporalGrounding()
        self.av_fusion_fc = nn.Linear(3072, 512)
        self.answer_pred_fc = nn.Linear(512, 42)
        self.last = {}

    def forward(self, reshaped_data: Dict[str, Tensor]):
        ops.new_pass()                          # operand caches on activations / input tensors live for one pass only
        audio, visual, patch = reshaped_data['audio'], reshaped_data['video'], reshaped_data['patch']
        question, prompt = reshaped_data['quest'], reshaped_data['prompt']
        if not (torch.is_tensor(question) and question.is_floating_point()):
            raise RuntimeError('TSPM here takes float question / prompt features [B,1,qst_dim] (tspm.py:373-376); '
                               'encode tokens with encoders.CLIP_TEncoder first')
        question, prompt = question.squeeze(1), prompt.squeeze(1)
        a = ops.linear(audio, self.input_a.weight, self.input_a.bias)
        v = ops.linear(visual, self.input_v.weight, self.input_v.bias)
        qf = ops.linear(question, self.input_qst.weight, self.input_qst.bias)
        qp = ops.linear(prompt, self.input_qst_prompt.weight, self.input_qst_prompt.bias)
        a_av, v_av = self.AV_Attn(a, v)
        a_ts, v_ts, idx = self.TemporalPerception(a, v, qp)
        B = a.shape[0]
        ar = torch.arange(B, device=a.device)[:, None]
        p_sel = patch[ar, idx[:, 0]]                                 # gather first, project the K selected frames only
        p_sel = ops.linear(p_sel, self.input_v_patch.weight, self.input_v_patch.bias)
        v_sp = self.SpatioPerception(a_ts, p_sel)
        a_q, v_q = self.QstTempGrd_Module(qf, a_ts, v_sp)
        feat = torch.cat([a_q, a_av.mean(dim=-2), a_ts.mean(dim=-2), v_q, v_av.mean(dim=-2), v_sp.mean(dim=-2)], dim=-1)
        av = ops.linear(torch.tanh(feat), self.av_fusion_fc.weight, self.av_fusion_fc.bias)
        out = ops.linear(torch.tanh(av * qf), self.answer_pred_fc.weight, self.answer_pred_fc.bias)
        self.last = dict(top_k_index_sort=idx.detach(), audio_tssm=a_ts.detach(), visual_sp=v_sp.detach(), av_feat=av.detach())
        return dict(out=out)
