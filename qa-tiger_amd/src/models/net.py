"""`QA_TIGER` on MI355X kernels, behind the reference's module surface.

Mirrors reference src/models/net.py:20-144: same constructor keywords (extra
keys swallowed by **kwargs, net.py:31), same sub-module and parameter names
(so `best.pt` / the public checkpoint load with `load_state_dict(strict=False)`,
reference trainutils.py:94-101), same dict-in / `{'out': logits}`-out call.

Question input.  The reference accepts int tokens and runs the frozen CLIP text
tower; its float-feature branch crashes (`words=None` -> `words_proj(None)`,
net.py:94-97,122).  Here:
  * `reshaped_data['quest_feat']` [B,Dv] + `reshaped_data['words_feat']` [B,L,Dv]
    (pre-extracted text-tower outputs) are used directly when present — the
    float path that works;
  * otherwise `reshaped_data['quest']` goes through `self.quest_encoder`
    (see encoders.py) exactly as in the reference.
"""
from __future__ import annotations

import importlib
from typing import Dict

import torch
import torch.nn as nn
from torch import Tensor

_qt = importlib.import_module('qa-tiger_amd')
ops = _qt.ops
_mods = importlib.import_module('qa-tiger_amd.src.models.modules')
_enc = importlib.import_module('qa-tiger_amd.src.models.encoders')
Projection, QstGrounding, TempMoE = _mods.Projection, _mods.QstGrounding, _mods.TempMoE
AVQCrossAttn, PatchSelecter = _mods.AVQCrossAttn, _mods.PatchSelecter
CLIP_TEncoder = _enc.CLIP_TEncoder


class QA_TIGER(nn.Module):
    def __init__(self, d_model: int = 512, video_dim: int = 512, patch_dim: int = 768, audio_dim: int = 128,
                 topK: int = 3, num_experts: int = 10, late_fusion: bool = False, nce_loss: bool = False,
                 encoder_type: str = 'ViT-L/14@336px', text_tower=None, **kwargs):
        super().__init__()
        self.nce_loss = nce_loss
        self.late_fusion = late_fusion
        self.audio_proj = Projection(audio_dim, d_model)
        self.video_proj = Projection(video_dim, d_model)
        self.patch_proj = Projection(patch_dim, d_model)
        self.patch_proj.emit_image = True            # (not a parameter or buffer: its output goes straight into PatchSelecter's in-projection GEMM)
        self.words_proj = Projection(video_dim, d_model)
        self.quest_proj = Projection(video_dim, d_model)
        self.quest_encoder = CLIP_TEncoder(encoder_type, tower=text_tower)   # text_tower: None (features are handed in) | 'vitl14' | TextConfig
        self.quest_encoder.freeze()
        self.crs_attn = AVQCrossAttn(d_model, 8)
        self.patch_selecter = PatchSelecter(d_model, 8)
        self.quest_grounding = QstGrounding(d_model, 8)
        self.at_aggregator = TempMoE(d_model, 8, topK=topK, n_experts=num_experts)
        self.vt_aggregator = TempMoE(d_model, 8, topK=topK, n_experts=num_experts, vis_branch=True)
        self.head_act = nn.ReLU()
        self.dropout = nn.Dropout(0.1)          # defined, never applied (reference net.py:59, quirk Q9)
        self.head = nn.Linear(d_model, 42)
        for m in (self.audio_proj, self.video_proj, self.words_proj, self.quest_proj, self.patch_proj, self.head):
            m.apply(self.init_weight)

    def init_weight(self, m):
        if isinstance(m, nn.Linear):            # reference net.py:71-75
            nn.init.kaiming_normal_(m.weight)
            if m.bias is not None:
                nn.init.constant_(m.bias, 0)

    def sub_forward(self, reshaped_data: Dict[str, Tensor], prefix: str = ''):
        audio = reshaped_data[f'{prefix}audio']
        video = reshaped_data[f'{prefix}video']
        patch = reshaped_data.get(f'{prefix}patch')
        if prefix == 'n_':                      # reference net.py:89-91
            return None, None, audio, video, patch
        if reshaped_data.get(f'{prefix}quest_feat') is not None:
            quest, words = reshaped_data[f'{prefix}quest_feat'], reshaped_data[f'{prefix}words_feat']
        else:
            quest, words = self.quest_encoder(reshaped_data[f'{prefix}quest'])
        if quest.dim() == 3:
            quest = quest.squeeze(1)
        return quest, words, audio, video, patch

    def forward(self, reshaped_data: Dict[str, Tensor]):
        ops.new_pass()                          # operand images cached on activations (and on the INPUT tensors) live for one pass only
        quest, words, audio, video, patch = self.sub_forward(reshaped_data, prefix='')
        if patch is None:
            raise KeyError("QA_TIGER needs reshaped_data['patch'] [B,T,P,patch_dim] (reference net.py:124)")
        audio = self.audio_proj(audio)
        video = self.video_proj(video)
        words = self.words_proj(words)
        quest = ops.share_grad(self.quest_proj(quest))          # four Linear consumers (two aggregators, two grounding calls): one gradient buffer
        patch = self.patch_proj(patch)
        audio, video = self.crs_attn(audio, video, words)
        patch = self.patch_selecter(patch, audio, video)
        a_global = self.at_aggregator(quest, audio)
        ap_global, vp_global = self.vt_aggregator(quest, video, patch)
        fusion = self.quest_grounding(quest, [ap_global, vp_global])
        fusion = self.quest_grounding(quest, [fusion.unsqueeze(1), a_global], relu_out=True)   # + head_act (net.py:141)
        output = ops.linear(fusion, self.head.weight, self.head.bias)
        return {'out': output}
