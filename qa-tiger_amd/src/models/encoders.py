"""Text-tower slot of QA_TIGER (reference src/models/encoders.py:7-37).

The reference's `CLIP_TEncoder` is a frozen OpenAI-CLIP text transformer whose
weights are downloaded at construction time (reference src/models/clip.py:30-72)
— impossible offline, and outside the fusion hot path (SURVEY.md §2 row 3,
§8f rank 1).  This class keeps the slot and the call convention
`encoder(x) -> (quest [B,Dv], words [B,L,Dv])` but takes the two outputs
pre-extracted.  A real text tower (any nn.Module with that return convention)
can be plugged in with `model.quest_encoder = tower`.
"""
from __future__ import annotations

import torch
import torch.nn as nn


class CLIP_TEncoder(nn.Module):
    def __init__(self, model_name: str = 'ViT-L/14@336px'):
        super().__init__()
        self.model_name = model_name

    def freeze(self):
        for p in self.parameters():
            p.requires_grad = False

    def forward(self, text):
        if isinstance(text, dict):
            return text['quest_feat'], text['words_feat']
        if isinstance(text, (tuple, list)) and len(text) == 2:
            return text[0], text[1]
        raise RuntimeError(
            'QA_TIGER received raw question tokens, but the frozen CLIP text tower is not part of this build '
            '(its checkpoint is a network download in the reference, src/models/clip.py:30-72). Pass the '
            "pre-extracted text features instead: reshaped_data['quest_feat'] [B,Dv] and "
            "reshaped_data['words_feat'] [B,L,Dv], or assign a text tower to model.quest_encoder.")
