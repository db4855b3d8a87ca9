"""Text tower of QA_TIGER (reference src/models/encoders.py:7-37).

The reference's `CLIP_TEncoder` is a frozen OpenAI-CLIP text transformer whose
weights are downloaded at construction time (reference src/models/clip.py:30-72)
— impossible offline.  This class keeps the slot, the state_dict names and the
call convention `encoder(tokens) -> (quest [B,E], words [B,L,W])` and offers

* the pre-extracted route used by the synthetic benchmark path: the two outputs
  are handed in (`reshaped_data['quest_feat']/['words_feat']`, or a dict/tuple);
* the real tower, forward only, on the HIP kernels:
  embedding gather + positions (`qt_text_embed_fwd`), 12 x [LN -> packed QKV GEMM
  -> causal attention (`qt_attn_causal_fwd`) -> out-proj GEMM with the residual
  in its epilogue -> LN -> c_fc GEMM with QuickGELU in its epilogue -> c_proj
  GEMM + residual], ln_final, EOT-row gather (`qt_text_eot_rows`) and the
  text_projection GEMM.  Build it with `CLIP_TEncoder(name, tower=TextConfig|'vitl14')`
  or `QA_TIGER(..., text_tower='vitl14')`, then `load_state_dict` a CLIP
  checkpoint's text-side entries (same names as the reference module); without a
  checkpoint it starts from `CLIP.initialize_parameters`-scaled random weights.
"""
from __future__ import annotations

import importlib
from typing import Optional, Union

import torch
import torch.nn as nn

_pkg = __name__.rsplit('.', 3)[0]
_ops = importlib.import_module(_pkg + '.ops')
_synth = importlib.import_module(_pkg + '.synth')


class CLIP_TEncoder(nn.Module):
    def __init__(self, model_name: str = 'ViT-L/14@336px', tower: Union[None, str, '_synth.TextConfig'] = None):
        super().__init__()
        self.model_name = model_name
        self.cfg: Optional[_synth.TextConfig] = None
        if tower is not None:
            cfg = _synth.TEXT_VITL14 if isinstance(tower, str) and tower.lower() in ('vitl14', 'vit-l/14', 'vit-l/14@336px') else tower
            if not isinstance(cfg, _synth.TextConfig):
                raise ValueError(f"tower must be a TextConfig or 'vitl14', got {tower!r}")
            self.cfg = cfg
            for name, t in _synth.make_text_state_dict(cfg, seed=713).items():
                self._register(name, t)

    def _register(self, name: str, t: torch.Tensor):
        """Create (nested) nn.Parameters under the reference's state_dict names."""
        mod = self
        parts = name.split('.')
        for p in parts[:-1]:
            if not hasattr(mod, p):
                mod.add_module(p, nn.Module())
            mod = getattr(mod, p)
        mod.register_parameter(parts[-1], nn.Parameter(t.clone(), requires_grad=False))

    def freeze(self):
        for p in self.parameters():
            p.requires_grad = False

    # ------------------------------------------------------------------ #
    @torch.no_grad()
    def encode(self, text: torch.Tensor):
        cfg = self.cfg
        sd = dict(self.named_parameters())
        B, L = text.shape
        W, H = cfg.width, cfg.heads
        mode = _ops.precision['fwd']
        x = _ops.text_embed(text, sd['token_embedding.weight'], sd['positional_embedding']).view(B * L, W)
        for i in range(cfg.layers):
            p = f'transformer.resblocks.{i}.'
            h = _ops.add_layer_norm([x], sd[p + 'ln_1.weight'], sd[p + 'ln_1.bias'])
            qkv = _ops.gemm(h, sd[p + 'attn.in_proj_weight'], bias=sd[p + 'attn.in_proj_bias'], mode=mode)
            o = _ops.causal_attention(qkv, B, L, H)
            x = _ops.gemm(o, sd[p + 'attn.out_proj.weight'], bias=sd[p + 'attn.out_proj.bias'], add=x, mode=mode)
            h = _ops.add_layer_norm([x], sd[p + 'ln_2.weight'], sd[p + 'ln_2.bias'])
            f = _ops.gemm(h, sd[p + 'mlp.c_fc.weight'], bias=sd[p + 'mlp.c_fc.bias'], relu=2, mode=mode)   # 2 = QuickGELU
            x = _ops.gemm(f, sd[p + 'mlp.c_proj.weight'], bias=sd[p + 'mlp.c_proj.bias'], add=x, mode=mode)
        x = _ops.add_layer_norm([x], sd['ln_final.weight'], sd['ln_final.bias'])
        eot, _ = _ops.text_eot_rows(text, x.view(B, L, W))
        quest = _ops.gemm(eot, sd['text_projection'], trans_b=False, mode=mode)
        return quest, x.view(B, L, W)

    def forward(self, text):
        if isinstance(text, dict):
            return text['quest_feat'], text['words_feat']
        if isinstance(text, (tuple, list)) and len(text) == 2:
            return text[0], text[1]
        if self.cfg is not None and torch.is_tensor(text) and not text.is_floating_point():
            return self.encode(text)
        raise RuntimeError(
            'QA_TIGER received raw question tokens, but this model was built without the frozen CLIP text tower '
            '(its checkpoint is a network download in the reference, src/models/clip.py:30-72). Either build it with '
            "QA_TIGER(..., text_tower='vitl14') and load the text-side CLIP weights, or pass the pre-extracted "
            "features reshaped_data['quest_feat'] [B,Dv] and reshaped_data['words_feat'] [B,L,Dv].")
