"""Deterministic synthetic weights and inputs for the QA-TIGER fusion path.

The reference ships neither features nor checkpoints, so every
parity fixture and the benchmark run on seeded synthetic tensors.  The recipe
below is *order independent*: each tensor is generated from a seed derived from
its own name, so the reference model (in tools/make_golden.py), the CPU oracle
and the HIP path can all regenerate bit-identical weights without sharing a
construction order or an RNG stream.

Shapes follow the input contract of `get_items` (reference
src/trainutils.py:223-250) and the parameter names/shapes of `QA_TIGER`
(reference src/models/net.py:20-68, src/models/modules.py).
"""
from __future__ import annotations

import zlib
from dataclasses import dataclass, asdict
from typing import Dict

import numpy as np
import torch

N_ANSWERS = 42  # reference src/models/net.py:60
N_HEADS = 8     # reference src/models/net.py:50-56


@dataclass(frozen=True)
class PathConfig:
    """Dimensions of one workload (symbols as in SURVEY.md, conventions table)."""
    B: int = 256          # batch per device
    T: int = 60           # frames
    D: int = 768          # d_model
    Dv: int = 768         # video_dim (also CLIP text width)
    Da: int = 128         # audio_dim
    Dp: int = 1024        # patch_dim
    P: int = 14           # patch tokens per frame
    L: int = 77           # CLIP text context
    K: int = 8            # num_experts
    k: int = 8            # topK
    H: int = N_HEADS
    A: int = N_ANSWERS

    def model_kwargs(self) -> dict:
        """kwargs for `QA_TIGER(**cfg.hyper_params.model)` (reference trainutils.py:86-87)."""
        return dict(d_model=self.D, video_dim=self.Dv, patch_dim=self.Dp,
                    audio_dim=self.Da, topK=self.k, num_experts=self.K)

    def asdict(self) -> dict:
        return asdict(self)


# BASELINE.json configs (SURVEY.md §8d "configs -> shapes")
CONFIG_1 = PathConfig(B=4, T=10, D=512, Dv=512, Da=128, Dp=1024, P=14, L=77, K=4, k=4)
CONFIG_2 = PathConfig(B=256, T=60, D=768, Dv=768, Da=128, Dp=1024, P=14, L=77, K=8, k=8)
CONFIG_4 = PathConfig(B=128, T=300, D=1024, Dv=1024, Da=128, Dp=1024, P=14, L=77, K=16, k=16)
# small shapes used by the golden fixtures (everything committed)
CONFIG_TOY = PathConfig(B=5, T=7, D=64, Dv=48, Da=24, Dp=40, P=3, L=6, K=6, k=3)
CONFIG_TOY_KK = PathConfig(B=9, T=6, D=64, Dv=64, Da=16, Dp=32, P=4, L=5, K=4, k=4)


def _seed(name: str, base_seed: int) -> int:
    return (zlib.crc32(name.encode()) ^ (base_seed * 0x9E3779B1)) & 0xFFFFFFFF


def _normal(name: str, shape, base_seed: int) -> np.ndarray:
    rng = np.random.Generator(np.random.PCG64(_seed(name, base_seed)))
    return rng.standard_normal(size=tuple(shape), dtype=np.float32)


def param_shapes(cfg: PathConfig) -> Dict[str, tuple]:
    """Trainable state_dict names -> shapes.

    `quest_encoder.*` (frozen CLIP text tower, reference encoders.py:7-37) is
    outside the synthetic path and not listed.
    """
    D, Dh, K = cfg.D, cfg.D // 2, cfg.K
    s: Dict[str, tuple] = {}

    def lin(prefix, out_f, in_f):
        s[f'{prefix}.weight'] = (out_f, in_f)
        s[f'{prefix}.bias'] = (out_f,)

    def mha(prefix):
        s[f'{prefix}.in_proj_weight'] = (3 * D, D)
        s[f'{prefix}.in_proj_bias'] = (3 * D,)
        lin(f'{prefix}.out_proj', D, D)

    def ln(prefix):
        s[f'{prefix}.weight'] = (D,)
        s[f'{prefix}.bias'] = (D,)

    lin('audio_proj.proj', D, cfg.Da)
    lin('video_proj.proj', D, cfg.Dv)
    lin('patch_proj.proj', D, cfg.Dp)
    lin('words_proj.proj', D, cfg.Dv)
    lin('quest_proj.proj', D, cfg.Dv)
    for a in ('qst_attn', 'crs_attn', 'slf_attn'):
        mha(f'crs_attn.{a}')
    lin('crs_attn.linear1', D, D)
    lin('crs_attn.linear2', D, D)
    ln('crs_attn.norm1')
    ln('crs_attn.norm2')
    ln('patch_selecter.vnorm')
    ln('patch_selecter.anorm')
    mha('patch_selecter.slf_attn')
    mha('patch_selecter.crs_attn')
    lin('patch_selecter.mlp.0', Dh, D)
    lin('patch_selecter.mlp.2', D, Dh)
    ln('quest_grounding.norm')
    mha('quest_grounding.attn')
    lin('quest_grounding.mlp.0', Dh, D)
    lin('quest_grounding.mlp.2', D, Dh)
    for agg, norms in (('at_aggregator', ('norm',)), ('vt_aggregator', ('anorm', 'vnorm'))):
        for n in norms:
            ln(f'{agg}.{n}')
        mha(f'{agg}.qst_attn')
        lin(f'{agg}.gauss_pred.0', 2 * K, D)
        lin(f'{agg}.router.0', K, D)
        for e in range(K):
            lin(f'{agg}.experts.{e}.0', Dh, D)
            lin(f'{agg}.experts.{e}.2', D, Dh)
    lin('head', cfg.A, D)
    return s


def make_state_dict(cfg: PathConfig, seed: int = 713) -> Dict[str, torch.Tensor]:
    """Recipe weights.  Unlike the reference initialisers (zero biases, unit
    LayerNorm gains) every tensor is non-trivial so that bias / gain / shift
    code paths are exercised by the parity tests."""
    out: Dict[str, torch.Tensor] = {}
    for name, shape in param_shapes(cfg).items():
        z = _normal(name, shape, seed)
        is_norm = ('norm' in name.split('.')[-2])
        if len(shape) >= 2:
            w = z * np.float32(np.sqrt(2.0 / shape[1]) if 'in_proj' not in name
                               else np.sqrt(1.0 / shape[1]))
        elif is_norm and name.endswith('.weight'):
            w = np.float32(1.0) + np.float32(0.1) * z
        elif is_norm:
            w = np.float32(0.05) * z
        else:
            w = np.float32(0.02) * z
        out[name] = torch.from_numpy(np.ascontiguousarray(w, dtype=np.float32))
    return out


def make_inputs(cfg: PathConfig, seed: int = 713, batch_offset: int = 0) -> Dict[str, torch.Tensor]:
    """Synthetic batch in the layout `get_items` hands to the model
    (reference trainutils.py:224-236) plus the injected text-encoder outputs
    (`quest_feat`, `words_feat`) that stand in for the frozen CLIP tower.

    `batch_offset` shifts the per-sample seeds so different ranks/shards draw
    different samples (sample i of the global batch is always the same tensor).
    """
    def per_sample(name, shape):
        rows = [_normal(f'{name}[{batch_offset + b}]', shape, seed) for b in range(cfg.B)]
        return torch.from_numpy(np.stack(rows, 0))

    rng = np.random.Generator(np.random.PCG64(_seed('label', seed)))
    labels_all = rng.integers(0, cfg.A, size=(batch_offset + cfg.B,), dtype=np.int64)
    return dict(
        audio=per_sample('audio', (cfg.T, cfg.Da)),
        video=per_sample('video', (cfg.T, cfg.Dv)),
        patch=per_sample('patch', (cfg.T, cfg.P, cfg.Dp)),
        quest_feat=per_sample('quest', (cfg.Dv,)),
        words_feat=per_sample('words', (cfg.L, cfg.Dv)),
        label=torch.from_numpy(labels_all[batch_offset:batch_offset + cfg.B].copy()),
    )


def fast_device_inputs(cfg: PathConfig, device, seed: int = 713) -> Dict[str, torch.Tensor]:
    """Benchmark-only inputs generated directly on the device (same shapes and
    N(0,1) distribution as `make_inputs`; not bit-identical to it)."""
    g = torch.Generator(device=device)
    g.manual_seed(seed)

    def rn(*shape):
        return torch.randn(*shape, generator=g, device=device, dtype=torch.float32)

    return dict(
        audio=rn(cfg.B, cfg.T, cfg.Da), video=rn(cfg.B, cfg.T, cfg.Dv),
        patch=rn(cfg.B, cfg.T, cfg.P, cfg.Dp), quest_feat=rn(cfg.B, cfg.Dv),
        words_feat=rn(cfg.B, cfg.L, cfg.Dv),
        label=torch.randint(0, cfg.A, (cfg.B,), generator=g, device=device),
    )


# --------------------------------------------------------------------------- #
# frozen CLIP text tower: shapes, recipe weights, token batches
# --------------------------------------------------------------------------- #
@dataclass(frozen=True)
class TextConfig:
    """Text-transformer hyper-parameters as `build_model` derives them from a CLIP
    checkpoint (reference src/models/base/clip_base.py:479-484)."""
    width: int = 768
    layers: int = 12
    heads: int = 12           # width // 64
    context_length: int = 77
    vocab_size: int = 49408
    embed_dim: int = 768      # text_projection output = QA_TIGER's video_dim

    def asdict(self):
        return dict(width=self.width, layers=self.layers, heads=self.heads, context_length=self.context_length,
                    vocab_size=self.vocab_size, embed_dim=self.embed_dim)


TEXT_VITL14 = TextConfig()                                           # ViT-L/14@336px text side
TEXT_TOY = TextConfig(width=128, layers=2, heads=2, context_length=77, vocab_size=512, embed_dim=96)


def text_param_shapes(cfg: TextConfig) -> Dict[str, tuple]:
    """state_dict names of the reference `CLIP_TEncoder` (encoders.py:14-22) -> shapes."""
    W = cfg.width
    s: Dict[str, tuple] = {'token_embedding.weight': (cfg.vocab_size, W),
                           'positional_embedding': (cfg.context_length, W)}
    for i in range(cfg.layers):
        p = f'transformer.resblocks.{i}.'
        s[p + 'attn.in_proj_weight'] = (3 * W, W)
        s[p + 'attn.in_proj_bias'] = (3 * W,)
        s[p + 'attn.out_proj.weight'] = (W, W)
        s[p + 'attn.out_proj.bias'] = (W,)
        s[p + 'ln_1.weight'] = (W,)
        s[p + 'ln_1.bias'] = (W,)
        s[p + 'mlp.c_fc.weight'] = (4 * W, W)
        s[p + 'mlp.c_fc.bias'] = (4 * W,)
        s[p + 'mlp.c_proj.weight'] = (W, 4 * W)
        s[p + 'mlp.c_proj.bias'] = (W,)
        s[p + 'ln_2.weight'] = (W,)
        s[p + 'ln_2.bias'] = (W,)
    s['ln_final.weight'] = (W,)
    s['ln_final.bias'] = (W,)
    s['text_projection'] = (W, cfg.embed_dim)
    s['logit_scale'] = ()
    return s


def make_text_state_dict(cfg: TextConfig, seed: int = 713) -> Dict[str, torch.Tensor]:
    """Recipe weights for the text tower: the scales of `CLIP.initialize_parameters`
    (clip_base.py:361-388) with non-trivial biases and LayerNorm gains, name-seeded."""
    W = cfg.width
    proj_std = (W ** -0.5) * ((2 * cfg.layers) ** -0.5)
    out: Dict[str, torch.Tensor] = {}
    for name, shape in text_param_shapes(cfg).items():
        if name == 'logit_scale':
            out[name] = torch.tensor(float(np.log(1 / 0.07)), dtype=torch.float32)
            continue
        z = _normal('text.' + name, shape, seed)
        leaf = name.split('.')[-1]
        if name == 'token_embedding.weight':
            w = np.float32(0.02) * z
        elif name == 'positional_embedding':
            w = np.float32(0.01) * z
        elif name == 'text_projection' or leaf == 'in_proj_weight':
            w = np.float32(W ** -0.5) * z
        elif name.endswith('out_proj.weight') or name.endswith('c_proj.weight'):
            w = np.float32(proj_std) * z
        elif name.endswith('c_fc.weight'):
            w = np.float32((2 * W) ** -0.5) * z
        elif '.ln_' in name or name.startswith('ln_final'):
            w = (np.float32(1.0) + np.float32(0.1) * z) if leaf == 'weight' else np.float32(0.05) * z
        else:                                                          # Linear / in_proj biases
            w = np.float32(0.02) * z
        out[name] = torch.from_numpy(np.ascontiguousarray(w, dtype=np.float32))
    return out


def make_text_tokens(cfg: TextConfig, B: int, seed: int = 713) -> torch.Tensor:
    """int32 [B, context_length] batches shaped like CLIP's tokenizer output: start token,
    3..30 word-piece ids, end token (the largest id, so argmax finds it), zero padding."""
    rng = np.random.Generator(np.random.PCG64(_seed('text.tokens', seed)))
    sot, eot = cfg.vocab_size - 2, cfg.vocab_size - 1
    out = np.zeros((B, cfg.context_length), dtype=np.int32)
    for b in range(B):
        n = int(rng.integers(3, min(31, cfg.context_length - 2)))
        out[b, 0] = sot
        out[b, 1:1 + n] = rng.integers(1, cfg.vocab_size - 2, size=n)
        out[b, 1 + n] = eot
    return torch.from_numpy(out)


# --------------------------------------------------------------------------- #
# TSPM baseline model: shapes, recipe weights, inputs — oracle and goldens exist, HIP side next
# --------------------------------------------------------------------------- #
@dataclass(frozen=True)
class TspmConfig:
    """Workload of the reference's second model, `TSPM(**cfg.hyper_params.model)` (src/models/tspm.py:309-345,
    configs/tspm/vitl14.py:37-45).  The hidden width is 512 in every sub-module (hard-coded there)."""
    B: int = 32
    T: int = 60
    P: int = 14
    Da: int = 128
    Dv: int = 768
    Dp: int = 1024
    Dq: int = 768
    topK: int = 10
    C: int = 512
    A: int = 42

    def asdict(self):
        return dict(B=self.B, T=self.T, P=self.P, Da=self.Da, Dv=self.Dv, Dp=self.Dp, Dq=self.Dq, topK=self.topK, C=self.C, A=self.A)

    def model_kwargs(self):
        return dict(topK=self.topK, avq_cross_attn=False, audio_dim=self.Da, vis_dim=self.Dv, patch_dim=self.Dp,
                    qst_dim=self.Dq, hidden_size=self.C)


TSPM_REF = TspmConfig()
TSPM_TOY = TspmConfig(B=3, T=12, P=3, Da=32, Dv=48, Dp=40, Dq=56, topK=4)


def tspm_param_shapes(cfg: TspmConfig) -> Dict[str, tuple]:
    C = cfg.C
    s: Dict[str, tuple] = {}

    def lin(name, o, i):
        s[name + '.weight'] = (o, i); s[name + '.bias'] = (o,)

    def mha(name):
        s[name + '.in_proj_weight'] = (3 * C, C); s[name + '.in_proj_bias'] = (3 * C,)
        lin(name + '.out_proj', C, C)

    def ln(name):
        s[name + '.weight'] = (C,); s[name + '.bias'] = (C,)

    lin('input_a', C, cfg.Da); lin('input_v', C, cfg.Dv); lin('input_v_patch', C, cfg.Dp)
    lin('input_qst', C, cfg.Dq); lin('input_qst_prompt', C, cfg.Dq)
    p = 'AV_Attn.layers.0'
    mha(p + '.self_attn'); mha(p + '.cm_attn'); lin(p + '.linear1', C, C); lin(p + '.linear2', C, C); ln(p + '.norm1'); ln(p + '.norm2')
    ln('AV_Attn.norm1'); ln('AV_Attn.norm2')                       # defined, never applied (norm=None, tspm.py:71-73)
    for m in ('TemporalPerception', 'SpatioPerception'):
        mha(m + '.attn_qst_query'); lin(m + '.qst_query_linear1', C, C); lin(m + '.qst_query_linear2', C, C); ln(m + '.qst_query_visual_norm')
        if m == 'SpatioPerception':
            t = m + '.TokensAttn'
            mha(t + '.self_attn'); lin(t + '.linear1', C, C); lin(t + '.linear2', C, C); ln(t + '.norm1'); ln(t + '.norm2')
    m = 'QstTempGrd_Module'
    mha(m + '.attn_qst_query'); lin(m + '.qst_query_linear1', C, C); lin(m + '.qst_query_linear2', C, C); ln(m + '.qst_query_visual_norm')
    lin('av_fusion_fc', C, 6 * C); lin('answer_pred_fc', cfg.A, C)
    return s


def make_tspm_state_dict(cfg: TspmConfig, seed: int = 713) -> Dict[str, torch.Tensor]:
    out: Dict[str, torch.Tensor] = {}
    for name, shape in tspm_param_shapes(cfg).items():
        z = _normal('tspm.' + name, shape, seed)
        is_norm = 'norm' in name.split('.')[-2]
        if len(shape) == 2:
            w = z * np.float32(np.sqrt((1.0 if 'in_proj' in name else 2.0) / shape[1]))
        elif is_norm and name.endswith('.weight'):
            w = np.float32(1.0) + np.float32(0.1) * z
        elif is_norm:
            w = np.float32(0.05) * z
        else:
            w = np.float32(0.02) * z
        out[name] = torch.from_numpy(np.ascontiguousarray(w, dtype=np.float32))
    return out


def make_tspm_inputs(cfg: TspmConfig, seed: int = 713) -> Dict[str, torch.Tensor]:
    """`reshaped_data` as TSPM.forward reads it (tspm.py:355-357): float question / prompt features [B,1,Dq]."""
    def per_sample(name, shape):
        return torch.from_numpy(np.stack([_normal(f'tspm.{name}[{b}]', shape, seed) for b in range(cfg.B)], 0))
    rng = np.random.Generator(np.random.PCG64(_seed('tspm.label', seed)))
    return dict(audio=per_sample('audio', (cfg.T, cfg.Da)), video=per_sample('video', (cfg.T, cfg.Dv)),
                patch=per_sample('patch', (cfg.T, cfg.P, cfg.Dp)), quest=per_sample('quest', (1, cfg.Dq)),
                prompt=per_sample('prompt', (1, cfg.Dq)), label=torch.from_numpy(rng.integers(0, cfg.A, size=(cfg.B,), dtype=np.int64)))
