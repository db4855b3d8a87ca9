"""CPU ORACLE for the QA-TIGER question-aware Gaussian-expert fusion path.

*** TEST INFRASTRUCTURE ONLY. ***  Only `tests/`, `__graft_entry__.smoke()` and
the `cpu_baseline` leg of `bench.py` may import this file.  The product
(`qa-tiger_amd/`) never does: its forward/backward run on hand-written HIP
kernels and fail loudly without them.

What this is: a repo-owned restatement, in plain PyTorch CPU tensor algebra, of
the algorithm executed by the reference `QA_TIGER.forward`
(/root/reference/src/models/net.py:106-144) and the blocks it composes
(/root/reference/src/models/modules.py).  It follows the reference's *order of
operations* (K/V projections materialised, all K experts run on every token,
`(T,B,K,D)` expert stack gathered with the reference's row/index mismatch), so
it is also an honest CPU timing baseline for the reference's execution plan.
`nn.MultiheadAttention` (third-party, torch==2.4.0 per reference
pyproject.toml:26) is restated from its documented math path
(SURVEY.md Appendix C): packed in-projection, q scaled by 1/sqrt(d_h) after the
bias, softmax(QK^T), dropout on probabilities, ·V, head concat, out-projection.

Pinning: the reference has no tests.  This oracle is pinned by
golden vectors produced by importing the reference itself in the build
container (tools/make_golden.py -> tests/golden/*.npz), checked in
tests/test_oracle_golden.py.

Parameters are passed as a flat mapping `P[name] -> Tensor` using the
reference's state_dict names.  Gradients come from
torch autograd over these functions (eval mode).
"""
from __future__ import annotations

import math
from typing import Dict, List, Mapping, Optional, Sequence, Tuple

import torch
import torch.nn.functional as F

Tensor = torch.Tensor
N_HEADS = 8  # reference net.py:50-56
SIGMA = 9.0  # reference modules.py:361
LN_EPS = 1e-5


class _Ctx:
    """Execution options: training-mode dropout (timing baseline only) and a tap
    for intermediates."""

    def __init__(self, train: bool = False, p: float = 0.1, taps: Optional[dict] = None):
        self.train = train
        self.p = p
        self.taps = taps

    def drop(self, x: Tensor) -> Tensor:
        return F.dropout(x, self.p, True) if self.train else x

    def tap(self, name: str, x):
        if self.taps is not None:
            self.taps[name] = x


def linear(P: Mapping[str, Tensor], prefix: str, x: Tensor) -> Tensor:
    """nn.Linear: y = x W^T + b (reference modules.py:33,43)."""
    return x @ P[f'{prefix}.weight'].t() + P[f'{prefix}.bias']


def layer_norm(P: Mapping[str, Tensor], prefix: str, x: Tensor) -> Tensor:
    mu = x.mean(-1, keepdim=True)
    var = ((x - mu) ** 2).mean(-1, keepdim=True)
    return (x - mu) / torch.sqrt(var + LN_EPS) * P[f'{prefix}.weight'] + P[f'{prefix}.bias']


def mha(P: Mapping[str, Tensor], prefix: str, xq: Tensor, xkv: Tensor, ctx: _Ctx,
        H: int = N_HEADS, weights_out: Optional[list] = None) -> Tensor:
    """nn.MultiheadAttention math path, batch-first restatement.

    xq [N,Lq,E], xkv [N,Lk,E] -> [N,Lq,E].  key == value at every call site of
    the path (reference modules.py:217-221, 338, 562, 688, 698).
    `weights_out` (a list) receives the module's second return value: the mean over heads of the
    (post-dropout) probabilities, [N,Lq,Lk] (need_weights=True default).
    """
    E = xq.shape[-1]
    dh = E // H
    W, bI = P[f'{prefix}.in_proj_weight'], P[f'{prefix}.in_proj_bias']
    q = xq @ W[:E].t() + bI[:E]
    k = xkv @ W[E:2 * E].t() + bI[E:2 * E]
    v = xkv @ W[2 * E:].t() + bI[2 * E:]
    N, Lq, Lk = xq.shape[0], xq.shape[1], xkv.shape[1]
    q = (q * (1.0 / math.sqrt(dh))).reshape(N, Lq, H, dh).transpose(1, 2)
    k = k.reshape(N, Lk, H, dh).transpose(1, 2)
    v = v.reshape(N, Lk, H, dh).transpose(1, 2)
    a = torch.softmax(q @ k.transpose(-1, -2), dim=-1)
    a = ctx.drop(a)
    if weights_out is not None:
        weights_out.append(a.mean(dim=1))
    o = (a @ v).transpose(1, 2).reshape(N, Lq, E)
    return o @ P[f'{prefix}.out_proj.weight'].t() + P[f'{prefix}.out_proj.bias']


# --------------------------------------------------------------------------- #
# AVQCrossAttn (reference modules.py:159-265)
# --------------------------------------------------------------------------- #
def avq_direction(P, src: Tensor, oth: Tensor, words: Tensor, ctx: _Ctx, weights_out: Optional[list] = None) -> Tensor:
    """One direction of AVQCrossAttn.sub_forward (reference modules.py:194-235); `weights_out` receives the
    question-attention map the reference returns beside it (modules.py:217, 236)."""
    pre = 'crs_attn'
    a_q = mha(P, f'{pre}.qst_attn', src, words, ctx, weights_out=weights_out)
    a_s = mha(P, f'{pre}.slf_attn', src, src, ctx)
    a_c = mha(P, f'{pre}.crs_attn', src, oth, ctx)
    x = layer_norm(P, f'{pre}.norm1', src + ctx.drop(a_s) + ctx.drop(a_c) + ctx.drop(a_q))
    f = linear(P, f'{pre}.linear2', ctx.drop(torch.relu(linear(P, f'{pre}.linear1', x))))
    return layer_norm(P, f'{pre}.norm2', x + ctx.drop(f))


def avq_cross_attn(P, audio: Tensor, video: Tensor, words: Tensor, ctx: _Ctx, visualize: bool = False):
    """Both directions share weights and both read the ORIGINAL inputs
    (reference modules.py:257-259).  visualize=True: also [a_weight, v_weight]
    (modules.py:261-263)."""
    if visualize:
        w = []
        a, v = avq_direction(P, audio, video, words, ctx, w), avq_direction(P, video, audio, words, ctx, w)
        return a, v, w
    return avq_direction(P, audio, video, words, ctx), avq_direction(P, video, audio, words, ctx)


# --------------------------------------------------------------------------- #
# PatchSelecter (reference modules.py:615-716)
# --------------------------------------------------------------------------- #
def patch_selecter(P, patch: Tensor, audio: Tensor, video: Tensor, ctx: _Ctx):
    """patch [B,T,Pn,D]; audio/video [B,T,D] -> [a_patch, v_patch] each [B,T,D].

    Query row 0 is video, row 1 is audio; the return order is [audio, video]
    (reference modules.py:692,706,713-716)."""
    pre = 'patch_selecter'
    B, T, Pn, D = patch.shape
    p = patch.reshape(B * T, Pn, D)
    p = p + mha(P, f'{pre}.slf_attn', p, p, ctx)            # no norm (modules.py:688)
    q = torch.stack([video.reshape(B * T, D), audio.reshape(B * T, D)], dim=1)  # [BT,2,D]
    o = mha(P, f'{pre}.crs_attn', q, p, ctx)
    m = linear(P, f'{pre}.mlp.2', torch.relu(linear(P, f'{pre}.mlp.0', ctx.drop(o))))
    v_rel, a_rel = m[:, 0].reshape(B, T, D), m[:, 1].reshape(B, T, D)
    return layer_norm(P, f'{pre}.anorm', a_rel), layer_norm(P, f'{pre}.vnorm', v_rel)


# --------------------------------------------------------------------------- #
# TempMoE (reference modules.py:350-612)
# --------------------------------------------------------------------------- #
def expert_centers(K: int, dtype, device) -> Tuple[Tensor, float]:
    """margin = 1/(2K); centers = linspace(margin, 1-margin, K) (modules.py:405-407)."""
    margin = 1.0 / (K * 2)
    return torch.linspace(margin, 1 - margin, K, dtype=dtype, device=device), margin


def generate_gaussian(pred: Tensor, topk_inds: Tensor, T: int, K: int, sigma: float = SIGMA) -> Tensor:
    """pred [B,K,2] = (tanh*margin offset, sigmoid width); topk_inds [B,k] -> [B,k,T].

    Restates reference modules.py:420-471: centre clamp to [0,1], width floor
    0.09 then /sigma, N(t|c,w) on linspace(0,1,T), divided by its discrete max
    over t (quirk Q6).  Uses the sample's OWN topk_inds here."""
    centers, _ = expert_centers(K, pred.dtype, pred.device)
    c_all = centers[None, :] + pred[:, :, 0]
    c = torch.gather(c_all, 1, topk_inds).clamp(0, 1)[:, :, None]            # [B,k,1]
    w = (torch.gather(pred[:, :, 1], 1, topk_inds).clamp(min=0.09) / sigma)[:, :, None]
    tau = torch.linspace(0, 1, T, dtype=pred.dtype, device=pred.device)[None, None, :]
    g = 0.3989422804014327 / w * torch.exp(-(tau - c) ** 2 / (2 * w ** 2))
    return g / g.max(dim=-1, keepdim=True)[0]


def moe_combine(experts_out: Tensor, gauss: Tensor, topk_inds: Tensor, topk_probs: Tensor) -> Tensor:
    """experts_out [B,T,K,C] (all experts on all tokens) -> [B,1,C].

    Restates reference modules.py:473-534 INCLUDING its index mismatch
    (quirk Q1): tokens are flattened as row = b*T + t (modules.py:495) while the
    index table is `topk_inds.repeat(T, 1)` whose row r holds sample r % B
    (modules.py:502).  So token (b,t), slot i takes expert
    topk_inds[(b*T+t) % B, i] but is weighted with gauss[b,i,t], topk_probs[b,i].
    """
    B, T, K, C = experts_out.shape
    k = topk_inds.shape[1]
    rows = torch.arange(B * T, device=experts_out.device)
    e_idx = topk_inds[rows % B]                                               # [BT,k]
    flat = experts_out.reshape(B * T, K, C)
    sel = torch.take_along_dim(flat, e_idx[:, :, None].expand(B * T, k, C), dim=1)
    sel = sel.reshape(B, T, k, C)
    per_slot = torch.einsum('bit,btic->bic', gauss, sel)                      # Σ_t g·sel
    return torch.einsum('bi,bic->bc', topk_probs, per_slot)[:, None, :]


def temp_moe(P, prefix: str, qst: Tensor, data: Tensor, sub_data: Optional[Sequence[Tensor]],
             K: int, k: int, ctx: _Ctx, sigma: float = SIGMA):
    """TempMoE.forward (reference modules.py:536-612).

    qst [B,D]; data [B,T,D]; sub_data None or [a_patch, v_patch]."""
    B, T, C = data.shape
    temp_w = mha(P, f'{prefix}.qst_attn', qst[:, None, :], data, ctx)[:, 0]   # [B,C]
    router_logits = linear(P, f'{prefix}.router.0', temp_w)
    probs = torch.softmax(router_logits, dim=-1)
    topk_probs, topk_inds = torch.topk(probs, k, dim=-1)                      # sorted desc (Q7)
    topk_probs = topk_probs / topk_probs.sum(-1, keepdim=True)
    raw = linear(P, f'{prefix}.gauss_pred.0', temp_w).view(B, K, 2)
    _, margin = expert_centers(K, raw.dtype, raw.device)
    pred = torch.stack([torch.tanh(raw[:, :, 0]) * margin, torch.sigmoid(raw[:, :, 1])], dim=-1)
    gauss = generate_gaussian(pred, topk_inds, T, K, sigma)
    ctx.tap(f'{prefix}.temp_w', temp_w)
    ctx.tap(f'{prefix}.router_logits', router_logits)
    ctx.tap(f'{prefix}.topk_inds', topk_inds)
    ctx.tap(f'{prefix}.topk_probs', topk_probs)
    ctx.tap(f'{prefix}.gauss', gauss)

    def run_experts(x: Tensor) -> Tensor:
        outs = [linear(P, f'{prefix}.experts.{e}.2',
                       torch.relu(linear(P, f'{prefix}.experts.{e}.0', x))) for e in range(K)]
        return torch.stack(outs, dim=2)                                       # [B,T,K,C]

    if sub_data is not None:
        a_out = moe_combine(run_experts(data + sub_data[0]), gauss, topk_inds, topk_probs)
        v_out = moe_combine(run_experts(data + sub_data[1]), gauss, topk_inds, topk_probs)
        return layer_norm(P, f'{prefix}.anorm', a_out), layer_norm(P, f'{prefix}.vnorm', v_out)
    out = moe_combine(run_experts(data), gauss, topk_inds, topk_probs)
    return layer_norm(P, f'{prefix}.norm', out)


# --------------------------------------------------------------------------- #
# QstGrounding (reference modules.py:268-347)
# --------------------------------------------------------------------------- #
def qst_grounding(P, qst: Tensor, data: List[Tensor], ctx: _Ctx) -> Tensor:
    """data: list of [B,S_i,D] concatenated along the sequence axis."""
    pre = 'quest_grounding'
    d = torch.cat(data, dim=1)                                                # [B,S,D]
    a = mha(P, f'{pre}.attn', qst[:, None, :], d, ctx)[:, 0]
    m = linear(P, f'{pre}.mlp.2', torch.relu(linear(P, f'{pre}.mlp.0', a)))
    return layer_norm(P, f'{pre}.norm', d.mean(dim=1) + ctx.drop(m))


# --------------------------------------------------------------------------- #
# QA_TIGER.forward (reference net.py:106-144)
# --------------------------------------------------------------------------- #
def qa_tiger_forward(P: Mapping[str, Tensor], inputs: Mapping[str, Tensor], K: int, k: int,
                     train: bool = False, taps: Optional[dict] = None, head_gate: Optional[Tensor] = None) -> Tensor:
    """inputs: audio [B,T,Da], video [B,T,Dv], patch [B,T,P,Dp], quest_feat [B,Dv],
    words_feat [B,L,Dv] (the two outputs of the frozen text tower, injected).
    Returns logits [B,42].
    head_gate (tests only): a 0/1 mask [B,D] applied INSTEAD of the ReLU in front of the head (net.py:141).  The forward value is
    continuous in a pre-activation at ~0 but the gradient is not; a test that compares gradients on a batch with a unit at its
    threshold hands in the gate state the kernels took, after checking that it differs from the oracle's own only at such units."""
    ctx = _Ctx(train=train, taps=taps)
    audio = linear(P, 'audio_proj.proj', inputs['audio'])
    video = linear(P, 'video_proj.proj', inputs['video'])
    words = linear(P, 'words_proj.proj', inputs['words_feat'])
    quest = linear(P, 'quest_proj.proj', inputs['quest_feat'])
    patch = linear(P, 'patch_proj.proj', inputs['patch'])
    audio, video = avq_cross_attn(P, audio, video, words, ctx)
    ctx.tap('crs_attn.audio', audio)
    ctx.tap('crs_attn.video', video)
    a_patch, v_patch = patch_selecter(P, patch, audio, video, ctx)
    ctx.tap('patch_selecter.a', a_patch)
    ctx.tap('patch_selecter.v', v_patch)
    a_global = temp_moe(P, 'at_aggregator', quest, audio, None, K, k, ctx)
    ap_global, vp_global = temp_moe(P, 'vt_aggregator', quest, video, [a_patch, v_patch], K, k, ctx)
    ctx.tap('at_aggregator.out', a_global)
    ctx.tap('vt_aggregator.a_out', ap_global)
    ctx.tap('vt_aggregator.v_out', vp_global)
    fusion = qst_grounding(P, quest, [ap_global, vp_global], ctx)
    ctx.tap('quest_grounding.1', fusion)
    fusion = qst_grounding(P, quest, [fusion[:, None, :], a_global], ctx)
    ctx.tap('quest_grounding.2', fusion)
    act = torch.relu(fusion) if head_gate is None else fusion * head_gate.to(fusion.dtype)
    return linear(P, 'head', act)                                             # net.py:141-142


def cross_entropy(logits: Tensor, label: Tensor) -> Tensor:
    """nn.CrossEntropyLoss, mean reduction (reference train.py:41, trainutils.py:278)."""
    lse = torch.logsumexp(logits, dim=-1)
    return (lse - logits.gather(1, label[:, None])[:, 0]).mean()


def adam_step(params: Dict[str, Tensor], grads: Dict[str, Tensor], m: Dict[str, Tensor],
              v: Dict[str, Tensor], step: int, lr=1e-4, betas=(0.95, 0.999), eps=1e-8):
    """torch.optim.Adam, weight_decay 0 (reference trainutils.py:141-144,
    configs/qa-tiger/vitl14.py:49-55).  In place; `step` is 1-based."""
    b1, b2 = betas
    for n, p in params.items():
        g = grads[n]
        m[n].mul_(b1).add_(g, alpha=1 - b1)
        v[n].mul_(b2).addcmul_(g, g, value=1 - b2)
        denom = (v[n].sqrt() / math.sqrt(1 - b2 ** step)).add_(eps)
        p.addcdiv_(m[n], denom, value=-lr / (1 - b1 ** step))


def forward_backward(P: Dict[str, Tensor], inputs: Mapping[str, Tensor], K: int, k: int,
                     train: bool = False, taps: Optional[dict] = None, head_gate: Optional[Tensor] = None):
    """One `train()` step body without the optimizer (reference trainutils.py:273-286).
    Returns (loss, logits, grads-by-name)."""
    leaves = {n: t.detach().clone().requires_grad_(True) for n, t in P.items()}
    logits = qa_tiger_forward(leaves, inputs, K, k, train=train, taps=taps, head_gate=head_gate)
    loss = cross_entropy(logits, inputs['label'])
    loss.backward()
    grads = {n: (t.grad if t.grad is not None else torch.zeros_like(t)) for n, t in leaves.items()}
    return loss.detach(), logits.detach(), grads


def min_router_gap(router_logits: Tensor) -> float:
    """Smallest gap between sorted router logits of any sample: the margin that
    protects the top-k ORDER."""
    s = torch.sort(router_logits, dim=-1, descending=True)[0]
    return float((s[:, :-1] - s[:, 1:]).min())
