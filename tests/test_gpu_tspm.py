"""TSPM (the reference's second model) on the HIP kernels, through the nn.Module surface, against the
goldens produced by the reference's own class and against the CPU oracle.  Gate as for the main path: eval logits within
1e-3 of max|logit| with identical arg-max (asserted 1e-4), the SAME top-K time indices, gradients per parameter in L2."""
import importlib

import numpy as np
import pytest
import torch

from conftest import load_golden, rel_err

pytestmark = pytest.mark.gpu
DEV = 'cuda'


def l2rel(a, b):
    a = np.asarray(a, np.float64).reshape(-1); b = np.asarray(b, np.float64).reshape(-1)
    return float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-30))


def build(qt, synth, cfg, seed):
    m = qt.TSPM(**cfg.model_kwargs())
    m.load_state_dict(synth.make_tspm_state_dict(cfg, seed), strict=True)          # reference parameter names
    return m.to(DEV).eval()


@pytest.mark.parametrize('name,kw', [('tspm_toy', None), ('tspm_ref_b2', {'B': 2})])
def test_tspm_logits_and_selection_match_reference(qt, synth, name, kw):
    g = load_golden(name)
    cfg = synth.TSPM_TOY if kw is None else synth.TspmConfig(**{**synth.TSPM_REF.asdict(), **kw})
    model = build(qt, synth, cfg, int(g['seed']))
    inp = {k: v.to(DEV) for k, v in synth.make_tspm_inputs(cfg, int(g['seed'])).items()}
    with torch.no_grad():
        out = model(inp)['out'].cpu().numpy()
    assert np.array_equal(model.last['top_k_index_sort'].cpu().numpy(), g['top_k_index_sort'])
    assert rel_err(out, g['logits']) < 1e-4
    assert np.array_equal(out.argmax(-1), g['logits'].argmax(-1))
    for k in ('audio_tssm', 'visual_sp', 'av_feat'):
        t = model.last[k].cpu().numpy()
        want = g['tap/' + k]
        got = t if want.shape == t.shape else t.reshape(-1)[:: t.size // 4096][:4096]
        assert rel_err(got, want) < 1e-4, k


def test_tspm_gradients_match_oracle(qt, synth):
    tspm = importlib.import_module('oracle.tspm_oracle')
    cfg = synth.TSPM_TOY
    P = synth.make_tspm_state_dict(cfg, 21)
    inp = synth.make_tspm_inputs(cfg, 21)
    loss_r, logits_r, grads_r = tspm.forward_backward(P, inp, cfg.topK)
    model = qt.TSPM(**cfg.model_kwargs()); model.load_state_dict(P, strict=True); model = model.to(DEV).eval()
    dinp = {k: v.to(DEV) for k, v in inp.items()}
    out = model(dinp)['out']
    loss = qt.CrossEntropyLoss()(out, dinp['label'])
    loss.backward()
    assert abs(float(loss.detach()) - float(loss_r)) < 1e-5
    assert rel_err(out.detach().cpu().numpy(), logits_r.numpy()) < 1e-4
    errs = {}
    for n, p in model.named_parameters():
        ref = grads_r[n].numpy()
        got = np.zeros_like(ref) if p.grad is None else p.grad.cpu().numpy()
        if np.abs(ref).max() == 0:
            assert np.abs(got).max() == 0, n                        # unused parameters (TemporalPerception's block, AV_Attn.norm1/2)
            continue
        errs[n] = l2rel(got, ref)
    worst = max(errs, key=errs.get)
    assert errs[worst] < 5e-3, (worst, errs[worst])
    assert float(np.median(list(errs.values()))) < 2e-4


def test_tspm_train_mode_runs_and_learns(qt, synth):
    cfg = synth.TspmConfig(**{**synth.TSPM_TOY.asdict(), 'B': 8})
    torch.manual_seed(0)
    model = qt.TSPM(**cfg.model_kwargs()).to(DEV).train()
    eng = qt.DataParallelStep(model, lr=1e-3)
    inp = {k: v.to(DEV) for k, v in synth.make_tspm_inputs(cfg, 3).items()}
    losses = [float(eng.step(inp)) for _ in range(12)]
    assert all(np.isfinite(losses)) and losses[-1] < losses[0]


def test_bmm_attention_matches_reference_math(qt):
    ops = qt.ops
    g = torch.Generator().manual_seed(2)
    for (n, L, E, shift) in [(6, 60, 512, 3), (5, 14, 512, 0), (4, 7, 64, 2)]:
        qkv = torch.randn(n * L, 3 * E, generator=g, requires_grad=True)
        x = qkv.detach().to(DEV).requires_grad_(True)
        o = ops.bmm_attention(x, None, n, L, L, E, 0, E, 2 * E, kv_shift=shift)
        go = torch.randn(n * L, E, generator=g)
        o.backward(go.to(DEV))
        q, k, v = [t.reshape(n, L, E) for t in qkv.split(E, dim=-1)]
        k, v = torch.roll(k, -shift, 0), torch.roll(v, -shift, 0)    # item i reads kv item (i + shift) % n
        ref = (torch.softmax(q @ k.transpose(1, 2) / E ** 0.5, -1) @ v).reshape(n * L, E)
        ref.backward(go)
        assert rel_err(o.detach().cpu().numpy(), ref.detach().numpy()) < 2e-5
        assert rel_err(x.grad.cpu().numpy(), qkv.grad.numpy()) < 2e-4


def test_tspm_samples_are_independent(qt, synth):
    """Unlike QA_TIGER (index quirk Q1), TSPM has no cross-sample coupling: a sample's logits do not depend on its batch."""
    cfg = synth.TspmConfig(**{**synth.TSPM_TOY.asdict(), 'B': 4})
    model = build(qt, synth, cfg, 9)
    inp = {k: v.to(DEV) for k, v in synth.make_tspm_inputs(cfg, 9).items()}
    with torch.no_grad():
        full = model(inp)['out'].clone()
        for b in (0, 3):
            one = {k: v[b:b + 1] for k, v in inp.items()}
            single = model(one)['out']
            assert float((single[0] - full[b]).abs().max() / full.abs().max()) < 1e-5
