"""End-to-end parity of the HIP path (through the nn.Module surface and the
C ABI) against (a) golden vectors produced by the reference itself and (b) the
CPU oracle on the same seeded inputs.  GPU box only.

Gate (BASELINE.json north_star / SURVEY.md §8d): max|Δlogit| / max|logit| <= 1e-3
and identical arg-max answers, eval mode, identical per-device batch composition.
The fp32 kernels are expected to land near 1e-6; the asserted bound is 1e-4 so a
precision regression is caught long before the gate.

Gradients: the network is piecewise linear in places (ReLU gates, clamp gates, the
arg-max frame of the Gaussian peak normalisation), so gradients are DISCONTINUOUS in
the activations.  tools/grad_sensitivity.py shows that in the fp64 oracle itself a
1e-6 relative input perturbation flips one expert ReLU unit and moves that expert's
weight gradient by 2.2e-3 (L2) while the logits move by 1e-6.  Gradient parity is
therefore asserted per parameter as an L2 relative error < GRAD_L2 (5e-3), plus a
median over parameters at rounding level."""
import numpy as np
import pytest
import torch

from conftest import load_golden, golden_cfg, rel_err, GOLDEN_CONFIGS

pytestmark = pytest.mark.gpu
DEV = 'cuda'
GATE = 1e-3
EXPECT = 1e-4
GRAD_L2 = 5e-3
GRAD_MEDIAN = 5e-5


def l2rel(a, b):
    a = np.asarray(a, dtype=np.float64).reshape(-1)
    b = np.asarray(b, dtype=np.float64).reshape(-1)
    return float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-30))


def build(qt, synth, cfg, seed):
    model = qt.QA_TIGER(**cfg.model_kwargs())
    model.load_state_dict(synth.make_state_dict(cfg, seed), strict=True)
    return model.to(DEV).eval()


def to_dev(inp):
    return {k: v.to(DEV) for k, v in inp.items()}


def run(qt, synth, name):
    cfg, seed = golden_cfg(synth, name)
    model = build(qt, synth, cfg, seed)
    inp = to_dev(synth.make_inputs(cfg, seed))
    out = model(inp)['out']
    loss = qt.CrossEntropyLoss()(out, inp['label'])
    loss.backward()
    return cfg, model, inp, out, loss


@pytest.mark.parametrize('name', list(GOLDEN_CONFIGS))
def test_logits_argmax_loss_vs_reference_golden(qt, synth, name):
    g = load_golden(name)
    cfg, model, inp, out, loss = run(qt, synth, name)
    e = rel_err(out.detach().cpu().numpy(), g['logits'])
    assert e < EXPECT < GATE, e
    assert (out.argmax(-1).cpu().numpy() == g['logits'].argmax(-1)).all()
    assert float(loss) == pytest.approx(float(g['loss']), rel=1e-5)
    for agg in ('at_aggregator', 'vt_aggregator'):
        last = getattr(model, agg).last
        if f'tap/{agg}.topk_inds' in g:
            assert (last['topk_inds'].cpu().numpy() == g[f'tap/{agg}.topk_inds']).all(), \
                f'top-k order differs (min router gap {float(g[f"router_gap/{agg}"]):.2e})'
            assert rel_err(last['gauss'].cpu().numpy(), g[f'tap/{agg}.gauss']) < 1e-4
            assert rel_err(last['topk_probs'].cpu().numpy(), g[f'tap/{agg}.topk_probs']) < 1e-4
            assert rel_err(last['temp_w'].cpu().numpy(), g[f'tap/{agg}.temp_w']) < 1e-4


@pytest.mark.parametrize('name', list(GOLDEN_CONFIGS))
def test_gradients_vs_reference_golden(qt, synth, name):
    g = load_golden(name)
    cfg, model, inp, out, loss = run(qt, synth, name)
    errs = []
    for pname, p in model.named_parameters():
        assert p.grad is not None, f'{pname} got no gradient (DDP find_unused_parameters=False contract)'
        gr = p.grad.detach().cpu().numpy()
        if f'grad/{pname}' in g:
            e = l2rel(gr, g[f'grad/{pname}'])
        else:
            assert float(np.sqrt((gr.astype(np.float64) ** 2).sum())) == pytest.approx(float(g[f'gradnorm/{pname}']), rel=GRAD_L2, abs=1e-9), pname
            flat = gr.reshape(-1)
            e = l2rel(flat[:: max(1, flat.size // 256)][:256], g[f'gradslice/{pname}'])
        assert e < GRAD_L2, (pname, e)
        errs.append(e)
    assert len(errs) == len(synth.param_shapes(cfg))
    assert float(np.median(errs)) < GRAD_MEDIAN, float(np.median(errs))


def test_module_outputs_vs_reference_golden(qt, synth):
    g = load_golden('toy')
    cfg, seed = golden_cfg(synth, 'toy')
    model = build(qt, synth, cfg, seed)
    inp = to_dev(synth.make_inputs(cfg, seed))
    with torch.no_grad():
        audio = model.audio_proj(inp['audio']); video = model.video_proj(inp['video'])
        words = model.words_proj(inp['words_feat']); quest = model.quest_proj(inp['quest_feat'])
        patch = model.patch_proj(inp['patch'])
        audio, video = model.crs_attn(audio, video, words)
        assert rel_err(audio.cpu().numpy(), g['tap/crs_attn.audio']) < EXPECT
        assert rel_err(video.cpu().numpy(), g['tap/crs_attn.video']) < EXPECT
        pa, pv = model.patch_selecter(patch, audio, video)
        assert rel_err(pa.cpu().numpy(), g['tap/patch_selecter.a']) < EXPECT
        assert rel_err(pv.cpu().numpy(), g['tap/patch_selecter.v']) < EXPECT
        a_g = model.at_aggregator(quest, audio)
        assert rel_err(a_g.cpu().numpy(), g['tap/at_aggregator.out']) < EXPECT
        ap, vp = model.vt_aggregator(quest, video, [pa, pv])
        assert rel_err(ap.cpu().numpy(), g['tap/vt_aggregator.a_out']) < EXPECT
        assert rel_err(vp.cpu().numpy(), g['tap/vt_aggregator.v_out']) < EXPECT
        f1 = model.quest_grounding(quest, [ap, vp])
        assert rel_err(f1.cpu().numpy(), g['tap/quest_grounding.1']) < EXPECT
        f2 = model.quest_grounding(quest, [f1.unsqueeze(1), a_g])
        assert rel_err(f2.cpu().numpy(), g['tap/quest_grounding.2']) < EXPECT


def _oracle_vs_hip(qt, synth, oracle, cfg, seed, with_grads):
    P = synth.make_state_dict(cfg, seed)
    inp = synth.make_inputs(cfg, seed)
    model = qt.QA_TIGER(**cfg.model_kwargs())
    model.load_state_dict(P, strict=True)
    model = model.to(DEV).eval()
    dinp = to_dev(inp)
    taps = {}
    if with_grads:
        out = model(dinp)['out']
        qt.CrossEntropyLoss()(out, dinp['label']).backward()
        loss_r, logits_r, grads_r = oracle.forward_backward(P, inp, cfg.K, cfg.k, taps=taps)
    else:
        with torch.no_grad():
            out = model(dinp)['out']
            logits_r = oracle.qa_tiger_forward(P, inp, cfg.K, cfg.k, taps=taps)
        grads_r = None
    gaps = [oracle.min_router_gap(taps[f'{a}.router_logits']) for a in ('at_aggregator', 'vt_aggregator')]
    e = rel_err(out.detach().cpu().numpy(), logits_r.numpy())
    same = bool((out.argmax(-1).cpu() == logits_r.argmax(-1)).all())
    for a in ('at_aggregator', 'vt_aggregator'):
        assert torch.equal(getattr(model, a).last['topk_inds'].cpu(), taps[f'{a}.topk_inds']), f'top-k order flip, gaps {gaps}'
    assert e < EXPECT < GATE, (e, gaps)
    assert same, gaps
    if grads_r is not None:
        errs = []
        for n, p in model.named_parameters():
            errs.append(l2rel(p.grad.cpu().numpy(), grads_r[n].numpy()))
            assert errs[-1] < GRAD_L2, (n, errs[-1])
        assert float(np.median(errs)) < GRAD_MEDIAN, float(np.median(errs))
    return e, gaps


def test_config2_b32_fwd_bwd_vs_oracle(qt, synth, oracle):
    """ViT-L/14 shape (T=60, D=768, K=8) at B=32: logits, arg-max, top-k order and every gradient."""
    cfg = synth.PathConfig(**{**synth.CONFIG_2.asdict(), 'B': 32})
    _oracle_vs_hip(qt, synth, oracle, cfg, 713, with_grads=True)


def test_config2_full_b256_gradients_vs_oracle(qt, synth, oracle):
    """EVERY gradient at the size the metric is quoted on (B=256, T=60, D=768, K=8): forward + CE + backward on the HIP
    path against the CPU oracle's autograd on the same batch (fwd+bwd of the oracle: tens of seconds on the box's cores).
    This is the bound on the split-precision BACKWARD GEMMs at the headline size.  ONE evaluation, strict bounds.

    This batch holds a gate that sits at its threshold: one unit of the ReLU in front of the head (net.py:141) has a pre-activation of
    1.9e-6 (196,608 units of O(1) LayerNorm outputs: a few within 1e-5 of zero in every batch).  Logits and loss are continuous
    there, the gradient is not: with the last bits of the forward (the kernels' summation order differs from the oracle's) the unit is on or off,
    and every gradient below the head then moves by 1.4e-3 ... 9.6e-3 (`tools/dbg_tie.py`).  Instead of evaluating until the kernels
    land on the oracle's side, the oracle is evaluated WITH THE KERNELS' GATE STATE (`head_gate`) — after asserting that it differs
    from the oracle's own ReLU only at units whose pre-activation is below 2e-5 in magnitude, so a wrong gate anywhere else fails."""
    cfg = synth.CONFIG_2
    P = synth.make_state_dict(cfg, 713)
    inp = synth.make_inputs(cfg, 713)
    model = qt.QA_TIGER(**cfg.model_kwargs())
    model.load_state_dict(P, strict=True)
    model = model.to(DEV).eval()
    dinp = to_dev(inp)
    grounded = []
    hook = model.quest_grounding.register_forward_hook(lambda m, a, out: grounded.append(out.detach()))
    out = model(dinp)['out']
    hook.remove()
    loss = qt.CrossEntropyLoss()(out, dinp['label'])
    loss.backward()
    gate = (grounded[1] > 0).cpu()                                   # second call: LayerNorm + ReLU fused (relu_out=True)
    taps = {}
    loss_r, logits_r, grads_r = oracle.forward_backward(P, inp, cfg.K, cfg.k, taps=taps, head_gate=gate)
    pre = taps['quest_grounding.2']
    flipped = gate != (pre > 0)
    print(f'head gate: {int(flipped.sum())} unit(s) on the other side of the oracle\'s ReLU, |pre-activation| there '
          f'{[float(v) for v in pre[flipped].abs()]}; smallest |pre-activation| of the batch {float(pre.abs().min()):.2e}')
    assert int(flipped.sum()) <= 4 and (int(flipped.sum()) == 0 or float(pre[flipped].abs().max()) < 2e-5)
    for a in ('at_aggregator', 'vt_aggregator'):
        assert torch.equal(getattr(model, a).last['topk_inds'].cpu(), taps[f'{a}.topk_inds'])
    assert rel_err(out.detach().cpu().numpy(), logits_r.numpy()) < EXPECT
    assert float(loss) == pytest.approx(float(loss_r), rel=1e-5)
    errs = {n: l2rel(p.grad.cpu().numpy(), grads_r[n].numpy()) for n, p in model.named_parameters()}
    worst = sorted(errs.items(), key=lambda kv: -kv[1])[:5]
    print(f'B=256 gradient L2 rel err: median {np.median(list(errs.values())):.2e}, worst {worst}')
    assert float(np.median(list(errs.values()))) < GRAD_MEDIAN
    for n, e in errs.items():
        assert e < GRAD_L2, (n, e)


def test_config2_full_b256_forward_vs_oracle(qt, synth, oracle):
    """BASELINE config 2 at its full size: B=256, T=60, D=768, K=8 (CPU oracle forward ~10-20 s)."""
    e, gaps = _oracle_vs_hip(qt, synth, oracle, synth.CONFIG_2, 713, with_grads=False)
    print(f'B=256 rel logit err {e:.2e}, min router gaps {gaps}')


def test_config4_long_video_b8_vs_oracle(qt, synth, oracle):
    """BASELINE config 4 shape (T=300, D=1024, K=16; streaming attention, d_h=128) at B=8: logits, arg-max, top-k order."""
    cfg = synth.PathConfig(**{**synth.CONFIG_4.asdict(), 'B': 8})
    _oracle_vs_hip(qt, synth, oracle, cfg, 713, with_grads=False)


def test_config4_full_size_b128_invariances(qt, synth):
    """Config 4 at its full size (B=128, T=300, D=1024, K=16): finite, deterministic to rounding, and
    invariant to patch-token / word permutations (no positional terms, no masks)."""
    cfg = synth.CONFIG_4
    model = build(qt, synth, cfg, 713)
    inp = qt.synth.fast_device_inputs(cfg, DEV, seed=9)
    with torch.no_grad():
        base = model(inp)['out']
        pp = torch.randperm(cfg.P, device=DEV)
        wp = torch.randperm(cfg.L, device=DEV)
        out_p = model({**inp, 'patch': inp['patch'][:, :, pp], 'words_feat': inp['words_feat'][:, wp]})['out']
    scale = float(base.abs().max())
    assert torch.isfinite(base).all() and base.shape == (cfg.B, 42)
    assert float((out_p - base).abs().max()) / scale < 1e-4


def test_config4_full_size_b128_backward(qt, synth):
    """Config 4 at its full size, BACKWARD: 537600 patch rows, gradient images of up to 6.6 GB (537600 x 3072 fp32) — beyond the
    32-bit per-lane offsets of the LDS-DMA loads unless the operand base moves with the tile (regression: the dX GEMM of the packed
    in-projection was refused at this size and the step died).  No CPU oracle at this size (T = 300): the pre-split pipeline's
    gradients are compared with the split-in-the-consumer kernels' (`ops.plane = False`: other kernels, other addressing, same
    math), every parameter, at the gates of the other gradient tests."""
    cfg = synth.CONFIG_4
    model = build(qt, synth, cfg, 713)
    inp = qt.synth.fast_device_inputs(cfg, DEV, seed=9)

    def grads(plane):
        keep = qt.ops.plane
        qt.ops.plane = plane
        try:
            model.zero_grad(set_to_none=True)
            qt.CrossEntropyLoss()(model(inp)['out'], inp['label']).backward()
        finally:
            qt.ops.plane = keep
        return {n: p.grad.clone() for n, p in model.named_parameters()}
    ga, gb = grads(True), grads(False)
    errs = {n: float((ga[n] - gb[n]).norm() / gb[n].norm().clamp_min(1e-30)) for n in ga}
    assert all(bool(torch.isfinite(g).all()) for g in ga.values())
    print(f'config 4 (B=128) gradients, pre-split vs split-in-the-consumer: median {np.median(list(errs.values())):.2e}, worst {max(errs.values()):.2e}')
    for n, e in errs.items():
        assert e < GRAD_L2, (n, e)
    assert float(np.median(list(errs.values()))) < GRAD_MEDIAN
    del ga, gb
    torch.cuda.empty_cache()


def test_full_size_invariances(qt, synth):
    """Size-independent properties at B=256: the path has no positional term on patch
    tokens or words and no masks, so permuting them must not
    change the logits; and eval mode is deterministic."""
    cfg = synth.CONFIG_2
    model = build(qt, synth, cfg, 713)
    inp = qt.synth.fast_device_inputs(cfg, DEV, seed=5)
    with torch.no_grad():
        base = model(inp)['out']
        again = model(inp)['out']
        # no GEMM of the forward meets its K splits in atomics any more (workspace / in-workgroup split-K): the same bits every run.
        # (The diagnostic modes QT_GEMM=fp32 / bf16x1 send the head-batched products to kernels that still split K with atomics.)
        import os
        if qt.ops.precision['fwd'] == 'f16x3' and not os.environ.get('QT_PS_ATOMIC'):     # (QT_PS_ATOMIC=1: round 2's atomics, for A/B runs)
            assert torch.equal(base, again)
        else:
            assert float((base - again).abs().max()) / float(base.abs().max()) < 1e-5
        pp = torch.randperm(cfg.P, device=DEV)
        out_p = model({**inp, 'patch': inp['patch'][:, :, pp]})['out']
        wp = torch.randperm(cfg.L, device=DEV)
        out_w = model({**inp, 'words_feat': inp['words_feat'][:, wp]})['out']
    scale = float(base.abs().max())
    assert torch.isfinite(base).all()
    assert float((out_p - base).abs().max()) / scale < 1e-4
    assert float((out_w - base).abs().max()) / scale < 1e-4


def test_batch_composition_dependence_matches_reference(qt, synth):
    """Quirk Q1 must be reproduced: the same samples in a smaller batch give different logits."""
    cfg, seed = golden_cfg(synth, 'toy')
    model = build(qt, synth, cfg, seed)
    inp = to_dev(synth.make_inputs(cfg, seed))
    with torch.no_grad():
        full = model(inp)['out']
        part = model({k: v[:3] for k, v in inp.items()})['out']
    assert float((full[:3] - part).abs().max()) > 1e-3


def test_train_step_matches_oracle_adam_in_eval_mode(qt, synth, oracle):
    """zero_grad -> forward -> CE -> backward -> Adam (reference trainutils.py:273-287), dropout off."""
    cfg, seed = golden_cfg(synth, 'toy_kk')
    P = synth.make_state_dict(cfg, seed)
    inp = synth.make_inputs(cfg, seed)
    model = build(qt, synth, cfg, seed)            # eval(): dropout off, same arithmetic as the oracle
    eng = qt.DataParallelStep(model, lr=1e-4, betas=(0.95, 0.999))
    ref = {n: t.clone() for n, t in P.items()}
    m = {n: torch.zeros_like(t) for n, t in P.items()}
    v = {n: torch.zeros_like(t) for n, t in P.items()}
    for step in range(1, 4):
        loss = eng.step(to_dev(inp))
        loss_r, _, grads = oracle.forward_backward(ref, inp, cfg.K, cfg.k)
        oracle.adam_step(ref, grads, m, v, step)
        assert float(loss) == pytest.approx(float(loss_r), rel=2e-5)
        if step == 1:
            solid = {n: gr.abs() > 1e-3 * gr.abs().max() for n, gr in grads.items()}
    for n, p in model.named_parameters():
        # after 3 steps each weight moved by ~3e-4; compare the UPDATE, not the weight.  Adam turns any
        # gradient >> eps into a full +-lr step, so elements whose true gradient is zero (e.g. the key
        # bias of an MHA: softmax is shift-invariant) carry pure rounding noise and are excluded.
        upd, upd_r = p.detach().cpu() - P[n], ref[n] - P[n]
        d = (upd - upd_r).abs()[solid[n]]
        assert float(d.max()) < 2e-5, n
        assert float(d.median()) < 1e-6, n


def test_direct_gradients_equal_autograd_gradients(qt, synth):
    """Under `DataParallelStep` every parameter's `.grad` is its slice of the zeroed flat gradient buffer for the whole step:
    producers that claim it (dW GEMMs, bias column sums, LayerNorm, the row blocks of packed in-projections, the stacked expert
    parameters) accumulate in place and hand autograd None; everything else is added in place by AccumulateGrad.  The result
    must be what plain autograd computes on an identical model without any of that (same kernels, p.grad from autograd) — to
    the last bits of the fp32 atomics — at a shape where every Linear class occurs (pre-split pipeline for the patch rows, small
    tiles for the rest), twice in a row (the second step must not see the first one's claims)."""
    cfg = synth.PathConfig(**{**synth.CONFIG_2.asdict(), 'B': 16})
    P = synth.make_state_dict(cfg, 9)
    inp = to_dev(synth.make_inputs(cfg, 9))

    def model():
        m = qt.QA_TIGER(**cfg.model_kwargs())
        m.load_state_dict(P, strict=True)
        return m.to(DEV).eval()
    plain = model()
    qt.CrossEntropyLoss()(plain(inp)['out'], inp['label']).backward()
    ref = {n: p.grad.clone() for n, p in plain.named_parameters()}
    eng = qt.DataParallelStep(model(), lr=1e-4)
    names = [n for n, _ in eng.model.named_parameters()]
    # Tolerances: a mistake in the protocol (a gradient added twice, a missed or overwritten slice) shows as an O(1) error.
    # Two CORRECT evaluations differ in the last bits of the forward (split-K atomics), and a ReLU pre-activation that sits at
    # ~0 for this seed then flips: every gradient upstream of that unit moves by ~1e-3 (the discontinuity DESIGN.md section 4
    # describes; observed here as the same two or three discrete states from pass to pass, with and without this engine).
    for rep in range(2):
        eng.forward_backward(inp)
        errs = []
        for n, p in eng.model.named_parameters():
            assert p.grad is not None and p.grad.data_ptr() == p._qt_gv.data_ptr(), n       # .grad IS the flat slice
            errs.append(float((p.grad - ref[n]).norm() / ref[n].norm().clamp_min(1e-30)))
            assert errs[-1] < GRAD_L2, (rep, n, errs[-1])
        assert float(np.median(errs)) < 1e-5, float(np.median(errs))
        print(f'direct vs autograd gradients, pass {rep}: median {np.median(errs):.1e}, worst {max(errs):.1e} over {len(names)} parameters')
    assert not qt.ops.direct_grads and qt.ops.grad_ready_cb is None                          # the mode ends with finish()


def test_train_mode_dropout_and_loss_decreases(qt, synth):
    cfg = synth.CONFIG_TOY_KK
    torch.manual_seed(0)
    qt.ops.seeds.reset()
    model = qt.QA_TIGER(**cfg.model_kwargs()).to(DEV).train()
    inp = to_dev(synth.make_inputs(cfg, 3))
    with torch.no_grad():
        a, b = model(inp)['out'], model(inp)['out']
    assert not torch.equal(a, b)                   # dropout active, fresh masks per call
    eng = qt.DataParallelStep(model, lr=1e-3)
    losses = [float(eng.step(inp)) for _ in range(30)]
    assert all(np.isfinite(losses))
    assert np.mean(losses[-5:]) < np.mean(losses[:5]) - 0.3


def test_training_soak_at_real_dims(qt, synth):
    """40 train steps at the ViT-L/14 shape (T=60, D=768, K=8, B=16), dropout on, split-precision GEMMs:
    loss and every parameter / moment stay finite and the loss on the fixed batch goes down."""
    cfg = synth.PathConfig(**{**synth.CONFIG_2.asdict(), 'B': 16})
    torch.manual_seed(1)
    qt.ops.seeds.reset()
    model = qt.QA_TIGER(**cfg.model_kwargs()).to(DEV).train()
    eng = qt.DataParallelStep(model, lr=1e-4, betas=(0.95, 0.999))
    batch = qt.synth.fast_device_inputs(cfg, DEV, seed=3)
    losses = [float(eng.step(batch)) for _ in range(40)]
    assert all(np.isfinite(losses)), losses
    assert np.mean(losses[-5:]) < np.mean(losses[:5]) - 0.2, (losses[:5], losses[-5:])
    for t in (eng.flat.flat, eng.flat.grad, eng.optim.exp_avg, eng.optim.exp_avg_sq):
        assert bool(torch.isfinite(t).all())
    assert eng.optim.step_count == 40


def test_repeated_backward_with_unchanged_weights_is_stable(qt, synth):
    """Weights' max|w| records and split images are cached while the weights do not change (evaluation, gradient accumulation,
    a stock optimizer stepping every n-th batch).  The records of one backward pass come from a recycled pool; a cached record
    must not live there (a wrapped cursor would hand the cached slot to another tensor and the weight would be scaled by that
    tensor's maximum).  Twelve fwd+bwd passes cycle a small pool several times: the last gradients must equal the first."""
    cfg = synth.PathConfig(**{**synth.CONFIG_2.asdict(), 'B': 16})
    model = qt.QA_TIGER(**cfg.model_kwargs())
    model.load_state_dict(synth.make_state_dict(cfg, 5), strict=True)
    model = model.to(DEV).eval()
    inp = to_dev(synth.make_inputs(cfg, 5))
    keep_half = qt.ops._AmaxPool.HALF
    qt.ops._AmaxPool.HALF = 64                                   # cycle quickly
    qt.ops._amax_pool.buf = None
    try:
        first = None
        for it in range(12):
            model.zero_grad(set_to_none=True)
            qt.CrossEntropyLoss()(model(inp)['out'], inp['label']).backward()
            g = torch.cat([p.grad.flatten() for p in model.parameters()])
            if first is None:
                first = g.clone()
        d = float((g - first).norm() / first.norm())
        print(f'gradient drift over 12 passes with cached weight images: {d:.2e}')
        assert d < 5e-4          # (a weight scaled by another tensor's record is off by orders of magnitude; two correct passes can differ by a flipped ReLU gate: ~3e-5 of the whole vector)
    finally:
        qt.ops._AmaxPool.HALF = keep_half
        qt.ops._amax_pool.buf = None


def test_range_guard_large_activations_and_tiny_weights(qt, synth, oracle):
    """fp16-split GEMMs outside fp16's comfortable range: features of magnitude ~1e5 and projection weights of
    ~1e-4.  Weight operands are always scaled by max|w| (free: one launch per optimizer step), so tiny weights cost nothing.
    Activations beyond fp16's range need `ops.robust_range` (max|x| of every Linear input: one extra read each): with it the
    logits stay within 1e-4 of the oracle; WITHOUT it the overflow shows as non-finite logits — never as plausible wrong numbers."""
    cfg = synth.PathConfig(**{**synth.CONFIG_2.asdict(), 'B': 8})
    P = synth.make_state_dict(cfg, 713)
    inp = synth.make_inputs(cfg, 713)
    for n in ('audio_proj.proj.weight', 'video_proj.proj.weight', 'patch_proj.proj.weight'):
        P[n] = P[n] * 1e-5                                  # ~1e-6 weights: features x weights keeps the projected activations O(1),
                                                            # so the rest of the network stays as well conditioned as in the other tests
    big = {k: (v * 1e5 if k in ('audio', 'video', 'patch') else v) for k, v in inp.items()}
    model = qt.QA_TIGER(**cfg.model_kwargs())
    model.load_state_dict(P, strict=True)
    model = model.to(DEV).eval()
    with torch.no_grad():
        ref = oracle.qa_tiger_forward(P, big, cfg.K, cfg.k).numpy()
        keep = qt.ops.robust_range
        try:
            qt.ops.robust_range = True
            out = model(to_dev(big))['out'].cpu().numpy()
            qt.ops.robust_range = False
            raw = model(to_dev(big))['out'].cpu().numpy()
        finally:
            qt.ops.robust_range = keep
    assert np.isfinite(ref).all()
    e = rel_err(out, ref)
    print(f'range guard: robust mode rel logit err {e:.2e}; default mode finite: {bool(np.isfinite(raw).all())}')
    assert e < 1e-4 and (out.argmax(-1) == ref.argmax(-1)).all()
    assert not np.isfinite(raw).all() or rel_err(raw, ref) < 1e-4       # loud (non-finite) or right — never silently wrong
    # the BACKWARD in the robust mode: the dW GEMMs read the inputs' saved images / rows scaled by the same max|x| records
    # (regression: the pre-split path saved the scaled image but told the dW GEMM it was unscaled: gradients off by 2^11)
    keep = qt.ops.robust_range
    try:
        qt.ops.robust_range = True
        dbig = to_dev(big)
        qt.CrossEntropyLoss()(model(dbig)['out'], dbig['label']).backward()
    finally:
        qt.ops.robust_range = keep
    loss_r, _, grads_r = oracle.forward_backward(P, big, cfg.K, cfg.k)
    errs = {n: l2rel(p.grad.cpu().numpy(), grads_r[n].numpy()) for n, p in model.named_parameters()}
    print(f'range guard: robust mode gradient L2 rel err: median {np.median(list(errs.values())):.2e}, worst {max(errs.values()):.2e}')
    for n, e2 in errs.items():
        assert e2 < GRAD_L2, (n, e2)
    assert float(np.median(list(errs.values()))) < GRAD_MEDIAN


def test_cpu_tensors_fail_loudly(qt, synth):
    cfg = synth.CONFIG_TOY
    model = qt.QA_TIGER(**cfg.model_kwargs())
    with pytest.raises(qt.QTError, match='no CPU fallback'):
        model(synth.make_inputs(cfg))


def test_reduced_precision_mode_argmax_agreement(qt, synth):
    """BASELINE config 5 direction: GEMMs on ONE bf16 MFMA per product (ops.precision 'bf16x1').  It is outside
    the 1e-3 gate by construction; the check is the one config 5 names — arg-max agreement with the fp32 path —
    plus a bound on the logit drift so that a broken kernel cannot hide behind 'reduced precision'."""
    name = 'config2_b8'
    g = load_golden(name)
    cfg, seed = golden_cfg(synth, name)
    model = build(qt, synth, cfg, seed)
    inp = {k: v.to(DEV) for k, v in synth.make_inputs(cfg, seed).items()}
    keep = dict(qt.ops.precision)
    qt.ops.precision.update(fwd='bf16x1', bwd='bf16x1')
    try:
        with torch.no_grad():
            out = model(inp)['out'].cpu().numpy()
    finally:
        qt.ops.precision.update(keep)
    err = rel_err(out, g['logits'])
    agree = float((out.argmax(-1) == g['logits'].argmax(-1)).mean())
    print(f'bf16x1: rel logit err {err:.2e}, arg-max agreement {agree:.3f}')
    assert 1e-5 < err < 5e-2          # really reduced precision, and not garbage
    assert agree >= 0.75


def _topk_order_agreement(model, ref_inds):
    """fraction of samples whose top-k expert ORDER equals the reference's, per aggregator"""
    return {a: float((getattr(model, a).last['topk_inds'].cpu() == ref_inds[a]).all(dim=1).float().mean())
            for a in ('at_aggregator', 'vt_aggregator')}


def test_config5_fp8_projection_gemms_agreement(qt, synth, oracle):
    """BASELINE configs[4]: projection GEMMs on fp8 MFMA (e4m3 images, fp32 accumulate; router / Gaussian / softmax / LayerNorm
    statistics in fp32), forward.  Outside the 1e-3 gate by construction; what config 5 asks for is agreement with the fp32 path:
    arg-max agreement RATE and top-k ORDER agreement, here against (a) the reference's own config2_b8
    golden and (b) the CPU oracle at the config's B = 512."""
    keep = dict(qt.ops.precision)
    report = {}
    try:
        # (a) reference golden, B = 8
        g = load_golden('config2_b8')
        cfg, seed = golden_cfg(synth, 'config2_b8')
        model = build(qt, synth, cfg, seed)
        inp = to_dev(synth.make_inputs(cfg, seed))
        qt.ops.precision.update(fwd='fp8', bwd='fp8')
        with torch.no_grad():
            out = model(inp)['out'].cpu().numpy()
        report['b8'] = dict(argmax=float((out.argmax(-1) == g['logits'].argmax(-1)).mean()), rel_logit_err=rel_err(out, g['logits']),
                            topk_order=_topk_order_agreement(model, {a: torch.from_numpy(g[f'tap/{a}.topk_inds']) for a in ('at_aggregator', 'vt_aggregator')}))
        assert np.isfinite(out).all() and 1e-4 < report['b8']['rel_logit_err'] < 0.5      # really fp8, and not garbage
        with pytest.raises(qt.QTError, match='forward-only'):
            model(inp)                                                                    # autograd on: refused loudly
        # (b) oracle at B = 512
        qt.ops.precision.update(keep)
        cfg = synth.PathConfig(**{**synth.CONFIG_2.asdict(), 'B': 512})
        P = synth.make_state_dict(cfg, 713)
        inp = synth.make_inputs(cfg, 713)
        model = qt.QA_TIGER(**cfg.model_kwargs())
        model.load_state_dict(P, strict=True)
        model = model.to(DEV).eval()
        taps = {}
        with torch.no_grad():
            ref = oracle.qa_tiger_forward(P, inp, cfg.K, cfg.k, taps=taps).numpy()
        qt.ops.precision.update(fwd='fp8', bwd='fp8')
        with torch.no_grad():
            out = model(to_dev(inp))['out'].cpu().numpy()
        report['b512'] = dict(argmax=float((out.argmax(-1) == ref.argmax(-1)).mean()), rel_logit_err=rel_err(out, ref),
                              topk_order=_topk_order_agreement(model, {a: taps[f'{a}.topk_inds'] for a in ('at_aggregator', 'vt_aggregator')}))
    finally:
        qt.ops.precision.update(keep)
    print('config5 (fp8 projection GEMMs) agreement with the fp32 path:', report)
    assert np.isfinite(out).all()
    assert report['b512']['argmax'] >= 0.5 and report['b8']['argmax'] >= 0.5
    assert report['b512']['rel_logit_err'] < 0.5


@pytest.mark.parametrize('shape', [(300, 512, 256, False, True), (512, 300, 1000, True, False), (256, 256, 4096, False, False)])
def test_bf16x1_gemm(qt, shape):
    M, N, K, ta, tb = shape
    g = torch.Generator().manual_seed(8)
    a = torch.randn((K, M) if ta else (M, K), generator=g)
    b = torch.randn((N, K) if tb else (K, N), generator=g)
    out = qt.ops.gemm(a.to(DEV), b.to(DEV), trans_a=ta, trans_b=tb, mode='bf16x1').cpu()
    # exact reference of the mode's definition: operands rounded to bf16, products and sums in high precision
    ar, br = a.bfloat16().double(), b.bfloat16().double()
    ref = (ar.t() if ta else ar) @ (br.t() if tb else br)
    assert rel_err(out.numpy(), ref.float().numpy()) < 2e-6
    full = (a.double().t() if ta else a.double()) @ (b.double().t() if tb else b.double())
    assert 1e-4 < rel_err(out.numpy(), full.float().numpy()) < 2e-2



def test_graphed_forward_in_a_fresh_process(qt):
    """HIP-graph capture of the eval forward (qt.GraphedForward) at the runtime's DEFAULT settings — round 1 needed
    DEBUG_CLR_GRAPH_PACKET_CAPTURE=0 because the library's own hipMemsetAsync nodes made replays fault; every zeroing is a kernel
    now.  The check runs tools/bench_graph.py in a child process WITHOUT that variable: captured logits equal the eager ones for
    the captured batch, for another batch, and after an eager forward plus runtime blits (memset, copy) between replays."""
    import os, subprocess, sys
    from conftest import REPO
    env = {k: v for k, v in os.environ.items() if k != 'DEBUG_CLR_GRAPH_PACKET_CAPTURE'}
    res = subprocess.run([sys.executable, os.path.join(REPO, 'tools', 'bench_graph.py'), '8', '32'], cwd=REPO, env=env,
                         capture_output=True, text=True, timeout=300)
    assert res.returncode == 0 and res.stdout.strip().endswith('OK'), res.stdout[-2000:] + res.stderr[-2000:]
    print(res.stdout)


def test_graphed_train_step_matches_eager_and_redraws_dropout(qt, synth):
    """qt.GraphedTrainStep: the whole train step (zero grads, forward, CE, backward, Adam) replayed from ONE HIP graph.
    (a) eval mode (no dropout): replays reproduce an eager engine's loss sequence and weights, i.e. the device-side Adam step count and
        bias corrections advance per replay; (b) train mode with lr = 0: two replays on the same batch give DIFFERENT losses — the
        dropout masks are salted from device memory, not frozen into the graph — and both are finite."""
    cfg = synth.PathConfig(**{**synth.CONFIG_2.asdict(), 'B': 8})
    P = synth.make_state_dict(cfg, 713)
    batch = to_dev(synth.make_inputs(cfg, 713))

    def engine(train, lr):
        m = qt.QA_TIGER(**cfg.model_kwargs())
        m.load_state_dict(P, strict=True)
        m = m.to(DEV)
        m.train() if train else m.eval()
        return qt.DataParallelStep(m, lr=lr, betas=(0.95, 0.999))

    try:
        eager, graphed = engine(False, 1e-4), engine(False, 1e-4)
        gs = qt.GraphedTrainStep(graphed, batch, warmup=2)          # 2 eager warm-up steps, then the capture (which records, not runs)
        for _ in range(2):
            eager.step(batch)
        for i in range(3):
            le, lg = float(eager.step(batch)), float(gs(batch))
            # two engines: split-K atomics order + Adam's +-lr steps on noise-level gradients make the trajectories drift apart step by step
            assert lg == pytest.approx(le, rel=3e-4 if i == 0 else 5e-3), (i, le, lg)
        assert graphed.optim.step_count == eager.optim.step_count == 5
        assert int(gs.step_dev.item()) == 5                          # the device-side Adam step count advanced once per replay
        d = (graphed.flat.flat - eager.flat.flat).abs()
        assert float(d.median()) < 2e-6 and float((d > 1e-4).float().mean()) < 0.02      # (Adam turns rounding-noise gradients into +-lr steps)
        del gs
        tr = engine(True, 0.0)
        gs = qt.GraphedTrainStep(tr, batch, warmup=1)
        losses = [float(gs(batch)) for _ in range(4)]
        assert all(np.isfinite(losses)) and len(set(round(l, 6) for l in losses)) >= 3, losses
    finally:
        qt._lib.load().qt_set_dropout_salt(None)                     # the salt pointer is library-global: do not leak it into other tests


def test_graphed_train_step_follows_scheduler_and_loaded_state(qt, synth):
    """The captured train step reads lr / betas / eps / weight_decay and the step count from DEVICE memory written before every
    replay (qt_adam_set_dev), so the reference's schedulers (StepLR, trainutils.py:147-182) and `optimizer.load_state_dict` act on
    replays exactly as on eager steps.  (a) lr set to 0 through the param group after the capture: replays stop moving the weights;
    (b) StepLR(step_size=2, gamma=0.1) over 4 steps: the graphed engine's weights follow an eager engine under the same scheduler and
    are FAR from an eager engine at the constant capture-time lr; (c) `load_state_dict` of an optimizer state with another step count:
    the device-side count follows."""
    cfg = synth.PathConfig(**{**synth.CONFIG_2.asdict(), 'B': 8})
    P = synth.make_state_dict(cfg, 713)
    batch = to_dev(synth.make_inputs(cfg, 713))

    def engine(lr):
        m = qt.QA_TIGER(**cfg.model_kwargs())
        m.load_state_dict(P, strict=True)
        return qt.DataParallelStep(m.to(DEV).eval(), lr=lr, betas=(0.95, 0.999))

    try:
        g = engine(1e-3)
        gs = qt.GraphedTrainStep(g, batch, warmup=1)
        gs(batch)
        w0 = g.flat.flat.clone()
        g.optim.param_groups[0]['lr'] = 0.0                          # (a)
        gs(batch)
        assert torch.equal(g.flat.flat, w0), 'a replay ignored the param group\'s new learning rate'
        del gs, g

        e, c, g = engine(1e-3), engine(1e-3), engine(1e-3)           # (b) eager + StepLR, eager at constant lr, graphed + StepLR
        gs = qt.GraphedTrainStep(g, batch, warmup=1)
        e.step(batch); c.step(batch)                                  # (the warm-up step of the graphed engine)
        se = torch.optim.lr_scheduler.StepLR(e.optim, step_size=2, gamma=0.1)
        sg = torch.optim.lr_scheduler.StepLR(g.optim, step_size=2, gamma=0.1)
        for _ in range(4):
            e.step(batch); se.step()
            c.step(batch)
            gs(batch); sg.step()
        assert g.optim.param_groups[0]['lr'] == pytest.approx(1e-5) and e.optim.param_groups[0]['lr'] == pytest.approx(1e-5)
        d_sched = (g.flat.flat - e.flat.flat).abs()
        d_const = (g.flat.flat - c.flat.flat).abs()
        # steps 3 and 4 ran at 1e-4 instead of 1e-3: against the constant-lr engine most weights differ by ~2 * 9e-4
        # (two engines: Adam turns last-bit gradient differences into +-lr steps; measured median 5.5e-6 at lr = 1e-3)
        assert float(d_sched.median()) < 3e-5 and float((d_sched > 3e-4).float().mean()) < 0.03, (float(d_sched.median()), float(d_sched.max()))
        assert float(d_const.median()) > 5e-4, float(d_const.median())
        assert g.optim.step_count == e.optim.step_count == 5 and int(gs.step_dev.item()) == 5

        sd = e.optim.state_dict()                                     # (c)
        for st in sd['state'].values():
            st['step'] = torch.tensor(40.0)
        g.optim.load_state_dict(sd)
        assert g.optim.step_count == 40
        gs(batch)
        assert int(gs.step_dev.item()) == 41 and g.optim.step_count == 41
    finally:
        qt._lib.load().qt_set_dropout_salt(None)


def test_graphed_forward_follows_weight_changes(qt, synth):
    """A replay of the captured eval forward computes from the weights as they are at replay time: the weights' max|w| records and
    split images are refreshed INSIDE the graph, not taken from host-side caches filled during the warm-up.  (a) parameters in a
    FlatParams buffer, changed by an optimizer step; (b) plain parameters, changed by `load_state_dict` followed by an eager forward
    (which replaces the per-parameter image cache the warm-up filled)."""
    cfg = synth.PathConfig(**{**synth.CONFIG_2.asdict(), 'B': 8})
    P = synth.make_state_dict(cfg, 713)
    P2 = synth.make_state_dict(cfg, 714)
    batch = to_dev(synth.make_inputs(cfg, 713))

    def relerr(a, b):
        return float((a - b).abs().max() / b.abs().max())

    m = qt.QA_TIGER(**cfg.model_kwargs())
    m.load_state_dict(P, strict=True)
    eng = qt.DataParallelStep(m.to(DEV).eval(), lr=1e-2, betas=(0.95, 0.999))       # (a)
    gf = qt.GraphedForward(eng.model, batch)
    first = gf(batch)['out'].clone()
    with torch.no_grad():
        assert relerr(first, eng.model(batch)['out']) < 1e-5
    eng.step(batch)                                                   # lr = 1e-2: every weight moves by ~1e-2
    with torch.no_grad():
        eager = eng.model(batch)['out'].clone()
    assert relerr(first, eager) > 1e-2                                # the step changed the logits
    assert relerr(gf(batch)['out'], eager) < 1e-5, 'the replay used weight images of the captured model state'
    eng.step(batch)
    got = gf(batch)['out'].clone()                                    # replay BEFORE any eager forward refreshed the host-side caches
    with torch.no_grad():
        assert relerr(got, eng.model(batch)['out']) < 1e-5

    m = qt.QA_TIGER(**cfg.model_kwargs())                             # (b)
    m.load_state_dict(P, strict=True)
    m = m.to(DEV).eval()
    gf = qt.GraphedForward(m, batch)
    m.load_state_dict({k: v.to(DEV) for k, v in P2.items()}, strict=True)
    got = gf(batch)['out'].clone()
    with torch.no_grad():
        eager = m(batch)['out'].clone()                               # rebuilds the per-parameter caches
    assert relerr(got, eager) < 1e-5
    torch.zeros(64 << 20, device=DEV).fill_(7.0)                       # recycle whatever the eager pass freed
    assert relerr(gf(batch)['out'], eager) < 1e-5


def test_gradient_sinks_survive_a_second_backward_over_a_retained_graph(qt, synth):
    """`ops.share_grad` tensors (the stacked AVQ input, norm1's output, the projected question) collect their consumers' gradients in
    one buffer that the first consumer hands to autograd.  The buffer must be forgotten once the gradient is complete: a second
    backward over the same (retained) graph has to produce the same gradients as the first, not accumulate into the old buffer."""
    cfg = synth.PathConfig(**{**synth.CONFIG_2.asdict(), 'B': 8})
    model = build(qt, synth, cfg, 713)
    batch = to_dev(synth.make_inputs(cfg, 713))
    loss = qt.CrossEntropyLoss()(model(batch)['out'], batch['label'])
    grads = []
    for _ in range(2):
        model.zero_grad(set_to_none=True)
        loss.backward(retain_graph=True)
        grads.append({n: p.grad.clone() for n, p in model.named_parameters() if p.grad is not None})
    for n, g in grads[0].items():
        assert l2rel(grads[1][n].cpu().numpy(), g.cpu().numpy()) < 1e-5, n


def test_train_steps_leave_no_reference_cycles_behind(qt, synth):
    """Device memory after a train step must not depend on Python's cyclic garbage collector: with the collector switched off, eight more
    steps may not raise the allocated bytes (a tensor that references itself through a cached-operand attribute would keep itself and the
    autograd graph behind it alive — found by the 400-step soak in round 3: +38 MB per step at the primary config)."""
    import gc
    cfg = synth.PathConfig(**{**synth.CONFIG_2.asdict(), 'B': 8})
    model = qt.QA_TIGER(**cfg.model_kwargs())
    model.load_state_dict(synth.make_state_dict(cfg, 713), strict=True)
    eng = qt.DataParallelStep(model.to(DEV).train(), lr=1e-4, betas=(0.95, 0.999))
    batch = to_dev(synth.make_inputs(cfg, 713))
    for _ in range(3):
        eng.step(batch)
    gc.collect()
    torch.cuda.synchronize()
    gc.disable()
    try:
        base = torch.cuda.memory_allocated()
        for _ in range(8):
            eng.step(batch)
        torch.cuda.synchronize()
        grown = torch.cuda.memory_allocated() - base
    finally:
        gc.enable()
    assert grown <= 1 << 20, f'{grown / 1e6:.1f} MB of device memory held by reference cycles after 8 steps'
