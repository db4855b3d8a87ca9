"""How much of the HIP-vs-oracle gradient gap is fp32 noise (ReLU gate flips, summation order)?
Compares HIP fp32 and oracle fp32 gradients against the oracle in fp64."""
import importlib, sys, torch
sys.path.insert(0, '.')
qt = importlib.import_module('qa-tiger_amd'); oracle = importlib.import_module('oracle.qatiger_oracle'); synth = qt.synth
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
cfg = synth.PathConfig(**{**synth.CONFIG_2.asdict(), 'B': B})
P = synth.make_state_dict(cfg, 713); inp = synth.make_inputs(cfg, 713)
model = qt.QA_TIGER(**cfg.model_kwargs()); model.load_state_dict(P); model = model.cuda().eval()
dinp = {k: v.cuda() for k, v in inp.items()}
out = model(dinp)['out']; qt.CrossEntropyLoss()(out, dinp['label']).backward()
_, lg32, g32 = oracle.forward_backward(P, inp, cfg.K, cfg.k)
P64 = {n: t.double() for n, t in P.items()}; i64 = {n: (t.double() if t.dtype.is_floating_point else t) for n, t in inp.items()}
_, lg64, g64 = oracle.forward_backward(P64, i64, cfg.K, cfg.k)
print('logits rel err: hip', float((out.detach().cpu().double() - lg64).abs().max() / lg64.abs().max()), ' oracle32', float((lg32.double() - lg64).abs().max() / lg64.abs().max()))
rows = []
for n, p in model.named_parameters():
    r = g64[n]; h = p.grad.cpu().double(); o = g32[n].double()
    mx = r.abs().max().clamp_min(1e-30); l2 = r.norm().clamp_min(1e-30)
    rows.append((float((h - r).abs().max() / mx), float((o - r).abs().max() / mx), float((h - r).norm() / l2), float((o - r).norm() / l2), n))
rows.sort(reverse=True)
print('%-12s %-12s %-12s %-12s  name' % ('hip maxrel', 'orc32 maxrel', 'hip l2rel', 'orc32 l2rel'))
for r in rows[:14]: print('%-12.2e %-12.2e %-12.2e %-12.2e  %s' % r)
print('worst l2: hip %.2e oracle32 %.2e' % (max(r[2] for r in rows), max(r[3] for r in rows)))
