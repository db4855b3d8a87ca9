"""Gradients of the default path against the exact-fp32 MFMA path (QT_GEMM=fp32), with switches to bisect a precision regression."""
import importlib, sys, torch
sys.path.insert(0, '.')
qt = importlib.import_module('qa-tiger_amd'); ops, synth = qt.ops, qt.synth
B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
cfg = synth.PathConfig(**{**synth.CONFIG_2.asdict(), 'B': B})
P = synth.make_state_dict(cfg, 713)
inp = {k: v.cuda() for k, v in synth.make_inputs(cfg, 713).items()}
model = qt.QA_TIGER(**cfg.model_kwargs()); model.load_state_dict(P); model = model.cuda().eval()
def grads(**kw):
    keep = dict(ops.precision), ops.plane, ops.use_amax_tags
    for k, v in kw.items():
        if k == 'precision': ops.precision.update(v)
        else: setattr(ops, k, v)
    model.zero_grad(set_to_none=True)
    loss = qt.CrossEntropyLoss()(model(inp)['out'], inp['label']); loss.backward()
    ops.precision.update(keep[0]); ops.plane, ops.use_amax_tags = keep[1], keep[2]
    return {n: p.grad.clone() for n, p in model.named_parameters()}, float(loss)
ref, l0 = grads(precision=dict(fwd='fp32', bwd='fp32'))
for name, kw in [('default', {}), ('no tags', dict(use_amax_tags=False)), ('plane off', dict(plane=False)), ('default again', {})]:
    g, l = grads(**kw)
    errs = sorted(((float((g[n] - ref[n]).norm() / ref[n].norm()), n) for n in g), reverse=True)
    med = errs[len(errs) // 2][0]
    print(f'{name:14s} loss {l:.6f} (fp32 {l0:.6f})  median {med:.2e}  worst ' + ', '.join(f'{n} {e:.1e}' for e, n in errs[:4]))
