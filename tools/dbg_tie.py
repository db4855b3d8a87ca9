"""B=256 primary batch: do repeated evaluations fall into discrete gradient states, and which parameters does the other state touch?"""
import importlib, sys, torch
sys.path.insert(0, '.')
qt = importlib.import_module('qa-tiger_amd'); ops, synth = qt.ops, qt.synth
cfg = synth.CONFIG_2
P = synth.make_state_dict(cfg, 713)
inp = {k: v.cuda() for k, v in synth.make_inputs(cfg, 713).items()}
model = qt.QA_TIGER(**cfg.model_kwargs()); model.load_state_dict(P); model = model.cuda().eval()
def ev():
    model.zero_grad(set_to_none=True)
    qt.CrossEntropyLoss()(model(inp)['out'], inp['label']).backward()
    gauss = {a: getattr(model, a).last['gauss'].clone() for a in ('at_aggregator', 'vt_aggregator')}
    return {n: p.grad.clone() for n, p in model.named_parameters()}, gauss
ref, gref = ev()
states = []
for i in range(14):
    g, ga = ev()
    errs = {n: float((g[n] - ref[n]).norm() / ref[n].norm().clamp_min(1e-30)) for n in g}
    med = sorted(errs.values())[len(errs) // 2]
    states.append(round(med, 5))
    if med > 1e-4 and not getattr(ev, 'shown', False):
        ev.shown = True
        srt = sorted(errs.items(), key=lambda kv: kv[1])
        print('other state: least affected:', [(n, f'{e:.1e}') for n, e in srt[:8]])
        print('other state: most affected:', [(n, f'{e:.1e}') for n, e in srt[-6:]])
        for a in ga:
            d = (ga[a] - gref[a]).abs()                       # gauss [B, k, T]
            b = int(d.amax(dim=(1, 2)).argmax())
            print(a, 'max |gauss diff|', float(d.max()), 'sample', b, 'per-sample max diff top3', [round(float(x), 6) for x in d.amax(dim=(1, 2)).topk(3).values])
print('median error of each evaluation vs the first:', states)
