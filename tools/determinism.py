import importlib, sys, torch
sys.path.insert(0, '.')
qt = importlib.import_module('qa-tiger_amd'); synth = qt.synth
cfg = synth.CONFIG_2
torch.manual_seed(713)
model = qt.QA_TIGER(**cfg.model_kwargs()).cuda().eval()
batch = synth.fast_device_inputs(cfg, 'cuda', 713)
with torch.no_grad():
    outs = [model(batch)['out'].clone() for _ in range(4)]
print('forward bit-identical:', all(torch.equal(outs[0], o) for o in outs[1:]), [float((outs[0]-o).abs().max()) for o in outs[1:]])
gs = []
for _ in range(3):
    model.zero_grad(set_to_none=True)
    out = model(batch)['out']
    qt.CrossEntropyLoss()(out, batch['label']).backward()
    gs.append(torch.cat([p.grad.flatten() for p in model.parameters() if p.grad is not None]).clone())
print('plain-autograd gradients bit-identical:', all(torch.equal(gs[0], g) for g in gs[1:]), [float((gs[0]-g).abs().max() / gs[0].abs().max()) for g in gs[1:]])
eng = qt.DataParallelStep(model)
fs = []
for _ in range(3):
    eng.forward_backward(batch)
    fs.append(eng.flat.grad.clone())
print('engine gradients bit-identical:', all(torch.equal(fs[0], g) for g in fs[1:]), [float((fs[0]-g).abs().max() / fs[0].abs().max()) for g in fs[1:]])
names = [n for n, p in model.named_parameters() if p.requires_grad]
for (n, gv0), gv1 in zip(zip(names, [fs[0][o:o+p.numel()] for p, o in zip(eng.flat.params, eng.flat.offsets)]), [fs[1][o:o+p.numel()] for p, o in zip(eng.flat.params, eng.flat.offsets)]):
    if not torch.equal(gv0, gv1): print('  differs:', n, float((gv0-gv1).abs().max() / gv0.abs().max().clamp_min(1e-30)))
