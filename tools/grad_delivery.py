import importlib, sys, torch, collections
sys.path.insert(0, '.')
qt = importlib.import_module('qa-tiger_amd'); ops, synth = qt.ops, qt.synth
cfg = synth.PathConfig(**{**synth.CONFIG_2.asdict(), 'B': 32})
torch.manual_seed(713)
model = qt.QA_TIGER(**cfg.model_kwargs()).cuda().train()
eng = qt.DataParallelStep(model)
batch = synth.fast_device_inputs(cfg, 'cuda', 713)
eng.step(batch)
cnt = collections.Counter()
names = {id(p): n for n, p in model.named_parameters()}
for p in model.parameters():
    if p.requires_grad:
        p.register_hook(lambda g, p=p: cnt.update([names[id(p)]]) if g is not None else None)
eng.step(batch)
for n, c in sorted(cnt.items()):
    print(c, n, tuple(dict(model.named_parameters())[n].shape))
print(len(cnt), 'parameters receive a gradient through autograd')
