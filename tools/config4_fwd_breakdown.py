import importlib, sys, time, torch, os
sys.path.insert(0, '.')
qt = importlib.import_module('qa-tiger_amd'); synth = qt.synth; ops = qt.ops
cfg = synth.CONFIG_4
torch.manual_seed(713)
model = qt.QA_TIGER(**cfg.model_kwargs()).cuda().eval()
batch = synth.fast_device_inputs(cfg, 'cuda', 713)
def t(n=5):
    with torch.no_grad():
        for _ in range(2): model(batch)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(n): model(batch)
        torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
print('default fwd ms', t())
ops.timer = ops.KernelTimer()
with torch.no_grad(): model(batch)
summ = ops.timer.summary(); ops.timer = None
for fam, d in summ.items():
    print(fam, '%.2f ms' % d['ms'], d['launches'])
    for k, v in sorted(d['by_detail'].items(), key=lambda kv: -kv[1][0])[:8]:
        print('   ', k, 'x%d %.3f ms' % (v[2], v[0]))
