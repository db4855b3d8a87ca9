import importlib, sys, time, torch
sys.path.insert(0, '.')
qt = importlib.import_module('qa-tiger_amd'); synth = qt.synth
cfg = synth.CONFIG_2
torch.manual_seed(713)
model = qt.QA_TIGER(**cfg.model_kwargs()).cuda().train()
eng = qt.DataParallelStep(model)
batches = [synth.fast_device_inputs(cfg, 'cuda', 713 + i) for i in range(4)]
mem0 = None; losses = []
t = time.perf_counter()
for i in range(400):
    l = eng.step(batches[i % 4])
    if i % 50 == 49:
        torch.cuda.synchronize(); losses.append(float(l))
        m = torch.cuda.memory_allocated() / 1e9
        mem0 = mem0 or m
        print('step %3d loss %.4f  mem %.2f GB  %.1f ms/step' % (i + 1, losses[-1], m, (time.perf_counter() - t) / (i + 1) * 1e3), flush=True)
assert all(x == x and abs(x) < 1e3 for x in losses) and losses[-1] < losses[0] and abs(torch.cuda.memory_allocated() / 1e9 - mem0) < 0.5
print('SOAK OK')
