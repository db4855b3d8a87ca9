"""Reference cycles: allocated device memory and the census of live CUDA tensors over ten train steps WITHOUT gc.collect() (memory held by
cycles grows until Python's cyclic collector runs), then after a collection."""
import importlib, sys, gc, collections, torch
sys.path.insert(0, '.')
qt = importlib.import_module('qa-tiger_amd'); synth = qt.synth; ops = qt.ops
cfg = synth.PathConfig(**{**synth.CONFIG_2.asdict(), 'B': 32})
torch.manual_seed(713)
model = qt.QA_TIGER(**cfg.model_kwargs()).cuda().train()
eng = qt.DataParallelStep(model)
batch = synth.fast_device_inputs(cfg, 'cuda', 713)
def census():
    c = collections.Counter()
    for o in gc.get_objects():
        try:
            if torch.is_tensor(o) and o.is_cuda: c[(tuple(o.shape), str(o.dtype))] += 1
        except Exception: pass
    return c
for _ in range(3): eng.step(batch)
torch.cuda.synchronize(); gc.collect()
m0 = torch.cuda.memory_allocated(); c0 = census()
for _ in range(10): eng.step(batch)
torch.cuda.synchronize()
m1 = torch.cuda.memory_allocated(); c1 = census()
print('allocated: +%.2f MB over 10 steps (no gc.collect)' % ((m1 - m0) / 1e6))
for k in c1:
    if c1[k] - c0.get(k, 0) > 5: print('  grew:', k, c0.get(k, 0), '->', c1[k])
gc.collect(); torch.cuda.synchronize()
print('after gc.collect: +%.2f MB' % ((torch.cuda.memory_allocated() - m0) / 1e6))
