"""Which ATen ops (not libqatiger kernels) run in a train step, by device time and call stack."""
import importlib, sys, torch
from torch.profiler import profile, ProfilerActivity
sys.path.insert(0, '.')
qt = importlib.import_module('qa-tiger_amd'); synth = qt.synth
cfg = synth.CONFIG_2
torch.manual_seed(713)
model = qt.QA_TIGER(**cfg.model_kwargs()).cuda().train()
eng = qt.DataParallelStep(model)
batch = synth.fast_device_inputs(cfg, 'cuda', 713)
for _ in range(2): eng.step(batch)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:   # record_shapes trips over the 64-bit dropout seeds
    eng.step(batch)
    torch.cuda.synchronize()
ev = [e for e in prof.key_averages() if e.key.startswith('aten::') and e.device_time_total > 0]
ev.sort(key=lambda e: -e.self_device_time_total)
tot = sum(e.self_device_time_total for e in ev)
print('aten device time per step: %.3f ms' % (tot / 1e3))
for e in ev[:28]:
    print('%-28s x%-4d %8.1f us' % (e.key, e.count, e.self_device_time_total))
