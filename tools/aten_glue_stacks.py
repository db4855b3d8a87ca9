"""Where the ATen ops of a train step come from: device time per (op, Python call site).   python tools/aten_glue_stacks.py [B]"""
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
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    eng.step(batch)
    torch.cuda.synchronize()
ev = [e for e in prof.key_averages(group_by_stack_n=8) if e.key.startswith('aten::') and e.self_device_time_total > 0]
ev.sort(key=lambda e: -e.self_device_time_total)
print('aten device time per step: %.3f ms in %d launches-ish' % (sum(e.self_device_time_total for e in ev) / 1e3, sum(e.count for e in ev)))
for e in ev[:60]:
    site = [s for s in e.stack if 'qa-tiger_amd' in s or 'bench' in s or 'tools/' in s][:3]
    print('%-22s x%-3d %8.1f us   %s' % (e.key, e.count, e.self_device_time_total, ' <- '.join(s.split('/')[-1] for s in site)))
