"""One process: N eager train steps at the 8-GPU strong-scaling shard (B=32) — for rocprofv3 --kernel-trace --stats."""
import importlib, sys, torch
sys.path.insert(0, '.')
qt = importlib.import_module('qa-tiger_amd'); synth = qt.synth
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
cfg = synth.PathConfig(**{**synth.CONFIG_2.asdict(), 'B': B})
torch.manual_seed(713)
model = qt.QA_TIGER(**cfg.model_kwargs()).cuda().train()
eng = qt.DataParallelStep(model)
batch = synth.fast_device_inputs(cfg, 'cuda', 713)
for _ in range(6): eng.step(batch)
torch.cuda.synchronize()
if len(sys.argv) > 2 and sys.argv[2] == 'time':          # eager and captured step time (HIP events)
    def timed(fn, n=20):
        for _ in range(3): fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n): fn()
        e1.record(); torch.cuda.synchronize()
        return e0.elapsed_time(e1) / n
    print(f'B={B}: eager step {timed(lambda: eng.step(batch)):.3f} ms')
    gs = qt.GraphedTrainStep(eng, batch)
    print(f'B={B}: captured step {timed(lambda: gs(batch)):.3f} ms')
