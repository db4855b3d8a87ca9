"""Per-shape GEMM / attention time of one TSPM train step (HIP events on every launch) + rocprof-free kernel mix."""
import importlib, sys, time, torch
sys.path.insert(0, '.')
qt = importlib.import_module('qa-tiger_amd'); ops, synth = qt.ops, qt.synth
cfg = synth.TspmConfig(**{**synth.TSPM_REF.asdict(), 'B': 256})
torch.manual_seed(713)
model = qt.TSPM(**cfg.model_kwargs()).cuda().train()
eng = qt.DataParallelStep(model)
g = torch.Generator(device='cuda').manual_seed(1)
B = cfg.B
tb = dict(audio=torch.randn(B, cfg.T, cfg.Da, device='cuda', generator=g), video=torch.randn(B, cfg.T, cfg.Dv, device='cuda', generator=g),
          patch=torch.randn(B, cfg.T, cfg.P, cfg.Dp, device='cuda', generator=g), quest=torch.randn(B, 1, cfg.Dq, device='cuda', generator=g),
          prompt=torch.randn(B, 1, cfg.Dq, device='cuda', generator=g), label=torch.randint(0, 42, (B,), device='cuda', generator=g))
for _ in range(3): eng.step(tb)
torch.cuda.synchronize(); t = time.perf_counter()
for _ in range(5): eng.step(tb)
torch.cuda.synchronize(); print('step %.2f ms' % ((time.perf_counter() - t) / 5 * 1e3))
ops.timer = ops.KernelTimer(); n = 2
for _ in range(n): eng.step(tb)
summ = ops.timer.summary(); ops.timer = None
for fam in summ:
    d = summ[fam]; print('== %s: %.2f ms/step, %d launches' % (fam, d['ms'] / n, d['launches'] // n))
    for k, v in sorted(d['by_detail'].items(), key=lambda kv: -kv[1][0])[:12]:
        print('  %-50s x%-3d %7.3f ms' % (str(k), v[2] // n, v[0] / n))
