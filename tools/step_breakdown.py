"""Per-shape GEMM/attention time of one train step (HIP events on every launch)."""
import importlib, sys, torch
sys.path.insert(0, '.')
qt = importlib.import_module('qa-tiger_amd'); ops, synth = qt.ops, qt.synth
cfg = synth.CONFIG_2
torch.manual_seed(713)
model = qt.QA_TIGER(**cfg.model_kwargs()).cuda().train()
eng = qt.DataParallelStep(model)
batch = synth.fast_device_inputs(cfg, 'cuda', 713)
for _ in range(2): eng.step(batch)
ops.timer = ops.KernelTimer(); n = 2
for _ in range(n): eng.step(batch)
summ = ops.timer.summary(); ops.timer = None
for fam in ('gemm', 'split', 'attn_fwd', 'attn_bwd'):
    if fam not in summ: continue
    d = summ[fam]; print('== %s: %.2f ms/step, %d launches' % (fam, d['ms'] / n, d['launches'] // n))
    rows = sorted(d['by_detail'].items(), key=lambda kv: -kv[1][0])
    acc = 0
    for k, v in rows[:int(sys.argv[1]) if len(sys.argv) > 1 else 28]:
        acc += v[0] / n
        rate = v[1] / (v[0] * 1e-3) / (1e12 if fam == 'gemm' else 1e9)
        print('  %-46s x%-2d %7.3f ms  %7.1f %s  (cum %.1f)' % (str(k), v[2] // n, v[0] / n, rate, 'TF' if fam == 'gemm' else 'GB/s', acc))
