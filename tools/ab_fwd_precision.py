"""A/B inside one process: forward GEMMs on fp16 parts vs bf16 parts (logit error vs golden, step time)."""
import importlib, sys, time, numpy as np, torch
sys.path.insert(0, '.')
qt = importlib.import_module('qa-tiger_amd'); ops, synth = qt.ops, qt.synth
g = dict(np.load('tests/golden/config2_b8.npz'))
cfg8 = synth.PathConfig(**{**synth.CONFIG_2.asdict(), 'B': 8})
m8 = qt.QA_TIGER(**cfg8.model_kwargs()); m8.load_state_dict(synth.make_state_dict(cfg8, 713)); m8 = m8.cuda().eval()
inp8 = {k: v.cuda() for k, v in synth.make_inputs(cfg8, 713).items()}
cfg = synth.CONFIG_2
torch.manual_seed(713)
model = qt.QA_TIGER(**cfg.model_kwargs()).cuda().train()
eng = qt.DataParallelStep(model)
batch = synth.fast_device_inputs(cfg, 'cuda', 713)
for rep in range(2):
    for fwd in ('f16x3', 'bf16x3'):
        ops.precision['fwd'] = fwd
        with torch.no_grad():
            out = m8(inp8)['out'].cpu().numpy()
        err = np.abs(out - g['logits']).max() / np.abs(g['logits']).max()
        for _ in range(2): eng.step(batch)
        torch.cuda.synchronize(); t = time.perf_counter()
        for _ in range(8): eng.step(batch)
        torch.cuda.synchronize(); dt = (time.perf_counter() - t) / 8
        print(f'fwd={fwd}: logit rel err {err:.2e}  step {dt*1e3:.2f} ms  {256/dt:.0f} QA/s', flush=True)
