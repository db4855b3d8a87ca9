"""Host-side (Python) cost of a train step at a small batch: cProfile over 20 steps at B=32."""
import cProfile, importlib, pstats, sys, time, torch
sys.path.insert(0, '.')
qt = importlib.import_module('qa-tiger_amd'); synth = qt.synth
cfg = synth.PathConfig(**{**synth.CONFIG_2.asdict(), 'B': 32})
torch.manual_seed(713)
model = qt.QA_TIGER(**cfg.model_kwargs()).cuda().train()
eng = qt.DataParallelStep(model)
batch = synth.fast_device_inputs(cfg, 'cuda', 713)
for _ in range(3): eng.step(batch)
torch.cuda.synchronize(); t = time.perf_counter()
for _ in range(20): eng.step(batch)
t_issue = (time.perf_counter() - t) / 20
torch.cuda.synchronize(); t_all = (time.perf_counter() - t) / 20
print('B=32: host issue time %.2f ms/step, wall %.2f ms/step' % (t_issue * 1e3, t_all * 1e3))
pr = cProfile.Profile(); pr.enable()
for _ in range(20): eng.step(batch)
torch.cuda.synchronize(); pr.disable()
st = pstats.Stats(pr); st.sort_stats('tottime').print_stats(18)
