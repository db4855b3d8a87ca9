"""Config-4 (T=300, D=1024, K=16, B=128) eval forward: wall time per pass and host issue time of ten consecutive passes."""
import importlib, sys, time, torch, os
sys.path.insert(0, '.')
qt = importlib.import_module('qa-tiger_amd'); synth = qt.synth; ops = qt.ops
cfg = synth.CONFIG_4
torch.manual_seed(713)
model = qt.QA_TIGER(**cfg.model_kwargs()).cuda().eval()
batch = synth.fast_device_inputs(cfg, 'cuda', 713)
host = []
with torch.no_grad():
    for _ in range(3): model(batch)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(10):
        t1 = time.perf_counter(); model(batch); host.append((time.perf_counter() - t1) * 1e3)
    torch.cuda.synchronize(); wall = (time.perf_counter() - t0) / 10 * 1e3
print(sys.argv[1:], 'fwd wall %.1f ms/pass; host issue per pass' % wall, ['%.1f' % h for h in host], 'peak %.1f GB' % (torch.cuda.max_memory_allocated() / 1e9))
