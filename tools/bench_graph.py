"""Eager vs HIP-graph forward at several batch sizes (eval, no_grad), at the runtime's default settings (no
DEBUG_CLR_GRAPH_PACKET_CAPTURE: see qa-tiger_amd.GraphedForward); exits non-zero on any mismatch."""
import importlib, os, sys, time
import torch
sys.path.insert(0, '.')
qt = importlib.import_module('qa-tiger_amd'); synth = qt.synth
sizes = [int(a) for a in sys.argv[1:]] or [8, 32, 64, 256]
for B in sizes:
    cfg = synth.PathConfig(**{**synth.CONFIG_2.asdict(), 'B': B})
    torch.manual_seed(713)
    model = qt.QA_TIGER(**cfg.model_kwargs()).cuda().eval()
    batch = synth.fast_device_inputs(cfg, 'cuda', 713)
    other = synth.fast_device_inputs(cfg, 'cuda', 714)
    with torch.no_grad():
        e0 = model(batch)['out'].clone(); e1 = model(other)['out'].clone()
    gf = qt.GraphedForward(model, batch)
    g0 = gf(batch)['out'].clone(); g1 = gf(other)['out'].clone(); g0b = gf(batch)['out'].clone()
    with torch.no_grad():
        model(batch)                                                 # an eager forward between replays
    keep = torch.zeros(1 << 20, device='cuda'); keep2 = keep.clone()  # ... and runtime blits (memset + copy): what made round 1's graph fault
    g1b = gf(other)['out'].clone()
    worst = max(float((a - b).abs().max() / b.abs().max()) for a, b in ((g0, e0), (g1, e1), (g0b, e0), (g1b, e1)))
    assert worst < 1e-5, worst
    def t(fn, n=20):
        for _ in range(3): fn()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(n): fn()
        torch.cuda.synchronize(); return (time.perf_counter() - t0) / n
    with torch.no_grad():
        te = t(lambda: model(batch))
    tg = t(lambda: gf(batch))
    print('B=%4d  eager %7.3f ms = %7.0f QA/s   graph %7.3f ms = %7.0f QA/s   max rel diff vs eager %.1e' % (B, te * 1e3, B / te, tg * 1e3, B / tg, worst), flush=True)
    del gf, model
print('OK')
