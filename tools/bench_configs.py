"""Train-step time of the other BASELINE configs (parity cases, not the bench line): config 3's per-GPU shard and config 4."""
import importlib, sys, time, torch
sys.path.insert(0, '.')
qt = importlib.import_module('qa-tiger_amd'); synth = qt.synth
for label, cfg in [('config3 shard (B=32/GPU of the 8-GPU global 256)', synth.PathConfig(**{**synth.CONFIG_2.asdict(), 'B': 32})),
                   ('config4 (B=128, T=300, D=1024, K=16)', synth.CONFIG_4),
                   ('config5 shape (B=512, T=60, D=768) on the fp32-accurate path', synth.PathConfig(**{**synth.CONFIG_2.asdict(), 'B': 512}))]:
    torch.manual_seed(713)
    model = qt.QA_TIGER(**cfg.model_kwargs()).cuda().train()
    eng = qt.DataParallelStep(model)
    batch = synth.fast_device_inputs(cfg, 'cuda', 713)
    for _ in range(2): eng.step(batch)
    torch.cuda.synchronize(); t = time.perf_counter()
    n = 5
    for _ in range(n): eng.step(batch)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t) / n
    gdt = None
    if cfg.B <= 64:                                  # host-bound regime: the whole train step as ONE HIP graph (qt.GraphedTrainStep)
        model.train()
        gs = qt.GraphedTrainStep(eng, batch)
        for _ in range(3): gs(batch)
        torch.cuda.synchronize(); t = time.perf_counter()
        for _ in range(20): gs(batch)
        torch.cuda.synchronize(); gdt = (time.perf_counter() - t) / 20
        del gs
    model.eval()
    with torch.no_grad():
        for _ in range(2): model(batch)
        torch.cuda.synchronize(); t = time.perf_counter()
        for _ in range(n): model(batch)
        torch.cuda.synchronize(); ft = (time.perf_counter() - t) / n
    print('%-62s fwd+bwd %8.2f ms = %7.0f QA/s   fwd %7.2f ms = %7.0f QA/s   peak mem %.1f GB' % (label, dt * 1e3, cfg.B / dt, ft * 1e3, cfg.B / ft, torch.cuda.max_memory_allocated() / 1e9)
          + ('' if gdt is None else '   captured train step %.2f ms = %.0f QA/s' % (gdt * 1e3, cfg.B / gdt)), flush=True)
    del eng, model, batch; torch.cuda.empty_cache(); torch.cuda.reset_peak_memory_stats()
