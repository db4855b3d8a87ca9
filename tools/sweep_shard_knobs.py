"""Captured train step at a small per-GPU batch under different routing thresholds (ops.plane_min_rows, ops.plane_dw_min_rows,
ops.fold_slab_min_rows): one engine, one capture per variant, 20 replays each.   python tools/sweep_shard_knobs.py [B]"""
import importlib, sys, torch
sys.path.insert(0, '.')
qt = importlib.import_module('qa-tiger_amd'); synth, ops = qt.synth, qt.ops
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
cfg = synth.PathConfig(**{**synth.CONFIG_2.asdict(), 'B': B})
torch.manual_seed(713)
model = qt.QA_TIGER(**cfg.model_kwargs()).cuda().train()
eng = qt.DataParallelStep(model)
batch = synth.fast_device_inputs(cfg, 'cuda', 713)
def timed(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
base = dict(plane_min_rows=ops.plane_min_rows, plane_dw_min_rows=ops.plane_dw_min_rows, fold_slab_min_rows=ops.fold_slab_min_rows)
variants = [dict(), dict(plane_dw_min_rows=8192), dict(plane_min_rows=1024, plane_dw_min_rows=1024), dict(fold_slab_min_rows=2048),
            dict(fold_slab_min_rows=1 << 30), dict(plane_min_rows=4096, plane_dw_min_rows=4096)]
for v in variants:
    for k, val in {**base, **v}.items(): setattr(ops, k, val)
    for _ in range(2): eng.step(batch)
    gs = qt.GraphedTrainStep(eng, batch, warmup=2)
    print('B=%d %-60s captured step %.3f ms' % (B, str(v) if v else '(defaults) ' + str(base), timed(lambda: gs(batch))), flush=True)
    del gs
