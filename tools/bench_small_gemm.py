"""Small GEMMs of the path (the per-sample layers, M = batch size; small-batch shards): time per launch (us) on the skinny kernel
(gemm_skinny.hip, forced) against the 128x128 / 256x256 tiles of gemm_split.hip (skinny forbidden, auto split-K) and the exact-fp32
kernel; `route` is what qt_gemm_x3 picks by itself.  Shapes: argv pairs M,N,K,tA,tB[,batch] or the built-in list."""
import importlib, sys, torch
sys.path.insert(0, '.')
qt = importlib.import_module('qa-tiger_amd'); ops = qt.ops
g = torch.Generator(device='cuda').manual_seed(0)
def t(fn, n=20, reps=5):
    """GPU time per launch: n launches captured into one HIP graph (a Python/ctypes call costs ~10 us, more than these kernels), replayed."""
    for _ in range(3): fn()
    torch.cuda.synchronize()
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gr):
        for _ in range(n): fn()
    gr.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): gr.replay()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / (n * reps) * 1e3
SHAPES = [(256, 768, 768, 0, 1), (256, 768, 768, 0, 0), (768, 768, 256, 1, 0), (256, 2304, 768, 0, 1), (512, 768, 768, 0, 1), (256, 384, 768, 0, 1),
          (768, 384, 256, 1, 0), (256, 768, 384, 0, 0), (256, 3072, 768, 0, 0), (256, 768, 3072, 0, 1), (768, 3072, 256, 1, 0), (512, 768, 3072, 0, 1),
          (512, 3072, 768, 0, 0), (768, 3072, 512, 1, 0), (1536, 768, 512, 1, 0), (512, 1536, 768, 0, 1), (512, 768, 1536, 0, 0), (256, 42, 768, 0, 1),
          (32, 768, 768, 0, 1), (768, 768, 32, 1, 0), (64, 768, 768, 0, 1), (960, 768, 768, 0, 1), (1920, 768, 768, 0, 1), (768, 768, 1920, 1, 0),
          (3840, 768, 768, 0, 1), (768, 768, 3840, 1, 0), (3840, 768, 768, 0, 0), (2464, 1536, 768, 0, 1), (768, 768, 7680, 1, 0),
          (256, 96, 768, 0, 1, 8), (96, 768, 256, 1, 0, 8), (256, 768, 96, 0, 0, 8), (3840, 96, 768, 0, 1, 8), (96, 768, 3840, 1, 0, 8), (3840, 768, 96, 0, 0, 8)]
if len(sys.argv) > 1:
    SHAPES = [tuple(int(v) for v in s.split(',')) for s in sys.argv[1:]]
for shp in SHAPES:
    M, N, K, ta, tb = shp[:5]
    nb = shp[5] if len(shp) > 5 else 1
    # batched problems as the fold GEMMs lay them out: heads are column / row blocks of one tensor
    a = torch.randn((nb, K, M) if ta else (nb, M, K), device='cuda', generator=g)
    b = torch.randn((nb, N, K) if tb else (nb, K, N), device='cuda', generator=g)
    out = torch.empty(nb, M, N, device='cuda')
    batch = (nb, a[0].numel(), b[0].numel(), M * N) if nb > 1 else None
    kw = dict(trans_a=bool(ta), trans_b=bool(tb), out=out[0], batch=batch)
    res = []
    for name, mode in (('skinny', 'f16x3_sk'), ('tiles', 'f16x3_nosk'), ('route', 'f16x3'), ('fp32', 'fp32')):
        res.append('%s=%.1f' % (name, t(lambda: ops.gemm(a[0], b[0], mode=mode, **kw))))
    ref = (a.double().transpose(1, 2) if ta else a.double()) @ (b.double().transpose(1, 2) if tb else b.double())
    ops.gemm(a[0], b[0], mode='f16x3_sk', **kw)
    err = float((out.double() - ref).abs().max() / ref.abs().max())
    print('%5dx%5dx%5d t%d%d b%d  ' % (M, N, K, ta, tb, nb) + '  '.join(res) + '  (us)   skinny rel err %.1e' % err, flush=True)
