import importlib, sys, torch
sys.path.insert(0, '.')
qt = importlib.import_module('qa-tiger_amd'); ops = qt.ops
g = torch.Generator(device='cuda').manual_seed(0)
for (M, N, K, ta, tb) in [(256, 768, 768, 0, 1), (256, 768, 768, 0, 0), (768, 768, 256, 1, 0), (256, 2304, 768, 0, 1), (512, 768, 768, 0, 1), (256, 384, 768, 0, 1), (768, 768, 30720, 1, 0), (30720, 768, 768, 0, 1)]:
    a = torch.randn((K, M) if ta else (M, K), device='cuda', generator=g); b = torch.randn((N, K) if tb else (K, N), device='cuda', generator=g)
    out = torch.empty(M, N, device='cuda')
    for _ in range(3): ops.gemm(a, b, trans_a=bool(ta), trans_b=bool(tb), out=out, mode='bf16x3')
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): ops.gemm(a, b, trans_a=bool(ta), trans_b=bool(tb), out=out, mode='bf16x3')
    e1.record(); torch.cuda.synchronize()
    print('%6dx%5dx%6d t%d%d  %.1f us' % (M, N, K, ta, tb, e0.elapsed_time(e1) / 20 * 1e3))
