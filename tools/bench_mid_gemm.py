import importlib, sys, torch
sys.path.insert(0, '.')
qt = importlib.import_module('qa-tiger_amd'); ops = qt.ops
g = torch.Generator(device='cuda').manual_seed(0)
for (M, N, K, ta, tb) in [(3840, 768, 768, 0, 1), (3840, 768, 768, 0, 0), (768, 768, 3840, 1, 0), (3840, 2304, 768, 0, 1), (1920, 3072, 768, 0, 1), (2464, 1536, 768, 0, 1),
                          (7680, 768, 768, 0, 1), (7680, 2304, 768, 0, 1), (15360, 768, 768, 0, 1), (256, 768, 768, 0, 1), (512, 768, 768, 0, 1), (1024, 768, 768, 0, 1)]:
    a = torch.randn((K, M) if ta else (M, K), device='cuda', generator=g); b = torch.randn((N, K) if tb else (K, N), device='cuda', generator=g)
    out = torch.empty(M, N, device='cuda')
    line = '%6dx%5dx%6d t%d%d tiles256=%4d ' % (M, N, K, ta, tb, ((M + 255) // 256) * ((N + 255) // 256))
    for mode in ('bf16x3_t128', 'bf16x3_t256'):
        for _ in range(3): ops.gemm(a, b, trans_a=bool(ta), trans_b=bool(tb), out=out, mode=mode)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): ops.gemm(a, b, trans_a=bool(ta), trans_b=bool(tb), out=out, mode=mode)
        e1.record(); torch.cuda.synchronize()
        line += ' %s %6.1f us' % (mode[-4:], e0.elapsed_time(e1) / 20 * 1e3)
    print(line)
