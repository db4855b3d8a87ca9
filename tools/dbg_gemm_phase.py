import importlib, sys, torch
sys.path.insert(0, '.')
qt = importlib.import_module('qa-tiger_amd'); ops = qt.ops
for (M, N, K, ta, tb) in [(215040, 2304, 768, 0, 1), (30720, 768, 768, 0, 1)]:
    g = torch.Generator(device='cuda').manual_seed(0)
    a = torch.randn((K, M) if ta else (M, K), device='cuda', generator=g); b = torch.randn((N, K) if tb else (K, N), device='cuda', generator=g) * 0.05
    out = torch.empty(M, N, device='cuda')
    for label, alpha in [('normal', 1.0), ('no global loads', -12345.0), ('no convert/LDS store', -12346.0), ('no epilogue', -12347.0)]:
        for _ in range(2): ops.gemm(a, b, trans_a=bool(ta), trans_b=bool(tb), out=out, mode='bf16x3', alpha=alpha)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5): ops.gemm(a, b, trans_a=bool(ta), trans_b=bool(tb), out=out, mode='bf16x3', alpha=alpha)
        e1.record(); torch.cuda.synchronize()
        print('%dx%dx%d %-22s %.3f ms' % (M, N, K, label, e0.elapsed_time(e1) / 5))
