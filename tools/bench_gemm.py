"""GEMM micro-benchmark on the path's dominant shapes (MI355X)."""
import importlib, sys, torch
sys.path.insert(0, '.')
qt = importlib.import_module('qa-tiger_amd'); ops = qt.ops
shapes = [  # (M, N, K, ta, tb, label)
    (215040, 2304, 768, 0, 1, 'patch slf in-proj fwd (NT)'),
    (215040, 768, 1024, 0, 1, 'patch_proj fwd (NT)'),
    (30720, 768, 768, 0, 1, 'AVQ out-proj fwd (NT)'),
    (215040, 768, 2304, 0, 0, 'patch slf in-proj dX (NN)'),
    (2304, 768, 215040, 1, 0, 'patch slf in-proj dW (TN)'),
    (768, 1024, 215040, 1, 0, 'patch_proj dW (TN)'),
    (768, 768, 30720, 1, 0, 'AVQ out-proj dW (TN)'),
    (3072, 768, 30720, 1, 0, 'expert W1 dW (TN)'),
    (256, 768, 768, 0, 1, 'quest-side small (NT)'),
    (30720, 2304, 768, 0, 1, 'AVQ in-proj fwd (NT)'),
    (30720, 768, 768, 0, 0, 'AVQ out-proj dX (NN)'),
    (15360, 3072, 768, 0, 1, 'expert W1 fwd (NT)'),
    (19712, 1536, 768, 0, 1, 'words kv fwd (NT)'),
]
modes = sys.argv[1:] or ['fp32', 'f16x3', 'bf16x3']
g = torch.Generator(device='cuda').manual_seed(0)
for M, N, K, ta, tb, label in shapes:
    a = torch.randn((K, M) if ta else (M, K), device='cuda', generator=g)
    b = torch.randn((N, K) if tb else (K, N), device='cuda', generator=g) * 0.05
    out = torch.empty(M, N, device='cuda')
    line = '%-28s %7dx%5dx%7d ' % (label, M, N, K)
    for mode in modes:
        for _ in range(2): ops.gemm(a, b, trans_a=bool(ta), trans_b=bool(tb), out=out, mode=mode)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        n = 5
        e0.record()
        for _ in range(n): ops.gemm(a, b, trans_a=bool(ta), trans_b=bool(tb), out=out, mode=mode)
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / n
        line += ' %s %7.3f ms %6.1f TF' % (mode, ms, 2.0 * M * N * K / ms / 1e9)
    print(line)
