"""Small-M GEMMs of the path (256-row layers): time per launch on the split kernel for every split_k, on the exact-fp32 kernel, and with amax operands."""
import importlib, sys, torch
sys.path.insert(0, '.')
qt = importlib.import_module('qa-tiger_amd'); ops = qt.ops
g = torch.Generator(device='cuda').manual_seed(0)
def t(fn, n=30):
    for _ in range(3): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
for (M, N, K, ta, tb) in [(256, 768, 768, 0, 1), (256, 768, 768, 0, 0), (768, 768, 256, 1, 0), (256, 2304, 768, 0, 1), (512, 768, 768, 0, 1), (256, 384, 768, 0, 1),
                          (256, 3072, 768, 0, 1), (256, 768, 3072, 0, 0), (3072, 768, 256, 1, 0), (960, 768, 768, 0, 1)]:
    a = torch.randn((K, M) if ta else (M, K), device='cuda', generator=g); b = torch.randn((N, K) if tb else (K, N), device='cuda', generator=g)
    out = torch.empty(M, N, device='cuda')
    am = ops.amax_of(a, keep=True)
    res = ['auto=%.0f' % t(lambda: ops.gemm(a, b, trans_a=bool(ta), trans_b=bool(tb), out=out, mode='f16x3'))]
    res.append('auto+amax=%.0f' % t(lambda: ops.gemm(a, b, trans_a=bool(ta), trans_b=bool(tb), out=out, mode='f16x3', a_amax=am)))
    for sk in (1, 2, 4, 8):
        res.append('sk%d=%.0f' % (sk, t(lambda: ops.gemm(a, b, trans_a=bool(ta), trans_b=bool(tb), out=out, mode='f16x3', split_k=sk))))
    res.append('fp32=%.0f' % t(lambda: ops.gemm(a, b, trans_a=bool(ta), trans_b=bool(tb), out=out, mode='fp32')))
    print('%5dx%5dx%5d t%d%d  ' % (M, N, K, ta, tb) + '  '.join(res) + '  (us)')
