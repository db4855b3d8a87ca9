import importlib, sys, torch
sys.path.insert(0, '.')
qt = importlib.import_module('qa-tiger_amd'); ops = qt.ops
M, N, K = 215040, 768, 768
g = torch.Generator(device='cuda').manual_seed(0)
a = torch.randn(M, K, device='cuda', generator=g); b = torch.randn(N, K, device='cuda', generator=g) * 0.05
bias = torch.randn(N, device='cuda', generator=g); add = torch.randn(M, N, device='cuda', generator=g)
out = torch.empty(M, N, device='cuda')
def t(label, **kw):
    for _ in range(2): ops.gemm(a, b, out=out, mode='f16x3', **kw)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5): ops.gemm(a, b, out=out, mode='f16x3', **kw)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 5
    print('%-28s %.3f ms %.1f TF' % (label, ms, 2.0 * M * N * K / ms / 1e9))
t('plain'); t('bias', bias=bias); t('bias+add', bias=bias, add=add); t('bias+relu', bias=bias, relu=True)
t('bias+relu+drop', bias=bias, relu=True, drop_p=0.1, seed=5); t('accumulate', accumulate=True)
