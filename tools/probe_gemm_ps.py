#!/usr/bin/env python3
"""Phase probes of qt_gemm_ps (QT_PS_DEBUG bits: 1 no write-out, 2 no DMA after a tile's first step, 4 no MFMAs): run once per setting."""
import importlib, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
qt = importlib.import_module('qa-tiger_amd')
ops = qt.ops
dev = 'cuda'
def timeit(fn, n=5):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
g = torch.Generator(device=dev).manual_seed(0)
for name, M, N, K, aks, bks in [('fwd qkv 215040x2304x768', 215040, 2304, 768, False, False), ('dX qkv 215040x768x2304', 215040, 768, 2304, False, True),
                                ('dW qkv 2304x768x215040', 2304, 768, 215040, True, True)]:
    a = torch.randn((K, M) if aks else (M, K), device=dev, generator=g)
    b = torch.randn((K, N) if bks else (N, K), device=dev, generator=g)
    out = torch.empty((M, N), device=dev)
    ia, ib = ops.split_image(a), ops.split_image(b)
    t = sorted(timeit(lambda: ops.gemm_ps(ia, ib, a_kslow=aks, b_kslow=bks, out=out)) for _ in range(3))[1]
    print(f'QT_PS_DEBUG={os.environ.get("QT_PS_DEBUG", "0")} {name}: {t:.3f} ms')
