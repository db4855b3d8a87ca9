#!/usr/bin/env python3
"""qt_gemm_ps time vs split_k on the shapes of the 8-GPU strong-scaling shard (B=32 per GPU): what the launcher's cost model should pick."""
import importlib, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
qt = importlib.import_module('qa-tiger_amd'); ops = qt.ops
dev = 'cuda'
def timeit(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
# (M, N, K, a_kslow, b_kslow)
shapes = [(768, 768, 3840, 1, 1), (2304, 768, 3840, 1, 1), (768, 768, 26880, 1, 1), (2304, 768, 26880, 1, 1), (768, 1024, 26880, 1, 1),
          (3840, 768, 768, 0, 1), (3840, 768, 2304, 0, 1), (26880, 768, 768, 0, 1), (26880, 768, 2304, 0, 1),
          (3840, 768, 768, 0, 0), (3840, 2304, 768, 0, 0), (26880, 768, 768, 0, 0), (26880, 2304, 768, 0, 0)]
g = torch.Generator(device=dev).manual_seed(0)
for M, N, K, aks, bks in shapes:
    a = torch.randn((K, M) if aks else (M, K), device=dev, generator=g)
    b = torch.randn((K, N) if bks else (N, K), device=dev, generator=g)
    ia, ib = ops.split_image(a), ops.split_image(b)
    out = torch.empty((M, N), device=dev)
    res = []
    for sk in (0, 1, 4, 8, 16):
        if sk > 1 and K // sk < 128: continue
        res.append((sk, timeit(lambda: ops.gemm_ps(ia, ib, a_kslow=bool(aks), b_kslow=bool(bks), out=out, split_k=sk))))
    tx = timeit(lambda: ops.gemm(a, b, trans_a=bool(aks), trans_b=not bks, out=out, mode='f16x3'))
    print(f'{M}x{N}x{K} aks={aks} bks={bks}: x3(auto)={tx:.0f}us  ' + '  '.join(f'sk{sk}={t:.0f}us' for sk, t in res))
