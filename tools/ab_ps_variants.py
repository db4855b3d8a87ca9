#!/usr/bin/env python3
"""A/B/C of several per-call environment settings of qt_gemm_ps on the SAME images: interleaved rounds in one process, HIP events, random
data; every variant's product is compared with the first one's (bit-identical summation order is expected: max|diff| = 0).
    python tools/ab_ps_variants.py [rounds] name:VAR=v,VAR=v ...
    e.g.  python tools/ab_ps_variants.py 5 base: pipe8:QT_PS_PIPE=1 pipe4:QT_PS_PIPE=1,QT_PS_W4=1"""
import importlib, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
qt = importlib.import_module('qa-tiger_amd')
ops = qt.ops
rounds = int(sys.argv[1])
variants = []
for spec in sys.argv[2:]:
    name, _, env = spec.partition(':')
    variants.append((name, dict(kv.split('=') for kv in env.split(',') if kv)))
allvars = sorted({k for _, e in variants for k in e})


def timeit(fn, n=5):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


shapes = [(215040, 2304, 768, 0, 0), (215040, 768, 1024, 0, 0), (215040, 768, 768, 0, 0), (30720, 2304, 768, 0, 0), (30720, 768, 768, 0, 0), (30720, 3072, 768, 0, 0),
          (215040, 768, 2304, 0, 1), (215040, 768, 768, 0, 1), (30720, 768, 2304, 0, 1), (30720, 768, 768, 0, 1), (3840, 768, 768, 0, 0), (26880, 768, 768, 0, 0),
          (2304, 768, 215040, 1, 1), (768, 1024, 215040, 1, 1), (768, 768, 215040, 1, 1), (768, 768, 30720, 1, 1), (3072, 768, 30720, 1, 1),
          (512, 256, 32, 0, 0), (512, 256, 64, 0, 0), (512, 256, 96, 0, 0), (300, 320, 128, 0, 1)]
g = torch.Generator(device='cuda').manual_seed(0)
print(f'{"M x N x K":26s} aks bks ' + ' '.join(f'{n + " ms":>12s}' for n, _ in variants) + '   ' + ' '.join(f'{n + " TF":>10s}' for n, _ in variants) + '   max|diff|/max vs first')
for M, N, K, aks, bks in shapes:
    a = torch.randn((K, M) if aks else (M, K), device='cuda', generator=g)
    b = torch.randn((K, N) if bks else (N, K), device='cuda', generator=g)
    out = torch.empty((M, N), device='cuda')
    ia, ib = ops.split_image(a), ops.split_image(b)
    f = lambda: ops.gemm_ps(ia, ib, a_kslow=bool(aks), b_kslow=bool(bks), out=out)
    t = {n: [] for n, _ in variants}
    res = {}
    for _ in range(rounds):
        for n, env in variants:
            for k in allvars:
                os.environ.pop(k, None)
            os.environ.update(env)
            t[n].append(timeit(f))
            if n not in res:
                res[n] = out.clone()
    for k in allvars:
        os.environ.pop(k, None)
    m = {n: sorted(x)[len(x) // 2] for n, x in t.items()}
    fl = 2.0 * M * N * K / 1e9
    first = res[variants[0][0]]
    d = [float((res[n] - first).abs().max() / first.abs().max()) for n, _ in variants[1:]]
    print(f'{M}x{N}x{K:<10d}'.ljust(26) + f' {aks}   {bks}  ' + ' '.join(f'{m[n]:12.3f}' for n, _ in variants) + '   ' + ' '.join(f'{fl / m[n]:10.1f}' for n, _ in variants)
          + '   ' + ' '.join(f'{x:.1e}' for x in d), flush=True)
    del a, b, out, ia, ib, res
