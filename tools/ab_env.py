#!/usr/bin/env python3
"""A/B of one per-call environment switch of qt_gemm_ps (QT_PS_MF, QT_PS_RELAX, QT_PS_ATOMIC, QT_PS_WMT ...) on the SAME images: interleaved
rounds in one process, HIP events, random data; also checks that both settings give the same product.
    python tools/ab_env.py VAR VALUE_A VALUE_B [rounds]        e.g.  python tools/ab_env.py QT_PS_RELAX 0 1"""
import importlib, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
qt = importlib.import_module('qa-tiger_amd')
ops = qt.ops
var, va, vb = sys.argv[1], sys.argv[2], sys.argv[3]
rounds = int(sys.argv[4]) if len(sys.argv) > 4 else 7

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
          (2304, 768, 215040, 1, 1), (768, 1024, 215040, 1, 1), (768, 768, 215040, 1, 1), (768, 768, 30720, 1, 1), (3072, 768, 30720, 1, 1)]
g = torch.Generator(device='cuda').manual_seed(0)
print(f'{"M x N x K":26s} aks bks {var+"="+va+" ms":>16s} {var+"="+vb+" ms":>16s}   {"TF":>7s} {"TF":>7s}  ratio   max|diff|/max')
for M, N, K, aks, bks in shapes:
    a = torch.randn((K, M) if aks else (M, K), device='cuda', generator=g)
    b = torch.randn((K, N) if bks else (N, K), device='cuda', generator=g)
    out = torch.empty((M, N), device='cuda')
    ia, ib = ops.split_image(a), ops.split_image(b)
    f = lambda: ops.gemm_ps(ia, ib, a_kslow=bool(aks), b_kslow=bool(bks), out=out)
    t = {va: [], vb: []}
    res = {}
    for _ in range(rounds):
        for v in (va, vb):
            os.environ[var] = v
            t[v].append(timeit(f))
            if v not in res: res[v] = out.clone()
    os.environ.pop(var, None)
    m = {v: sorted(x)[len(x) // 2] for v, x in t.items()}
    fl = 2.0 * M * N * K / 1e9
    d = float((res[va] - res[vb]).abs().max() / res[va].abs().max())
    print(f'{M}x{N}x{K:<10d}'.ljust(26) + f' {aks}   {bks}  {m[va]:16.3f} {m[vb]:16.3f}   {fl / m[va]:7.1f} {fl / m[vb]:7.1f}  {m[va] / m[vb]:.3f}   {d:.1e}', flush=True)
    del a, b, out, ia, ib, res
