#!/usr/bin/env python3
"""A/B of qt_gemm_ps's two MFMA shapes on the SAME images: v_mfma_f32_32x32x16_f16 (QT_PS_MF=0) against v_mfma_f32_16x16x32_f16
(QT_PS_MF=1, the default), interleaved rounds in one process, HIP events, random data (MI355X_MICROARCH.md "DVFS give-back" item 7: the clock the chip
holds under a matrix-dense load depends on the shape).  Also checks that both forms give the same product.   python tools/ab_mfma_shape.py [rounds]
Needs a library built with QT_PS_WITH_MF0=1 in the environment (the 32x32x16 instances are not part of the default build any more)."""
import importlib, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
qt = importlib.import_module('qa-tiger_amd')
ops = qt.ops
dev = 'cuda'
rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 7

def timeit(fn, n=5):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n

shapes = [(215040, 2304, 768, 0, 0), (215040, 768, 1024, 0, 0), (215040, 768, 768, 0, 0), (30720, 2304, 768, 0, 0), (30720, 768, 768, 0, 0), (30720, 3072, 768, 0, 0),
          (215040, 768, 2304, 0, 1), (215040, 768, 768, 0, 1), (30720, 768, 2304, 0, 1), (30720, 768, 768, 0, 1), (3840, 768, 768, 0, 0),
          (2304, 768, 215040, 1, 1), (768, 1024, 215040, 1, 1), (768, 768, 215040, 1, 1), (768, 768, 30720, 1, 1), (3072, 768, 30720, 1, 1)]
g = torch.Generator(device=dev).manual_seed(0)
print(f'{"M x N x K":26s} aks bks {"32x32x16 ms":>12s} {"16x16x32 ms":>12s}   {"TF":>7s} {"TF":>7s}  ratio   max|diff|/max')
for M, N, K, aks, bks in shapes:
    a = torch.randn((K, M) if aks else (M, K), device=dev, generator=g)
    b = torch.randn((K, N) if bks else (N, K), device=dev, generator=g)
    out = torch.empty((M, N), device=dev)
    ia, ib = ops.split_image(a), ops.split_image(b)
    f = lambda: ops.gemm_ps(ia, ib, a_kslow=bool(aks), b_kslow=bool(bks), out=out)
    t = {0: [], 1: []}
    res = {}
    for _ in range(rounds):
        for mf in (0, 1):
            os.environ['QT_PS_MF'] = str(mf)
            t[mf].append(timeit(f))
            res[mf] = out.clone() if mf not in res else res[mf]
    os.environ.pop('QT_PS_MF', None)
    m = {w: sorted(v)[len(v) // 2] for w, v in t.items()}
    fl = 2.0 * M * N * K / 1e9
    d = float((res[0] - res[1]).abs().max() / res[0].abs().max())
    print(f'{M}x{N}x{K:<10d}'.ljust(26) + f' {aks}   {bks}  {m[0]:12.3f} {m[1]:12.3f}   {fl / m[0]:7.1f} {fl / m[1]:7.1f}  {m[0] / m[1]:.3f}   {d:.1e}', flush=True)
    del a, b, out, ia, ib, res
