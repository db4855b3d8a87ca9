#!/usr/bin/env python3
"""A/B of the pre-split GEMM (qt_gemm_ps) against the split-in-the-consumer GEMM (qt_gemm_x3) on the path's big shapes, inside
one process, interleaved rounds, HIP events.  Also times the producer passes (amax, split) the pre-split pipeline adds.
    python tools/bench_gemm_ps.py [rounds]"""
import importlib, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
qt = importlib.import_module('qa-tiger_amd')
ops = qt.ops
dev = 'cuda'
rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 5

def timeit(fn, n=3):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n

# (name, M, N, K, a_kslow, b_kslow, x3 mode)
shapes = [('fwd qkv   215040x2304x768', 215040, 2304, 768, False, False, 'f16x3'),
          ('fwd proj  215040x768x1024', 215040, 768, 1024, False, False, 'f16x3'),
          ('fwd out   215040x768x768 ', 215040, 768, 768, False, False, 'f16x3'),
          ('dX  qkv   215040x768x2304', 215040, 768, 2304, False, True, 'bf16x3'),
          ('dX  out   215040x768x768 ', 215040, 768, 768, False, True, 'bf16x3'),
          ('dW  qkv   2304x768x215040', 2304, 768, 215040, True, True, 'bf16x3'),
          ('dW  proj  768x1024x215040', 768, 1024, 215040, True, True, 'bf16x3'),
          ('fwd avq   30720x2304x768 ', 30720, 2304, 768, False, False, 'f16x3'),
          ('fwd avq   30720x768x768  ', 30720, 768, 768, False, False, 'f16x3'),
          ('dW  avq   768x768x30720  ', 768, 768, 30720, True, True, 'bf16x3'),
          ('fwd exp   30720x3072x768 ', 30720, 3072, 768, False, False, 'f16x3')]
g = torch.Generator(device=dev).manual_seed(0)
print(f'{"shape":28s} {"x3 ms":>8s} {"ps ms":>8s} {"x3 TF":>7s} {"ps TF":>7s} {"ratio":>6s}   split(A) ms  amax(A) ms')
for name, M, N, K, aks, bks, mode in shapes:
    a = torch.randn((K, M) if aks else (M, K), device=dev, generator=g)
    b = torch.randn((K, N) if bks else (N, K), device=dev, generator=g)
    out = torch.empty((M, N), device=dev)
    ia, ib = ops.split_image(a), ops.split_image(b)
    f_x3 = lambda: ops.gemm(a, b, trans_a=aks, trans_b=not bks, out=out, mode=mode)
    f_ps = lambda: ops.gemm_ps(ia, ib, a_kslow=aks, b_kslow=bks, out=out)
    r1 = out.clone(); f_x3(); r1.copy_(out); f_ps()
    err = float((out - r1).abs().max() / r1.abs().max())
    tx, tp = [], []
    for _ in range(rounds):
        tx.append(timeit(f_x3)); tp.append(timeit(f_ps))
    tx, tp = sorted(tx)[len(tx) // 2], sorted(tp)[len(tp) // 2]
    ts = timeit(lambda: ops.split_image(a)); ta = timeit(lambda: ops.amax_of(a))
    fl = 2.0 * M * N * K / 1e9
    print(f'{name:28s} {tx:8.3f} {tp:8.3f} {fl / tx:7.1f} {fl / tp:7.1f} {tx / tp:6.2f}   {ts:8.3f}  {ta:8.3f}   (x3 vs ps diff {err:.1e})')
    del a, b, out, ia, ib
