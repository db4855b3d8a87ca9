#!/usr/bin/env python3
"""A/B of qt_gemm_ps's 128- and 192-row tiles (QT_PS_WMT=2, 3) against the 256-row tile (QT_PS_WMT=4) and the launcher's own choice (unset),
interleaved rounds, HIP events.   python tools/ab_tile_height.py [rounds]"""
import importlib, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
qt = importlib.import_module('qa-tiger_amd')
ops = qt.ops
dev = 'cuda'
rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 5

def timeit(fn, n=5):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n

shapes = [(30720, 768, 768, False), (30720, 768, 768, True), (30720, 2304, 768, False), (30720, 768, 2304, True), (30720, 3072, 768, False),
          (30720, 768, 3072, True), (15360, 768, 768, False), (15360, 3072, 768, False), (19712, 1536, 768, False), (30720, 384, 768, False),
          (215040, 768, 768, False), (215040, 2304, 768, False), (215040, 768, 2304, True), (215040, 768, 1024, False),
          (3840, 768, 768, False), (3840, 768, 768, True), (3840, 2304, 768, False), (3840, 768, 2304, True), (3840, 3072, 768, False), (7680, 768, 768, False), (26880, 768, 768, False), (26880, 2304, 768, False)]
g = torch.Generator(device=dev).manual_seed(0)
print(f'{"M x N x K":24s} bks  {"256 ms":>8s} {"192 ms":>8s} {"128 ms":>8s} {"auto ms":>8s}   {"256 TF":>7s} {"192 TF":>7s} {"128 TF":>7s} {"auto TF":>7s}')
for M, N, K, bks in shapes:
    a = torch.randn((M, K), device=dev, generator=g)
    b = torch.randn((K, N) if bks else (N, K), device=dev, generator=g)
    out = torch.empty((M, N), device=dev)
    ia, ib = ops.split_image(a), ops.split_image(b)
    f = lambda: ops.gemm_ps(ia, ib, b_kslow=bks, out=out)
    t = {4: [], 3: [], 2: [], 0: []}
    for _ in range(rounds):
        for w in (4, 3, 2, 0):
            if w: os.environ['QT_PS_WMT'] = str(w)
            else: os.environ.pop('QT_PS_WMT', None)
            t[w].append(timeit(f))
    os.environ.pop('QT_PS_WMT', None)
    m = {w: sorted(v)[len(v) // 2] for w, v in t.items()}
    fl = 2.0 * M * N * K / 1e9
    print(f'{M}x{N}x{K:<10d}'.ljust(24) + f' {int(bks)}   {m[4]:8.3f} {m[3]:8.3f} {m[2]:8.3f} {m[0]:8.3f}   {fl / m[4]:7.1f} {fl / m[3]:7.1f} {fl / m[2]:7.1f} {fl / m[0]:7.1f}')
    del a, b, out, ia, ib
