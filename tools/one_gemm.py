"""Launch one GEMM shape a few times (for rocprofv3 --pmc passes)."""
import importlib, sys, torch
sys.path.insert(0, '.')
qt = importlib.import_module('qa-tiger_amd'); ops = qt.ops
M, N, K, ta, tb = [int(x) for x in sys.argv[1:6]]; mode = sys.argv[6]
g = torch.Generator(device='cuda').manual_seed(0)
a = torch.randn((K, M) if ta else (M, K), device='cuda', generator=g)
b = torch.randn((N, K) if tb else (K, N), device='cuda', generator=g) * 0.05
out = torch.empty(M, N, device='cuda')
if mode == 'ps':      # pre-split operands (csrc/gemm_plane.hip): tA/tB mean "K-slow image" here (tB=1 is the Linear weight layout = row form)
    ia, ib = ops.split_image(a), ops.split_image(b)
    for _ in range(3): ops.gemm_ps(ia, ib, a_kslow=bool(ta), b_kslow=not bool(tb), out=out)
else:
    for _ in range(3): ops.gemm(a, b, trans_a=bool(ta), trans_b=bool(tb), out=out, mode=mode)
torch.cuda.synchronize()
