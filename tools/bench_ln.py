"""Residual + dropout + LayerNorm kernels at the shapes of the path (HIP events around eager calls): forward and backward of
AVQCrossAttn.norm1 (four addends, three dropped), norm2 (two addends), PatchSelecter's anorm / vnorm (one addend)."""
import importlib, sys, torch
sys.path.insert(0, '.')
qt = importlib.import_module('qa-tiger_amd'); ops = qt.ops
def gtime(fn, n=30):
    """HIP events around n eager calls (the big shapes are GPU-bound; the 256-row one shows the host's issue time instead)."""
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
g = torch.Generator(device='cuda').manual_seed(0)
for rows, D, n_add, n_drop in [(30720, 768, 4, 3), (30720, 768, 2, 1), (15360, 768, 1, 0), (256, 768, 3, 1)]:
    xs = [torch.randn(rows, D, device='cuda', generator=g, requires_grad=True) for _ in range(n_add)]
    gamma, beta = torch.ones(D, device='cuda', requires_grad=True), torch.zeros(D, device='cuda', requires_grad=True)
    drops = [False] * (n_add - n_drop) + [True] * n_drop
    dy = torch.randn(rows, D, device='cuda', generator=g)
    y = ops.add_layer_norm(xs, gamma, beta, drops=drops, drop_p=0.1)
    tf = gtime(lambda: ops.add_layer_norm([x.detach() for x in xs], gamma.detach(), beta.detach(), drops=drops, drop_p=0.1))
    def bwd():
        for t in xs + [gamma, beta]: t.grad = None
        y.backward(dy, retain_graph=True)
    tb = gtime(bwd)
    fb, bb = 4.0 * rows * D * (n_add + 2), 4.0 * rows * D * (2 + n_add)
    print('rows %6d D %4d addends %d (dropped %d):  fwd %6.1f us = %5.2f TB/s   bwd %6.1f us = %5.2f TB/s' % (rows, D, n_add, n_drop, tf, fb / tf / 1e6, tb, bb / tb / 1e6), flush=True)
