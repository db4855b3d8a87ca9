import importlib, sys, torch
sys.path.insert(0, '.')
qt = importlib.import_module('qa-tiger_amd'); ops = qt.ops
def rnd(*shape, seed=0):
    g = torch.Generator(device='cpu').manual_seed(seed)
    return torch.randn(*shape, generator=g).cuda()
for rows, D, relu in [(300, 768, True), (300, 768, False), (2048, 512, False), (5, 1024, False)]:
    xs = [rnd(rows, D, seed=40 + i).requires_grad_(True) for i in range(3)]
    gamma = (1 + 0.1 * rnd(D, seed=44)).requires_grad_(True); beta = rnd(D, seed=45).requires_grad_(True)
    scales = [1.0, 0.5, 2.0]
    y = ops.add_layer_norm(xs, gamma, beta, scales=scales, relu_out=relu)
    gy = rnd(rows, D, seed=46); y.backward(gy)
    xr = [x.detach().double().requires_grad_(True) for x in xs]
    gr, br = gamma.detach().double().requires_grad_(True), beta.detach().double().requires_grad_(True)
    s = sum(sc * x for sc, x in zip(scales, xr))
    ref = torch.nn.functional.layer_norm(s, (D,), gr, br, 1e-5)
    if relu: ref = torch.relu(ref)
    ref.backward(gy.double())
    for i in range(3):
        err = (xs[i].grad.double() - xr[i].grad).abs().amax(1)
        bad = (err > 1e-4 * xr[i].grad.abs().max()).nonzero().flatten()
        print(rows, D, relu, 'x', i, 'max err', float(err.max()), 'bad rows', bad.tolist()[:10], len(bad))
    print('  gamma err', float((gamma.grad.double() - gr.grad).abs().max()), 'beta err', float((beta.grad.double() - br.grad).abs().max()),
          'mask mismatches', int(((y > 0) != (ref > 0)).sum()) if relu else '-')
