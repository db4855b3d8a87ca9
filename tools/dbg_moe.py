import importlib, sys, torch
sys.path.insert(0, '.')
qt = importlib.import_module('qa-tiger_amd'); oracle = importlib.import_module('oracle.qatiger_oracle'); ops = qt.ops
torch.manual_seed(0)
B, T, D, K, k = 8, 60, 768, 8, 8
for vis in (False, True):
    moe = qt.TempMoE(D, 8, topK=k, n_experts=K, vis_branch=vis).cuda().eval()
    for p in moe.parameters():
        if p.dim() == 1: p.data.normal_(0, 0.05)
    pre = 'm'
    P = {f'{pre}.{n}': p.detach().cpu().double() for n, p in moe.named_parameters()}
    qst = torch.randn(B, D).cuda().requires_grad_(True); data = torch.randn(B, T, D).cuda().requires_grad_(True)
    subs = [torch.randn(B, T, D).cuda().requires_grad_(True) for _ in range(2)] if vis else None
    outs = moe(qst, data, subs); outs = outs if vis else (outs,)
    gs = [torch.randn_like(o) for o in outs]
    sum((o * g).sum() for o, g in zip(outs, gs)).backward()
    leaves = {n: t.clone().requires_grad_(True) for n, t in P.items()}
    q64, d64 = qst.detach().cpu().double().requires_grad_(True), data.detach().cpu().double().requires_grad_(True)
    s64 = [s.detach().cpu().double().requires_grad_(True) for s in subs] if vis else None
    ro = oracle.temp_moe(leaves, pre, q64, d64, s64, K, k, oracle._Ctx()); ro = ro if vis else (ro,)
    sum((o * g.cpu().double()).sum() for o, g in zip(ro, gs)).backward()
    print('vis', vis, 'fwd err', max(float((o.detach().cpu().double() - r).abs().max() / r.abs().max()) for o, r in zip(outs, ro)))
    rows = []
    for n, p in moe.named_parameters():
        r = leaves[f'{pre}.{n}'].grad; h = p.grad.cpu().double()
        rows.append((float((h - r).norm() / r.norm().clamp_min(1e-30)), n))
    for nm, a, r in [('qst', qst, q64), ('data', data, d64)] + ([('sub0', subs[0], s64[0]), ('sub1', subs[1], s64[1])] if vis else []):
        rows.append((float((a.grad.cpu().double() - r.grad).norm() / r.grad.norm()), 'INPUT ' + nm))
    rows.sort(reverse=True)
    for r in rows[:8]: print('   %.2e  %s' % r)
