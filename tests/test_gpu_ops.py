"""Op-level parity of every C-ABI kernel against fp64 torch math on the same
seeded inputs (GPU box only).  fp32 kernels: tolerance 2e-5 relative to the
output scale unless stated (summation-order noise)."""
import importlib
import math

import numpy as np
import pytest
import torch

from conftest import load_golden

pytestmark = pytest.mark.gpu
DEV = 'cuda'


def rel(a, b):
    a, b = a.double(), b.double()
    return float((a - b).abs().max() / b.abs().max().clamp_min(1e-30))


@pytest.fixture(scope='module')
def ops(qt):
    return qt.ops


def rnd(*shape, seed=0, scale=1.0):
    g = torch.Generator(device='cpu').manual_seed(seed)
    return (torch.randn(*shape, generator=g) * scale).to(DEV)


# --------------------------------------------------------------------------- gemm
GEMM_SHAPES = [(256, 768, 768), (128, 128, 16), (1000, 384, 128), (40, 42, 512), (560, 512, 1024),
               (37, 24, 42), (300, 130, 20), (2048, 2304, 768), (64, 8, 768), (3, 5, 7)]


@pytest.mark.parametrize('M,N,K', GEMM_SHAPES)
@pytest.mark.parametrize('ta,tb', [(False, True), (False, False), (True, False), (True, True)])
def test_gemm_layouts(ops, M, N, K, ta, tb):
    a = rnd(*((K, M) if ta else (M, K)), seed=1)
    b = rnd(*((N, K) if tb else (K, N)), seed=2)
    ref = (a.double().t() if ta else a.double()) @ (b.double().t() if tb else b.double())
    for split in (1, 0, 3):
        out = ops.gemm(a, b, trans_a=ta, trans_b=tb, split_k=split)
        assert out.shape == (M, N)
        assert rel(out, ref) < 2e-5, (split,)


@pytest.mark.parametrize('mode,tol', [('f16x3', 2e-6), ('bf16x3', 6e-5)])
@pytest.mark.parametrize('M,N,K', [(256, 768, 768), (1000, 384, 128), (2048, 2304, 768), (300, 200, 100), (3840, 768, 1024), (129, 130, 36)])
@pytest.mark.parametrize('ta,tb', [(False, True), (False, False), (True, False), (True, True)])
def test_gemm_split_precision(ops, mode, tol, M, N, K, ta, tb):
    """Three-MFMA split products: fp16 parts stay at fp32-level accuracy, bf16 parts at ~1e-5."""
    a = rnd(*((K, M) if ta else (M, K)), seed=1)
    b = rnd(*((N, K) if tb else (K, N)), seed=2, scale=0.05)        # weight-like magnitudes (fp16 lo part goes subnormal)
    ref = (a.double().t() if ta else a.double()) @ (b.double().t() if tb else b.double())
    for split in (1, 0):
        out = ops.gemm(a, b, trans_a=ta, trans_b=tb, split_k=split, mode=mode)
        assert rel(out, ref) < tol, (split, rel(out, ref))
    bias = rnd(N, seed=3)
    out = ops.gemm(a, b, trans_a=ta, trans_b=tb, bias=bias, relu=True, mode=mode)
    assert rel(out, (ref + bias.double()).clamp_min(0)) < tol


@pytest.mark.parametrize('M,N,K', [(256, 768, 768), (32, 768, 768), (768, 768, 32), (129, 130, 36), (300, 200, 100), (64, 42, 768), (17, 33, 8),
                                   (512, 768, 3072), (1000, 384, 4100)])
@pytest.mark.parametrize('ta,tb', [(False, True), (False, False), (True, False), (True, True)])
def test_gemm_skinny_kernel(ops, M, N, K, ta, tb):
    """gemm_skinny.hip (forced with the `_sk` modes; `_nosk` = the tiled kernels of gemm_split.hip): one 64x64 tile per workgroup, K split
    over its four waves, partial tiles summed in LDS in a fixed order.  Same arithmetic and epilogue contract as qt_gemm_x3: fp32-level
    products, ragged M / N / K, every epilogue, per-tensor operand scales, the fused bias gradient, max|C| — and bit-identical results from
    run to run (no atomics unless a grid-level K split is asked for)."""
    a = rnd(*((K, M) if ta else (M, K)), seed=1)
    b = rnd(*((N, K) if tb else (K, N)), seed=2, scale=0.05)
    ad, bd = (a.double().t() if ta else a.double()), (b.double().t() if tb else b.double())
    ref = ad @ bd
    kw = dict(trans_a=ta, trans_b=tb)
    out = ops.gemm(a, b, mode='f16x3_sk', **kw)
    assert rel(out, ref) < 2e-6
    assert torch.equal(out, ops.gemm(a, b, mode='f16x3_sk', **kw))                      # deterministic
    assert rel(ops.gemm(a, b, mode='bf16x3_sk', **kw), ref) < 6e-5
    for split in (2, 5):                                                                 # grid-level K split (atomics)
        assert rel(ops.gemm(a, b, mode='f16x3_sk', split_k=split, **kw), ref) < 2e-6
    bias, add, msk, c0 = rnd(N, seed=3), rnd(M, N, seed=4), rnd(M, N, seed=5), rnd(M, N, seed=6)
    assert rel(ops.gemm(a, b, mode='f16x3_sk', bias=bias, add=add, relu=True, **kw), (ref + bias.double() + add.double()).clamp_min(0)) < 2e-6
    assert rel(ops.gemm(a, b, mode='f16x3_sk', mask_src=msk, mask_scale=1.25, **kw), ref * (msk > 0) * 1.25) < 2e-6
    c = c0.clone()
    ops.gemm(a, b, mode='f16x3_sk', out=c, accumulate=True, alpha=0.5, bias=bias, **kw)
    assert rel(c, c0.double() + 0.5 * ref + bias.double()) < 2e-6
    c = c0.clone()
    ops.gemm(a, b, mode='f16x3_sk', out=c, accumulate=True, split_k=3, **kw)
    assert rel(c, c0.double() + ref) < 2e-6
    # dropout: the same mask as the tiled kernel draws for the same (seed, element index)
    d_sk = ops.gemm(a, b, mode='f16x3_sk', relu=True, drop_p=0.1, seed=99, **kw)
    d_t = ops.gemm(a, b, mode='f16x3_nosk', relu=True, drop_p=0.1, seed=99, **kw)
    big = ref.abs() > 1e-3 * ref.abs().max()                                            # (away from the ReLU threshold)
    assert torch.equal((d_sk != 0)[big.to(DEV)], (d_t != 0)[big.to(DEV)]) and rel(d_sk, d_t) < 4e-6
    # operand scales: gradients of 1e-9 against weights of 1e-4 keep fp32-level products
    a9, b4 = a * 1e-9, b * 2e-3
    am, bm = ops.amax_of(a9, keep=True), ops.amax_of(b4, keep=True)
    cam = torch.zeros(ops.AMAX_SLOTS, device=DEV)
    out = ops.gemm(a9, b4, mode='f16x3_sk', a_amax=am, b_amax=bm, c_amax=cam, **kw)
    assert rel(out, ref * 2e-12) < 2e-6
    assert float(cam.max()) == float(out.abs().max())
    if ta:                                                                              # fused bias gradient (column sums of the K-slow A)
        db = torch.full((M,), 7.0, device=DEV)
        ops.gemm(a, b, mode='f16x3_sk', a_colsum=db, **kw)
        assert rel(db, a.double().sum(0)) < 1e-5
        db0 = rnd(M, seed=8)
        db, c = db0.clone(), c0.clone()
        ops.gemm(a, b, mode='f16x3_sk', a_colsum=db, out=c, accumulate=True, **kw)
        assert rel(db, db0.double() + a.double().sum(0)) < 1e-5 and rel(c, c0.double() + ref) < 2e-6


def test_gemm_skinny_batched_and_views(ops):
    """Head-batched problems (strides between problems) and operands / outputs that are column blocks of wider tensors."""
    nb, M, dh, D = 8, 200, 96, 768
    q, w = rnd(M, nb * dh, seed=1), rnd(nb * dh, D, seed=2, scale=0.05)
    u = torch.zeros(M, nb * D, device=DEV)
    ops.gemm(q[:, :dh], w[:dh], trans_b=False, out=u[:, :D], batch=(nb, dh, dh * D, D), mode='f16x3_sk')     # u_h = q_h W_h
    for hh in range(nb):
        assert rel(u[:, hh * D:(hh + 1) * D], q[:, hh * dh:(hh + 1) * dh].double() @ w[hh * dh:(hh + 1) * dh].double()) < 2e-6
    dq = torch.zeros(M, nb * dh, device=DEV)
    ops.gemm(u[:, :D], w[:dh], trans_b=True, out=dq[:, :dh], batch=(nb, D, dh * D, dh), mode='f16x3_sk')     # dq_h = u_h W_h^T
    for hh in range(nb):
        assert rel(dq[:, hh * dh:(hh + 1) * dh], u[:, hh * D:(hh + 1) * D].double() @ w[hh * dh:(hh + 1) * dh].double().t()) < 2e-6
    dw = torch.zeros(nb * dh, D, device=DEV)
    ops.gemm(q[:, :dh], u[:, :D], trans_a=True, trans_b=False, out=dw[:dh], batch=(nb, dh, D, dh * D), mode='f16x3_sk')
    for hh in range(nb):
        assert rel(dw[hh * dh:(hh + 1) * dh], q[:, hh * dh:(hh + 1) * dh].double().t() @ u[:, hh * D:(hh + 1) * D].double()) < 2e-6
    wide = torch.zeros(M, D + 50, device=DEV)
    ops.gemm(q, w, trans_b=False, out=wide[:, 10:10 + D], mode='f16x3_sk')              # unaligned output columns
    assert rel(wide[:, 10:10 + D], q.double() @ w.double()) < 2e-6
    assert float(wide[:, :10].abs().max()) == 0 and float(wide[:, 10 + D:].abs().max()) == 0


@pytest.mark.parametrize('M,N,K', [(768, 768, 30720), (2304, 512, 70000), (130, 200, 1000), (384, 768, 960)])
def test_gemm_fused_bias_gradient(ops, M, N, K):
    """a_colsum: the column sums of the K-slow A operand (= bias gradient) computed inside the dW GEMM."""
    g = rnd(K, M, seed=6, scale=1e-3)
    x = rnd(K, N, seed=7)
    for mode in ('bf16x3', 'fp32'):
        db = torch.full((M,), 7.0, device=DEV)
        dw = ops.gemm(g, x, trans_a=True, trans_b=False, mode=mode, a_colsum=db)
        assert rel(dw, g.double().t() @ x.double()) < 6e-5
        assert rel(db, g.double().sum(0)) < 1e-5


def test_gemm_bf16x3_keeps_gradient_range(ops):
    """Backward operands are gradients of magnitude ~1e-8: bf16 parts keep them, fp16 parts would flush."""
    a = rnd(512, 256, seed=4, scale=1e-8)
    b = rnd(384, 256, seed=5)
    ref = a.double() @ b.double().t()
    assert rel(ops.gemm(a, b, mode='bf16x3'), ref) < 6e-5
    assert rel(ops.gemm(a, b, mode='f16x3'), ref) > 1e-3


def test_gemm_strided_views_and_epilogues(ops):
    M, N, K = 300, 200, 96
    big_a, big_b = rnd(M, K + 32, seed=3), rnd(N, K + 16, seed=4)
    a, b = big_a[:, 8:8 + K], big_b[:, 4:4 + K]            # unaligned base pointers -> scalar-load build
    bias, add = rnd(N, seed=5), rnd(M, N, seed=6)
    ref = a.double() @ b.double().t() + bias.double() + add.double()
    out = ops.gemm(a, b, bias=bias, add=add)
    assert rel(out, ref) < 2e-5
    out = ops.gemm(a, b, bias=bias, add=add, relu=True)
    assert rel(out, ref.clamp_min(0)) < 2e-5
    msk = rnd(M, N, seed=7)
    out = ops.gemm(a, b, mask_src=msk, mask_scale=1.25)
    assert rel(out, (a.double() @ b.double().t()) * (msk > 0) * 1.25) < 2e-5
    c0 = rnd(M, N, seed=8)
    c = c0.clone()
    ops.gemm(a, b, out=c, accumulate=True, alpha=0.5)
    assert rel(c, c0.double() + 0.5 * (a.double() @ b.double().t())) < 2e-5
    c = c0.clone()
    ops.gemm(a, b, out=c, accumulate=True, split_k=4, bias=bias)
    assert rel(c, c0.double() + a.double() @ b.double().t() + bias.double()) < 2e-5
    # output written into a column slice of a wider buffer
    wide = torch.zeros(M, N + 50, device=DEV)
    ops.gemm(a, b, out=wide[:, 10:10 + N])
    assert rel(wide[:, 10:10 + N], a.double() @ b.double().t()) < 2e-5
    assert float(wide[:, :10].abs().max()) == 0 and float(wide[:, 10 + N:].abs().max()) == 0


def test_gemm_dropout_epilogue_and_mask_grad(ops):
    M, N, K, p = 512, 384, 64, 0.1
    a, b = rnd(M, K, seed=9), rnd(N, K, seed=10)
    base = ops.gemm(a, b)
    d1 = ops.gemm(a, b, drop_p=p, seed=1234)
    d2 = ops.gemm(a, b, drop_p=p, seed=1234)
    d3 = ops.gemm(a, b, drop_p=p, seed=1235)
    assert torch.equal(d1, d2) and not torch.equal(d1, d3)
    kept = d1 != 0
    rate = float(kept.float().mean())
    assert abs(rate - (1 - p)) < 0.01
    assert rel(d1[kept], base[kept] / (1 - p)) < 1e-6
    g = rnd(M, N, seed=11)
    mg = ops.mask_grad(g, None, p, 1234)                   # regenerates the same mask from the seed
    assert torch.equal(mg != 0, kept)
    assert rel(mg[kept], g[kept] / (1 - p)) < 1e-6
    y = ops.gemm(a, b, relu=True, drop_p=p, seed=77)
    mg = ops.mask_grad(g, y, p, 77)
    assert rel(mg, g * (y > 0) / (1 - p)) < 1e-6


def test_attention_mfma_forward_with_unaligned_output_rows(ops, qt):
    """The MFMA forward stores four consecutive columns per lane as one float4 when the rows of `o` are 16-byte aligned; through the C ABI
    a caller may hand it any float pointer / leading dimension: the dword-store fallback must write the same values (bit for bit)."""
    import ctypes as C
    lib = qt._lib.load()
    n, H, Lq, Lk, E = 48, 8, 14, 14, 256
    dh = E // H
    qkv = rnd(n * Lq, 3 * E, seed=71)
    o_al = torch.empty(n * Lq, E, device=DEV)
    buf = torch.zeros(n * Lq * (E + 1) + 1, device=DEV)                   # rows at a pitch of E + 1 floats, starting one float in
    o_un = buf[1:].view(n * Lq, E + 1)[:, :E]
    f4 = 4
    for o, ldo in ((o_al, E), (o_un, E + 1)):
        rc = lib.qt_attn_fwd(None, C.c_void_p(qkv.data_ptr()), 3 * E, Lq, 1, C.c_void_p(qkv.data_ptr() + E * f4), 3 * E,
                             C.c_void_p(qkv.data_ptr() + 2 * E * f4), 3 * E, Lk, C.c_void_p(o.data_ptr()), ldo, Lq, 1,
                             n, n, 0, H, Lq, Lk, dh, 1.0 / dh ** 0.5, 0.0, C.c_ulonglong(0), None)
        assert rc == 0, lib.qt_last_error()
    torch.cuda.synchronize()
    assert torch.equal(o_al, o_un.contiguous())
    q, k, v = (t.reshape(n, Lq, H, dh).transpose(1, 2).double().cpu() for t in qkv.split(E, dim=1))
    ref = (torch.softmax(q @ k.transpose(-1, -2) / dh ** 0.5, -1) @ v).transpose(1, 2).reshape(n * Lq, E)
    assert rel(o_al.cpu(), ref) < 2e-6
    assert float(buf[0]) == 0.0 and float(buf.view(-1)[1:].view(n * Lq, E + 1)[:, E].abs().max()) == 0.0      # nothing outside the rows


@pytest.mark.parametrize('Lq,Lk,n_items,n_kv,shift', [(14, 14, 1024, 1024, 0), (60, 60, 160, 160, 80), (60, 77, 160, 80, 0)])
def test_attention_writes_split_image_for_its_out_projection(ops, Lq, Lk, n_items, n_kv, shift):
    """`attention(..., for_linear=(n_out, k_in))`: when the out-projection that consumes the result runs on the pre-split GEMM, the MFMA
    attention kernel writes the split image itself (qt_attn_fwd_img).  The image is bit-identical to qt_split_f16 of the fp32 output,
    Linear finds it without a split pass, and forward / backward of attention + Linear match the fp32-rows route."""
    E, H = 768, 8
    packed = Lq == Lk and n_kv == n_items
    spec = ops.AttnSpec(n_items, n_kv, shift, H, Lq, Lk, E, 0, E if packed else 0, 2 * E if packed else E)
    qkv = rnd(n_items * Lq, 3 * E if packed else E, seed=31).requires_grad_(True)
    kv = None if packed else rnd(n_kv * Lk, 2 * E, seed=32).requires_grad_(True)
    w, b = rnd(E, E, seed=33, scale=0.05).requires_grad_(True), rnd(E, seed=34).requires_grad_(True)
    assert n_items * Lq >= ops.plane_dw_min_rows
    o_ref = ops.attention(qkv, kv, spec, 0.1, 77)
    o_img = ops.attention(qkv, kv, spec, 0.1, 77, for_linear=(E, E))
    assert getattr(o_img, '_qt_is_image', False) and not getattr(o_ref, '_qt_is_image', False)
    assert torch.equal(o_img.detach().view(torch.int32), ops.split_image(o_ref.detach()).data.view(torch.int32))
    gy = rnd(n_items * Lq, E, seed=35)
    outs = []
    for o in (o_ref, o_img):
        for t in (qkv, kv, w, b):
            if t is not None:
                t.grad = None
        y = ops.linear(o, w, b)
        y.backward(gy)
        outs.append((y.detach().clone(), qkv.grad.clone(), None if kv is None else kv.grad.clone(), w.grad.clone(), b.grad.clone()))
    for a_, b_ in zip(*outs):
        if a_ is not None:
            assert rel(a_, b_) < 2e-6
    with pytest.raises(Exception, match='split image'):                      # the tensor must never be read as fp32 rows: a layer off the pre-split path refuses it
        ops.linear(o_img, w[:64].detach(), None)


def test_linear_emits_the_next_layers_operand_image(ops):
    """`ops.linear(..., emit_image=True)`: the GEMM epilogue writes the split image of its result next to the fp32 rows
    (qt_gemm_ps_img); the image is bit-identical to qt_split_f16 of the result, survives `ops.reshape` of the leading dimensions, and
    the consuming Linear gives the same bits as on a tensor without the image (relu + dropout epilogue included)."""
    rows, k_in, n_mid, n_out = 16384, 1024, 768, 384
    x = rnd(32, rows // 32, k_in, seed=41)
    w1, b1 = rnd(n_mid, k_in, seed=42, scale=0.03), rnd(n_mid, seed=43)
    w2, b2 = rnd(n_out, n_mid, seed=44, scale=0.05), rnd(n_out, seed=45)
    for kw in (dict(), dict(relu=True, drop_p=0.1, seed=7)):
        ops.new_pass()
        y = ops.linear(x, w1, b1, emit_image=True, **kw)
        plain = ops.linear(x, w1, b1, **kw)
        assert torch.equal(y, plain) and getattr(y, '_qt_img', None) is not None and getattr(plain, '_qt_img', None) is None
        assert torch.equal(y._qt_img[1].data.view(torch.int32), ops.split_image(plain.reshape(rows, n_mid)).data.view(torch.int32))
        y2 = ops.reshape(y, (rows, n_mid))
        assert getattr(y2, '_qt_img', None) is not None and y2._qt_img[1] is y._qt_img[1]
        timer = ops.timer = ops.KernelTimer()
        z = ops.linear(y2, w2, b2)
        ops.timer = None
        assert not [r for r in timer.records if r[0] == 'split' and rows in r[2]], 'the consumer ran a split pass although its operand image was there'
        assert torch.equal(z, ops.linear(plain.reshape(rows, n_mid), w2, b2))


@pytest.mark.parametrize('rows,E,H', [(30720, 768, 8), (257, 1024, 8), (64, 96, 3)])
def test_head_bias_fn(ops, rows, E, H):
    """out = o + asum[:, head(e)] * b_v[e] (qt_headbias_fwd / qt_headbias_bwd): the value bias of a folded attention, forward and all three
    gradients against fp64 torch math."""
    o = rnd(rows, E, seed=51).requires_grad_(True)
    asum = (1.0 + 0.1 * rnd(rows, H, seed=52)).requires_grad_(True)
    bv = rnd(E, seed=53).requires_grad_(True)
    out = ops.HeadBiasFn.apply(o, asum, bv, H)
    g = rnd(rows, E, seed=54)
    out.backward(g)
    od, ad, bd = o.detach().double().requires_grad_(True), asum.detach().double().requires_grad_(True), bv.detach().double().requires_grad_(True)
    ref = od + ad.repeat_interleave(E // H, dim=1) * bd
    ref.backward(g.double())
    assert rel(out.detach(), ref.detach()) < 1e-6
    assert rel(o.grad, od.grad) < 1e-6 and rel(asum.grad, ad.grad) < 1e-5 and rel(bv.grad, bd.grad) < 1e-5


def test_linear_fn_grads(ops):
    x = rnd(5, 7, 96, seed=12).requires_grad_(True)
    w = rnd(130, 96, seed=13, scale=0.1).requires_grad_(True)
    b = rnd(130, seed=14).requires_grad_(True)
    y = ops.linear(x, w, b, relu=True)
    gy = rnd(5, 7, 130, seed=15)
    y.backward(gy)
    xr, wr, br = [t.detach().double().requires_grad_(True) for t in (x, w, b)]
    yr = torch.relu(xr @ wr.t() + br)
    yr.backward(gy.double())
    assert rel(y, yr) < 2e-5
    assert rel(x.grad, xr.grad) < 2e-5 and rel(w.grad, wr.grad) < 2e-5 and rel(b.grad, br.grad) < 2e-5


# --------------------------------------------------------------------------- attention
def ref_attention(q, k, v, H):
    """q [N,Lq,E], k/v [N,Lk,E] (already mapped) -> [N,Lq,E], fp64."""
    N, Lq, E = q.shape
    Lk, dh = k.shape[1], E // H
    qh = q.reshape(N, Lq, H, dh).transpose(1, 2) / math.sqrt(dh)
    kh = k.reshape(N, Lk, H, dh).transpose(1, 2)
    vh = v.reshape(N, Lk, H, dh).transpose(1, 2)
    a = torch.softmax(qh @ kh.transpose(-1, -2), -1)
    return (a @ vh).transpose(1, 2).reshape(N, Lq, E)


@pytest.mark.parametrize('N,nkv,shift,Lq,Lk,E', [
    (6, 6, 0, 10, 10, 64),      # self-attention
    (6, 6, 3, 10, 10, 64),      # audio<->video cross on the stacked batch
    (8, 8, 0, 14, 14, 96),      # patch self-attention (one-wave workgroups)
    (4, 4, 0, 60, 60, 768),     # AVQ self-attention at d_h = 96
    (3, 3, 1, 64, 64, 1024),    # the largest resident geometry at d_h = 128
    (2, 2, 0, 60, 60, 1024),
])
def test_attention_packed(ops, N, nkv, shift, Lq, Lk, E):
    H = 8
    qkv = rnd(N * Lq, 3 * E, seed=20).requires_grad_(True)
    spec = ops.AttnSpec(N, nkv, shift, H, Lq, Lk, E, 0, E, 2 * E)
    o = ops.attention(qkv, None, spec)
    go = rnd(N * Lq, E, seed=21)
    o.backward(go)
    x = qkv.detach().double().requires_grad_(True)
    q, k, v = [x[:, i * E:(i + 1) * E].reshape(N, Lq, E) for i in range(3)]
    idx = (torch.arange(N, device=DEV) + shift) % nkv
    ref = ref_attention(q, k[idx], v[idx], H).reshape(N * Lq, E)
    ref.backward(go.double())
    assert rel(o, ref) < 2e-5
    assert rel(qkv.grad, x.grad) < 5e-5


@pytest.mark.parametrize('N,nkv,Lq,Lk,E', [(8, 4, 10, 77, 64), (5, 5, 1, 60, 768), (7, 7, 1, 2, 64), (6, 3, 60, 77, 768),
                                          # the window where the MFMA BACKWARD would not fit 160 KiB of LDS (d_h >= 96, 81 <= Lk <= 128, Lq > 16):
                                          # forward and backward must take the same (streaming) family — csrc/attn.hip attn_path
                                          (4, 2, 60, 100, 768), (3, 3, 32, 128, 1024), (4, 4, 17, 81, 768), (2, 2, 64, 128, 1024),
                                          (3, 3, 16, 128, 1024)])
def test_attention_q_and_packed_kv(ops, N, nkv, Lq, Lk, E):
    H = 8
    qb = rnd(N * Lq, E, seed=22).requires_grad_(True)
    kvb = rnd(nkv * Lk, 2 * E, seed=23).requires_grad_(True)
    spec = ops.AttnSpec(N, nkv, 0, H, Lq, Lk, E, 0, 0, E)
    o = ops.attention(qb, kvb, spec)
    go = rnd(N * Lq, E, seed=24)
    o.backward(go)
    qr, kr = qb.detach().double().requires_grad_(True), kvb.detach().double().requires_grad_(True)
    idx = torch.arange(N, device=DEV) % nkv
    k = kr[:, :E].reshape(nkv, Lk, E)[idx]
    v = kr[:, E:].reshape(nkv, Lk, E)[idx]
    ref = ref_attention(qr.reshape(N, Lq, E), k, v, H).reshape(N * Lq, E)
    ref.backward(go.double())
    assert rel(o, ref) < 2e-5
    assert rel(qb.grad, qr.grad) < 5e-5 and rel(kvb.grad, kr.grad) < 5e-5


def test_attention_strided_queries(ops):
    """PatchSelecter cross-attention: item n, query row i lives at row i*BT + n."""
    BT, P, E, H = 12, 14, 64, 8
    qb = rnd(2 * BT, E, seed=25).requires_grad_(True)
    kvb = rnd(BT * P, 2 * E, seed=26).requires_grad_(True)
    spec = ops.AttnSpec(BT, BT, 0, H, 2, P, E, 0, 0, E, q_is=1, q_rs=BT, o_rows=2 * BT)
    o = ops.attention(qb, kvb, spec)
    go = rnd(2 * BT, E, seed=27)
    o.backward(go)
    qr, kr = qb.detach().double().requires_grad_(True), kvb.detach().double().requires_grad_(True)
    q = qr.reshape(2, BT, E).transpose(0, 1)               # [BT,2,E]
    ref = ref_attention(q, kr[:, :E].reshape(BT, P, E), kr[:, E:].reshape(BT, P, E), H)
    ref = ref.transpose(0, 1).reshape(2 * BT, E)
    ref.backward(go.double())
    assert rel(o, ref) < 2e-5
    assert rel(qb.grad, qr.grad) < 5e-5 and rel(kvb.grad, kr.grad) < 5e-5


def test_attention_dropout_statistics(ops):
    N, L, E, H, p = 16, 32, 64, 8, 0.1
    qkv = rnd(N * L, 3 * E, seed=28)
    v_ones = qkv.clone()
    v_ones[:, 2 * E:] = 1.0                                # V = 1  =>  O = row-sum of dropped probabilities
    spec = ops.AttnSpec(N, N, 0, H, L, L, E, 0, E, 2 * E)
    o = ops.attention(v_ones, None, spec, p, 99)
    assert abs(float(o.mean()) - 1.0) < 0.02               # E[sum_j keep_j p_j/(1-p)] = 1
    assert float(o.std()) > 1e-3
    assert torch.equal(o, ops.attention(v_ones, None, spec, p, 99))
    assert rel(ops.attention(v_ones, None, spec, 0.0, 0), torch.ones_like(o)) < 1e-5
    # backward regenerates the same mask: finite-difference-free check via linearity in V
    x = qkv.clone().requires_grad_(True)
    out = ops.attention(x, None, spec, p, 5)
    go = rnd(N * L, E, seed=29)
    out.backward(go)
    xv = qkv.clone()
    xv[:, 2 * E:] += 1e-2 * rnd(N * L, E, seed=30)
    dout = ops.attention(xv, None, spec, p, 5) - out.detach()
    lhs = float((dout * go).sum())
    rhs = float((x.grad[:, 2 * E:] * (xv - qkv)[:, 2 * E:]).sum())
    assert abs(lhs - rhs) < 1e-3 * max(1.0, abs(lhs))


@pytest.mark.parametrize('N,nkv,shift,Lq,Lk,E', [
    (3, 3, 0, 300, 300, 1024),   # BASELINE config 4: T=300, d_h=128 self-attention
    (4, 4, 2, 300, 300, 512),    # cross on the stacked batch, d_h=64
    (2, 2, 0, 130, 130, 768),    # just past the resident limits, ragged last chunk/tile
    (2, 2, 0, 100, 100, 256),    # d_h=32
])
def test_attention_streaming_packed(ops, N, nkv, shift, Lq, Lk, E):
    """Key-chunked online-softmax forward and the two-kernel backward (long sequences)."""
    H = 8
    qkv = rnd(N * Lq, 3 * E, seed=31).requires_grad_(True)
    spec = ops.AttnSpec(N, nkv, shift, H, Lq, Lk, E, 0, E, 2 * E)
    o = ops.attention(qkv, None, spec)
    go = rnd(N * Lq, E, seed=32)
    o.backward(go)
    x = qkv.detach().double().requires_grad_(True)
    q, k, v = [x[:, i * E:(i + 1) * E].reshape(N, Lq, E) for i in range(3)]
    idx = (torch.arange(N, device=DEV) + shift) % nkv
    ref = ref_attention(q, k[idx], v[idx], H).reshape(N * Lq, E)
    ref.backward(go.double())
    assert rel(o, ref) < 2e-5
    assert rel(qkv.grad, x.grad) < 5e-5


@pytest.mark.parametrize('N,nkv,Lq,Lk,E', [(6, 3, 300, 77, 1024), (5, 5, 1, 300, 1024), (4, 2, 70, 200, 768)])
def test_attention_streaming_q_and_kv(ops, N, nkv, Lq, Lk, E):
    H = 8
    qb = rnd(N * Lq, E, seed=33).requires_grad_(True)
    kvb = rnd(nkv * Lk, 2 * E, seed=34).requires_grad_(True)
    spec = ops.AttnSpec(N, nkv, 0, H, Lq, Lk, E, 0, 0, E)
    o = ops.attention(qb, kvb, spec)
    go = rnd(N * Lq, E, seed=35)
    o.backward(go)
    qr, kr = qb.detach().double().requires_grad_(True), kvb.detach().double().requires_grad_(True)
    idx = torch.arange(N, device=DEV) % nkv
    ref = ref_attention(qr.reshape(N, Lq, E), kr[:, :E].reshape(nkv, Lk, E)[idx], kr[:, E:].reshape(nkv, Lk, E)[idx], H).reshape(N * Lq, E)
    ref.backward(go.double())
    assert rel(o, ref) < 2e-5
    assert rel(qb.grad, qr.grad) < 5e-5 and rel(kvb.grad, kr.grad) < 5e-5


def test_attention_streaming_dropout_consistency(ops):
    """Forward and both backward kernels must regenerate the same mask: <dO, dOut/dV·δV> = <dV, δV>."""
    N, L, E, H, p = 2, 200, 512, 8, 0.1
    qkv = rnd(N * L, 3 * E, seed=36)
    spec = ops.AttnSpec(N, N, 0, H, L, L, E, 0, E, 2 * E)
    x = qkv.clone().requires_grad_(True)
    out = ops.attention(x, None, spec, p, 7)
    assert torch.equal(out, ops.attention(qkv, None, spec, p, 7))
    go = rnd(N * L, E, seed=37)
    out.backward(go)
    for cols in (slice(2 * E, 3 * E), slice(0, E), slice(E, 2 * E)):            # V (linear), then Q and K (first order)
        xv = qkv.clone()
        dlt = 1e-3 * rnd(N * L, E, seed=38)
        xv[:, cols] += dlt
        lhs = float(((ops.attention(xv, None, spec, p, 7) - out.detach()) * go).sum())
        rhs = float((x.grad[:, cols] * dlt).sum())
        assert abs(lhs - rhs) < 2e-2 * max(1e-3, abs(lhs)) + 1e-4


@pytest.mark.parametrize('n_items,Lq,Lk,E,u_is,u_rs', [(24, 2, 14, 768, 1, 24), (16, 1, 60, 768, 1, 0), (6, 1, 300, 1024, 1, 0),
                                                      (10, 2, 3, 64, 1, 10), (7, 1, 7, 512, 1, 0),
                                                      # the MFMA forward's shape class (<= 16 (query, head) vectors, <= 16 keys, D 768 / 1024):
                                                      (12, 2, 16, 1024, 1, 12), (9, 2, 5, 768, 1, 9), (6, 1, 9, 768, 1, 0),
                                                      # >= 2048 query rows: the head-slab GEMMs (u = q_h W_k,h and dC_h = dO_h W_v,h) run on the pre-split kernel
                                                      (4200, 2, 2, 768, 1, 4200)])
@pytest.mark.parametrize('p', [0.0, 0.1])
def test_folded_attention_matches_projected_attention(ops, n_items, Lq, Lk, E, u_is, u_rs, p):
    """K/V-folded few-query attention == nn.MultiheadAttention math with explicit K/V projections (fp64),
    forward and every gradient (q, raw rows, in-projection weight and bias)."""
    H, dh = 8, E // 8
    torch.manual_seed(0)
    mha = torch.nn.MultiheadAttention(E, H).to(DEV)
    with torch.no_grad():
        mha.in_proj_bias.normal_(0, 0.05)
    rows = n_items * Lq
    q = rnd(rows, E, seed=41).requires_grad_(True)
    x = rnd(n_items * Lk, E, seed=42).requires_grad_(True)
    o = ops.folded_attention(q, x, mha, n_items, Lq, Lk, u_is, u_rs, p, 11)
    if p > 0:       # statistical check only: deterministic for a seed, mean preserved
        o2 = ops.folded_attention(q, x, mha, n_items, Lq, Lk, u_is, u_rs, p, 11)                  # same seed -> same mask
        assert rel(o2, o) < 1e-5                                  # (split-K atomics: last-bit differences only)
        assert rel(ops.folded_attention(q, x, mha, n_items, Lq, Lk, u_is, u_rs, p, 12), o) > 1e-3
        o.backward(rnd(rows, E, seed=43))
        assert torch.isfinite(q.grad).all() and torch.isfinite(x.grad).all() and torch.isfinite(mha.in_proj_weight.grad).all()
        return
    go = rnd(rows, E, seed=43)
    o.backward(go)
    W, b = mha.in_proj_weight.detach().double().requires_grad_(True), mha.in_proj_bias.detach().double().requires_grad_(True)
    qr, xr = q.detach().double().requires_grad_(True), x.detach().double().requires_grad_(True)
    k = (xr @ W[E:2 * E].t() + b[E:2 * E]).reshape(n_items, Lk, E)
    v = (xr @ W[2 * E:].t() + b[2 * E:]).reshape(n_items, Lk, E)
    idx = torch.arange(n_items, device=DEV)
    qq = torch.stack([qr[(idx * u_is + i * u_rs)] for i in range(Lq)], 1)     # [n, Lq, E]
    ref = ref_attention(qq, k, v, H)
    out_ref = torch.zeros(rows, E, dtype=torch.double, device=DEV)
    for i in range(Lq):
        out_ref = out_ref.index_add(0, idx * u_is + i * u_rs, ref[:, i])
    out_ref.backward(go.double())
    assert rel(o, out_ref) < 5e-6
    assert rel(q.grad, qr.grad) < 1e-4 and rel(x.grad, xr.grad) < 1e-4
    gW, gb = mha.in_proj_weight.grad, mha.in_proj_bias.grad
    assert rel(gW[E:], W.grad[E:]) < 1e-4 and rel(gb[2 * E:], b.grad[2 * E:]) < 1e-4
    assert float(gb[:2 * E].abs().max()) == 0 and float(gW[:E].abs().max()) == 0   # q block not used here; key bias has zero true gradient


def test_folded_attention_mfma_forward_and_backward_share_the_dropout_mask(ops):
    """PatchSelecter shape: forward and backward run on their f32 MFMA kernels (qt_xattn_fwd_mfma_kernel / qt_xattn_bwd_mfma_kernel);
    both must draw the same attention-probability dropout mask from (seed, row, key): first-order check
    <dO, out(x + d) - out(x)> = <dx, d>."""
    n_items, Lq, Lk, E, H, p = 20, 2, 14, 768, 8, 0.1
    torch.manual_seed(1)
    mha = torch.nn.MultiheadAttention(E, H).to(DEV)
    q = rnd(n_items * Lq, E, seed=51)
    x0 = rnd(n_items * Lk, E, seed=52)
    x = x0.clone().requires_grad_(True)
    out = ops.folded_attention(q, x, mha, n_items, Lq, Lk, 1, n_items, p, 23)
    go = rnd(n_items * Lq, E, seed=53)
    out.backward(go)
    dlt = 1e-3 * rnd(n_items * Lk, E, seed=54)
    with torch.no_grad():
        out2 = ops.folded_attention(q, x0 + dlt, mha, n_items, Lq, Lk, 1, n_items, p, 23)
    lhs = float(((out2 - out.detach()) * go).sum())
    rhs = float((x.grad * dlt).sum())
    assert abs(lhs - rhs) < 2e-2 * max(1e-3, abs(lhs)) + 1e-4, (lhs, rhs)


def test_attention_unsupported_shape_fails_loudly(ops, qt):
    E, H, L = 192, 8, 300                                  # d_h = 24: neither MFMA path applies and the generic kernel needs > 160 KiB
    qkv = torch.zeros(L, 3 * E, device=DEV)
    with pytest.raises(qt.QTError, match='LDS'):
        ops.attention(qkv, None, ops.AttnSpec(1, 1, 0, H, L, L, E, 0, E, 2 * E))


# --------------------------------------------------------------------------- add + LayerNorm
@pytest.mark.parametrize('rows,D', [(37, 64), (300, 768), (5, 1024), (2048, 512)])
def test_add_layer_norm(ops, rows, D):
    xs = [rnd(rows, D, seed=40 + i).requires_grad_(True) for i in range(3)]
    gamma = (1 + 0.1 * rnd(D, seed=44)).requires_grad_(True)
    beta = rnd(D, seed=45).requires_grad_(True)
    scales = [1.0, 0.5, 2.0]
    y = ops.add_layer_norm(xs, gamma, beta, scales=scales, relu_out=True)
    gy = rnd(rows, D, seed=46)
    y.backward(gy)
    xr = [x.detach().double().requires_grad_(True) for x in xs]
    gr, br = gamma.detach().double().requires_grad_(True), beta.detach().double().requires_grad_(True)
    s = sum(sc * x for sc, x in zip(scales, xr))
    ln = torch.nn.functional.layer_norm(s, (D,), gr, br, 1e-5)
    # ReLU gate taken from the kernel's own output: an element with |y| ~ 1e-7 may round to the
    # other side of zero in fp32, which is not an error but would flip one gate of the fp64 reference
    flips = (y.detach() > 0) != (ln.detach() > 0)
    assert float(ln.detach().abs()[flips].max()) < 1e-5 if bool(flips.any()) else True
    ref = ln * (y.detach() > 0)
    ref.backward(gy.double())
    assert rel(y, ref) < 2e-5
    for x, r in zip(xs, xr):
        assert rel(x.grad, r.grad) < 1e-4
    assert rel(gamma.grad, gr.grad) < 1e-4 and rel(beta.grad, br.grad) < 1e-4


def test_add_layer_norm_strided_and_dropout(ops):
    B, S, D, p = 64, 2, 128, 0.1
    d = rnd(B, S, D, seed=47)
    m = rnd(B, D, seed=48)
    gamma, beta = 1 + 0.1 * rnd(D, seed=49), rnd(D, seed=50)
    y = ops.add_layer_norm([d[:, 0, :], d[:, 1, :], m], gamma, beta, scales=[0.5, 0.5, 1.0])
    ref = torch.nn.functional.layer_norm(d.double().mean(1) + m.double(), (D,), gamma.double(), beta.double(), 1e-5)
    assert rel(y, ref) < 2e-5
    torch.manual_seed(3)
    ops.seeds.reset()
    x0 = torch.zeros(4096, D, device=DEV)
    ones = torch.ones(4096, D, device=DEV).requires_grad_(True)
    # LN of a dropped all-ones row: kept elements share one value; the keep-rate shows in the sign pattern
    y = ops.add_layer_norm([x0, ones], torch.ones(D, device=DEV), torch.zeros(D, device=DEV), drops=[False, True], drop_p=p)
    rate = float((y > 0).float().mean())
    assert abs(rate - (1 - p)) < 0.01
    y.backward(rnd(4096, D, seed=51))
    assert ones.grad is not None and torch.isfinite(ones.grad).all()
    assert float((ones.grad[y.detach() <= 0]).abs().max()) == 0.0     # dropped elements get no gradient
    assert float((ones.grad[y.detach() > 0]).abs().min()) > 0.0


# --------------------------------------------------------------------------- TempMoE kernels
def test_moe_router_and_combine_known_answers(ops, oracle):
    """Reference vectors (tests/golden/tempmoe_kats.npz): top-k order, Gaussian
    weights and the (b*T+t) % B expert-index rule, k<K, k=K, B>T, B not dividing T."""
    g = load_golden('tempmoe_kats')
    n = len([k for k in g if k.endswith('/shape')])
    for ci in range(n):
        B, T, C, K, k = g[f'kat{ci}/shape'].tolist()
        ti = torch.from_numpy(g[f'kat{ci}/topk_inds']).to(DEV)
        tp = torch.from_numpy(g[f'kat{ci}/topk_probs']).to(DEV)
        gauss = torch.from_numpy(g[f'kat{ci}/gauss']).to(DEV)
        eo = torch.from_numpy(g[f'kat{ci}/experts_out_TBKC']).to(DEV).permute(1, 0, 2, 3).contiguous()   # [B,T,K,C]
        lib = importlib.import_module('qa-tiger_amd._lib').load()
        coef = torch.empty(B, T, K, device=DEV)
        csum = torch.empty(B, K, device=DEV)
        rc = lib.qt_moe_coef_fwd(ops._stream(), ops._ptr(ti), ops._ptr(tp), ops._ptr(gauss), B, T, K, k, ops._ptr(coef), ops._ptr(csum))
        assert rc == 0
        out = torch.einsum('btk,btkc->bc', coef.double(), eo.double())
        assert rel(out, torch.from_numpy(g[f'kat{ci}/out']).to(DEV)[:, 0]) < 1e-5
        assert rel(csum, coef.double().sum(1)) < 1e-6


@pytest.mark.parametrize('B,T,K,k', [(5, 7, 6, 3), (9, 6, 4, 4), (64, 60, 8, 8), (3, 60, 16, 5)])
def test_moe_coef_fn_matches_oracle_with_grads(ops, oracle, B, T, K, k):
    logits = rnd(B, K, seed=60).requires_grad_(True)
    raw = (rnd(B, 2 * K, seed=61) * 1.5).requires_grad_(True)
    coef, csum, ti, tp, gauss = ops.MoECoefFn.apply(logits, raw, T, K, k)
    gc, gs = rnd(B, T, K, seed=62), rnd(B, K, seed=63)
    (coef * gc).sum().add((csum * gs).sum()).backward()

    lr, rr = logits.detach().double().cpu().requires_grad_(True), raw.detach().double().cpu().requires_grad_(True)
    probs = torch.softmax(lr, -1)
    tpr, tir = torch.topk(probs, k, dim=-1)
    tpr = tpr / tpr.sum(-1, keepdim=True)
    r3 = rr.view(B, K, 2)
    pred = torch.stack([torch.tanh(r3[..., 0]) * (1 / (2 * K)), torch.sigmoid(r3[..., 1])], -1)
    gr = oracle.generate_gaussian(pred, tir, T, K)
    # coefficient table by brute force from the definition
    rows = torch.arange(B * T)
    e_idx = tir[rows % B]                                   # [BT,k]
    cref = torch.zeros(B * T, K, dtype=torch.double)
    vals = (tpr[:, :, None] * gr).permute(0, 2, 1).reshape(B * T, k)
    cref = cref.scatter_add(1, e_idx, vals).reshape(B, T, K)
    (cref * gc.double().cpu()).sum().add((cref.sum(1) * gs.double().cpu()).sum()).backward()
    assert torch.equal(ti.cpu(), tir)
    assert rel(tp.cpu(), tpr) < 1e-5 and rel(gauss.cpu(), gr) < 1e-5
    assert rel(coef.cpu(), cref) < 1e-5 and rel(csum.cpu(), cref.sum(1)) < 1e-5
    assert rel(logits.grad.cpu(), lr.grad) < 2e-4
    assert rel(raw.grad.cpu(), rr.grad) < 2e-4


def test_moe_reduce_fn(ops):
    S, B, T, K, Dh = 2, 6, 5, 4, 32
    H = torch.relu(rnd(S * B * T, K * Dh, seed=64)).requires_grad_(True)
    coef = rnd(B, T, K, seed=65).abs().requires_grad_(True)
    R = ops.MoEReduceFn.apply(H, coef, S, K)
    gR = rnd(S * B, K * Dh, seed=66)
    R.backward(gR)
    Hr, cr = H.detach().double().requires_grad_(True), coef.detach().double().requires_grad_(True)
    ref = torch.einsum('sbtkd,btk->sbkd', Hr.reshape(S, B, T, K, Dh), cr).reshape(S * B, K * Dh)
    ref.backward(gR.double())
    assert rel(R, ref) < 2e-5
    assert rel(coef.grad, cr.grad) < 2e-5
    assert rel(H.grad, Hr.grad * (H.detach() > 0)) < 2e-5  # dH carries the ReLU mask of H


# --------------------------------------------------------------------------- misc
def test_colsum_addrows(ops):
    x = rnd(1234, 300, seed=70)
    assert rel(ops.colsum(x), x.double().sum(0)) < 1e-5
    assert rel(ops.colsum(x[:, 4:204]), x[:, 4:204].double().sum(0)) < 1e-5
    a, b = rnd(10, 64, seed=71), rnd(30, 64, seed=72)
    assert rel(ops.add_rows(a, b), a.double().repeat(3, 1) + b.double()) < 1e-6


def test_cross_entropy(ops):
    logits = rnd(37, 42, seed=73).requires_grad_(True)
    label = torch.randint(0, 42, (37,), generator=torch.Generator().manual_seed(1)).to(DEV)
    loss = ops.cross_entropy(logits, label)
    loss.backward()
    lr = logits.detach().double().requires_grad_(True)
    ref = torch.nn.functional.cross_entropy(lr, label)
    ref.backward()
    assert abs(float(loss) - float(ref)) < 1e-6
    assert rel(logits.grad, lr.grad) < 1e-5


def test_fused_adam_matches_torch(ops):
    n = 1003
    p0 = rnd(n, seed=74)
    ref = p0.clone().requires_grad_(True)
    opt = torch.optim.Adam([ref], lr=1e-4, betas=(0.95, 0.999), eps=1e-8)
    p, m, v = p0.clone(), torch.zeros(n, device=DEV), torch.zeros(n, device=DEV)
    for step in range(1, 5):
        g = rnd(n, seed=80 + step)
        ref.grad = g.clone()
        opt.step()
        ops.adam_step_(p, g, m, v, step, 1e-4, (0.95, 0.999), 1e-8)
        assert float((p - ref.detach()).abs().max()) < 1e-7


@pytest.mark.parametrize('sa,sb,ta,tb', [(1e-7, 1.0, False, True), (3e-9, 2e-5, False, False), (1e-6, 1.0, True, False), (1e5, 1e-4, False, True)])
def test_gemm_f16x3_scaled_operands_are_fp32_equivalent(ops, sa, sb, ta, tb):
    """The split-in-the-consumer kernel with per-tensor power-of-two scales (qt_gemm_x3 a_amax / b_amax): gradient-like (1e-7 .. 1e-9),
    tiny-weight (1e-4) and huge (1e5) operands give the error of the exact-fp32 MFMA kernel; unscaled, the same operands lose
    their low parts to fp16 subnormals or overflow.  This is what makes the BACKWARD GEMMs fp32-equivalent."""
    M, N, K = 384, 256, 512
    g = torch.Generator().manual_seed(91)
    a = (torch.randn((K, M) if ta else (M, K), generator=g) * sa).to(DEV)
    b = (torch.randn((N, K) if tb else (K, N), generator=g) * sb).to(DEV)
    ref = (a.double().t() if ta else a.double()) @ (b.double().t() if tb else b.double())
    out = ops.gemm(a, b, trans_a=ta, trans_b=tb, mode='f16x3', a_amax=ops.amax_of(a), b_amax=ops.amax_of(b))
    e = rel(out, ref)
    e32 = rel(ops.gemm(a, b, trans_a=ta, trans_b=tb, mode='fp32'), ref)
    raw = ops.gemm(a, b, trans_a=ta, trans_b=tb, mode='f16x3')
    e_raw = rel(raw, ref) if bool(torch.isfinite(raw).all()) else float('inf')
    print(f'scales {sa:g} x {sb:g}: scaled f16x3 {e:.2e}, exact fp32 {e32:.2e}, unscaled f16x3 {e_raw:.2e}')
    assert e < 2e-6 and e < 4 * e32 + 1.5e-6       # (an operand whose maximum is already in [2^-6, 2^14] is left unscaled: >= 19 bits)
    assert e_raw > 10 * e
