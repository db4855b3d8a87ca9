"""Pre-split GEMM (csrc/gemm_plane.hip: qt_amax_f32 / qt_split_f16 / qt_gemm_ps) against an fp64 reference of the same
product (plain PyTorch on the CPU).  The kernel's claim is fp32-equivalence: its error against fp64 must be of the order
of the exact-fp32 MFMA kernel's own error, for operands of ANY magnitude once they are scaled by their max |x|."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = 'cuda'


@pytest.fixture(scope='module')
def ops(qt):
    return qt.ops


def rnd(*shape, seed, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(*shape, generator=g) * scale)


def rel(a, b):
    return float((a.double().cpu() - b.double().cpu()).abs().max() / b.double().abs().max())


def l2(a, b):
    return float((a.double().cpu() - b.double().cpu()).norm() / b.double().norm())


def test_split_image_layout_and_colsum(ops):
    x = rnd(70, 96, seed=1)
    cs = torch.empty(96, device=DEV)
    img = ops.split_image(x.to(DEV), colsum=cs)
    raw = img.data.cpu().view(torch.float16).reshape(70, 3, 2, 32)          # [row][block][hi|lo][32]
    hi, lo = raw[:, :, 0, :].reshape(70, 96).float(), raw[:, :, 1, :].reshape(70, 96).float()
    assert torch.equal(hi, x.half().float())
    assert torch.equal(lo, (x - x.half().float()).half().float())
    assert float((hi.double() + lo.double() - x.double()).abs().max()) < 2 ** -21 * float(x.abs().max())
    assert rel(cs, x.double().sum(0)) < 1e-6
    # scaled: amax lands in [2^13, 2^14)
    y = x * 1e-7
    im2 = ops.split_image(y.to(DEV), scaled=True)
    assert float(im2.amax.max()) == float(y.abs().max())
    hi2 = im2.data.cpu().view(torch.float16).reshape(70, 3, 2, 32)[:, :, 0, :].float()
    assert 2 ** 13 <= float(hi2.abs().max()) < 2 ** 14 + 8


CASES = [
    # M, N, K, a_kslow, b_kslow
    (512, 256, 256, False, False),     # forward: x [M,K] . W[N,K]^T
    (768, 384, 96, False, False),      # ragged N (384 = 1.5 tiles), short K
    (300, 256, 128, False, False),     # ragged M
    (512, 256, 256, False, True),      # dX: dY [M,K] . W[K,N]
    (600, 320, 160, False, True),
    (256, 256, 1024, True, True),      # dW: dY[K,M]^T . X[K,N]
    (768, 128, 2048, True, True),
    (320, 288, 640, True, True),       # ragged both, K-slow widths multiples of 32
    (512, 256, 256, True, False),
]


@pytest.mark.parametrize('M,N,K,aks,bks', CASES)
def test_gemm_ps_matches_fp64(ops, M, N, K, aks, bks):
    a = rnd(*((K, M) if aks else (M, K)), seed=3)
    b = rnd(*((K, N) if bks else (N, K)), seed=4)
    ia, ib = ops.split_image(a.to(DEV)), ops.split_image(b.to(DEV))
    out = ops.gemm_ps(ia, ib, a_kslow=aks, b_kslow=bks)
    assert out is not None
    ref = (a.double().t() if aks else a.double()) @ (b.double() if bks else b.double().t())
    e = rel(out, ref)
    exact = ops.gemm(a.to(DEV), b.to(DEV), trans_a=aks, trans_b=not bks, mode='fp32')
    e32 = rel(exact, ref)
    print(f'{M}x{N}x{K} aks={aks} bks={bks}: ps {e:.2e}  exact-fp32 kernel {e32:.2e}')
    assert e < 2e-6 and e < 4 * e32 + 5e-7


def test_gemm_ps_split_k_and_epilogues(ops):
    M, N, K = 512, 256, 4096
    a, b = rnd(K, M, seed=5), rnd(K, N, seed=6)
    ia, ib = ops.split_image(a.to(DEV)), ops.split_image(b.to(DEV))
    ref = a.double().t() @ b.double()
    for sk in (0, 1, 4):
        assert rel(ops.gemm_ps(ia, ib, a_kslow=True, b_kslow=True, split_k=sk), ref) < 2e-6
    # bias + residual + ReLU, then accumulate, then alpha
    x, w = rnd(512, 256, seed=7), rnd(256, 256, seed=8)
    bias, res = rnd(256, seed=9), rnd(512, 256, seed=10)
    ix, iw = ops.split_image(x.to(DEV)), ops.split_image(w.to(DEV))
    o = ops.gemm_ps(ix, iw, bias=bias.to(DEV), add=res.to(DEV), relu=1)
    ref = torch.relu(x.double() @ w.double().t() + bias.double() + res.double())
    assert rel(o, ref) < 2e-6
    acc = res.clone().to(DEV)
    ops.gemm_ps(ix, iw, out=acc, accumulate=True, alpha=0.5)
    assert rel(acc, res.double() + 0.5 * (x.double() @ w.double().t())) < 2e-6
    # dropout epilogue regenerates the mask of the split-precision kernel (same seed, same element index)
    d1 = ops.gemm_ps(ix, iw, drop_p=0.1, seed=77)
    d2 = ops.gemm(x.to(DEV), w.to(DEV), drop_p=0.1, seed=77, mode='f16x3')
    assert torch.equal(d1 == 0, d2 == 0) and rel(d1, d2) < 1e-5


@pytest.mark.parametrize('M,N,K', [(768, 768, 30720), (2304, 768, 9984), (512, 256, 4096)])
def test_gemm_ps_split_k_through_workspace(ops, M, N, K, monkeypatch):
    """dW-shaped products (both operands K-slow, small output over a long K): the K splits store their partial tiles into slabs of a
    workspace and one pass sums them in a fixed order (qt_gemm_ps `ws`) — fp32-level against fp64, bit-identical from run to run,
    `accumulate` adds to what C holds, and equal (to summation order) to the atomics route (QT_PS_ATOMIC=1)."""
    a, b = rnd(K, M, seed=5, scale=1e-3), rnd(K, N, seed=6)
    am = ops.amax_of(a.to(DEV), keep=True)
    ia, ib = ops.split_image(a.to(DEV), amax=am), ops.split_image(b.to(DEV))
    ref = a.double().t() @ b.double()
    o1 = ops.gemm_ps(ia, ib, a_kslow=True, b_kslow=True)
    o2 = ops.gemm_ps(ia, ib, a_kslow=True, b_kslow=True)
    assert rel(o1, ref) < 2e-6 and torch.equal(o1, o2)
    c0 = rnd(M, N, seed=7, scale=1e-2)
    c = c0.clone().to(DEV)
    ops.gemm_ps(ia, ib, a_kslow=True, b_kslow=True, out=c, accumulate=True)
    assert rel(c, c0.double() + ref) < 2e-6
    monkeypatch.setenv('QT_PS_ATOMIC', '1')
    o3 = ops.gemm_ps(ia, ib, a_kslow=True, b_kslow=True)
    assert rel(o3, ref) < 2e-6 and rel(o1, o3) < 1e-6


@pytest.mark.parametrize('wmt', [2, 3, 4])
def test_gemm_ps_both_tile_heights(ops, wmt, monkeypatch):
    """The launcher picks the 192-row (WMT 3) or the 256-row (WMT 4) tile by its cost model; QT_PS_WMT pins it.  Row-form A
    operands on both heights: ragged M, several rounds of the persistent walk (> 256 tiles), the dX orientation, every epilogue,
    split-K atomics — all against fp64 and bit-identical dropout masks across the two heights."""
    monkeypatch.setenv('QT_PS_WMT', str(wmt))
    for (M, N, K, bks) in [(1000, 256, 96, False), (192 * 7 + 5, 384, 160, True), (256 * 9, 256 * 40, 64, False), (2048, 128, 2048, False)]:
        a, b = rnd(M, K, seed=11), rnd(*((K, N) if bks else (N, K)), seed=12)
        out = ops.gemm_ps(ops.split_image(a.to(DEV)), ops.split_image(b.to(DEV)), b_kslow=bks)
        ref = a.double() @ (b.double() if bks else b.double().t())
        assert rel(out, ref) < 2e-6, (M, N, K, bks)
    x, w = rnd(1200, 256, seed=7), rnd(320, 256, seed=8)
    bias, res = rnd(320, seed=9), rnd(1200, 320, seed=10)
    ix, iw = ops.split_image(x.to(DEV)), ops.split_image(w.to(DEV))
    o = ops.gemm_ps(ix, iw, bias=bias.to(DEV), add=res.to(DEV), relu=1)
    assert rel(o, torch.relu(x.double() @ w.double().t() + bias.double() + res.double())) < 2e-6
    acc = res.clone().to(DEV)
    ops.gemm_ps(ix, iw, out=acc, accumulate=True, alpha=0.5)
    assert rel(acc, res.double() + 0.5 * (x.double() @ w.double().t())) < 2e-6
    for sk in (1, 3):
        assert rel(ops.gemm_ps(ix, iw, split_k=sk), x.double() @ w.double().t()) < 2e-6
    d1 = ops.gemm_ps(ix, iw, drop_p=0.1, seed=77)
    d2 = ops.gemm(x.to(DEV), w.to(DEV), drop_p=0.1, seed=77, mode='f16x3')
    assert torch.equal(d1 == 0, d2 == 0) and rel(d1, d2) < 1e-5


@pytest.mark.parametrize('aks,bks', [(False, False), (False, True), (True, True), (True, False)])
@pytest.mark.parametrize('wmt', [4, 3, 2])
def test_gemm_ps_loop_forms_agree(ops, aks, bks, wmt, monkeypatch):
    """The late-barrier pipeline (the default K loop, gemm_plane.hip "PIPE": barrier in front of a step's last segment — or last two
    segments, PIPE 3 —, the next step's fragments fetched behind it, transfers one step further ahead) against the barrier-first loop
    (QT_PS_PIPE=0): the same MFMAs per accumulator in the same order, so the products must be BIT-identical — for one, two, three and many K steps per tile, ragged tiles, several rounds of
    the persistent tile walk (stage rotation across tiles), K splits, and both against fp64."""
    if wmt != 4 and aks:
        pytest.skip('the 192- and 128-row tiles take row-form A operands only')
    monkeypatch.setenv('QT_PS_WMT', str(wmt))
    for (M, N, K, sk) in [(512, 256, 32, 0), (512, 256, 64, 0), (640, 384, 96, 0), (1000 if not aks else 992, 320, 160, 0), (256 * 9, 256 * 33, 128, 1), (512, 256, 4096, 4)]:
        a = rnd(*((K, M) if aks else (M, K)), seed=51)
        b = rnd(*((K, N) if bks else (N, K)), seed=52)
        ia, ib = ops.split_image(a.to(DEV)), ops.split_image(b.to(DEV))
        outs = {}
        for form in ('0', '1', '3'):                      # barrier-first; late barrier with the one-segment / the two-segment tail
            monkeypatch.setenv('QT_PS_PIPE', form)
            outs[form] = ops.gemm_ps(ia, ib, a_kslow=aks, b_kslow=bks, split_k=sk)
            assert outs[form] is not None
        ref = (a.double().t() if aks else a.double()) @ (b.double() if bks else b.double().t())
        for form in ('1', '3'):
            assert rel(outs[form], ref) < 2e-6, (M, N, K, sk, form)
            if sk <= 1 or (aks and bks):                  # (K splits that meet in atomics have no fixed summation order)
                assert torch.equal(outs['0'], outs[form]), (M, N, K, sk, form)


def test_gemm_ps_batched_equals_one_launch_per_problem(ops):
    """qt_gemm_ps_batched: the H head slabs of a fold GEMM (column block h of x's image times row block h of W's image, K = d_h = three
    K steps, K-slow B) as ONE launch — bit-identical to one qt_gemm_ps launch per head, several rounds of the persistent walk."""
    M, H, dh, D = 2112, 8, 96, 768
    x, w = rnd(M, H * dh, seed=61), rnd(H * dh, D, seed=62, scale=0.05)
    xi, wi = ops.split_image(x.to(DEV)), ops.split_image(w.to(DEV), scaled=True)
    one = torch.empty(M, H * D, device=DEV)
    for h in range(H):
        a = ops.SplitImage(xi.data[:, h * dh:(h + 1) * dh], M, dh, xi.amax)
        b = ops.SplitImage(wi.data[h * dh:(h + 1) * dh], dh, D, wi.amax)
        assert ops.gemm_ps(a, b, b_kslow=True, out=one[:, h * D:(h + 1) * D], split_k=1) is not None
    bat = torch.full((M, H * D), float('nan'), device=DEV)
    a = ops.SplitImage(xi.data[:, :dh], M, dh, xi.amax)
    b = ops.SplitImage(wi.data[:dh], dh, D, wi.amax)
    assert ops.gemm_ps_batched(a, b, bat[:, :D], H, 4 * dh, 4 * dh * wi.data.stride(0), D, b_kslow=True)
    assert torch.equal(one, bat)
    ref = torch.einsum('mhk,hkd->mhd', x.double().view(M, H, dh), w.double().view(H, dh, D)).reshape(M, H * D)
    assert rel(bat, ref) < 2e-6


def test_gemm_ps_records_max_of_what_it_stores(ops):
    """c_amax: the epilogue folds max|C| of the values AS STORED (bias, ReLU, accumulate included) into the caller's record; a
    split-K launch cannot (partial sums) and leaves the tensor untagged."""
    x, w = rnd(1100, 256, seed=21), rnd(320, 256, seed=22)
    ix, iw = ops.split_image(x.to(DEV)), ops.split_image(w.to(DEV))
    for kw in (dict(), dict(bias=rnd(320, seed=23).to(DEV), relu=1), dict(b_kslow=True)):
        b = ops.split_image(rnd(256, 320, seed=24).to(DEV)) if kw.get('b_kslow') else iw
        rec = ops.new_amax(DEV)
        out = ops.gemm_ps(ix, b, c_amax=rec, split_k=1, **kw)
        assert ops.amax_tag(out) is rec and float(rec.max()) == float(out.abs().max())
    acc = rnd(1100, 320, seed=25).to(DEV) * 100
    rec = ops.new_amax(DEV)
    out = ops.gemm_ps(ix, iw, out=acc, accumulate=True, c_amax=rec, split_k=1)
    assert float(rec.max()) == float(acc.abs().max())
    rec = ops.new_amax(DEV)
    out = ops.gemm_ps(ix, iw, c_amax=rec, split_k=2)
    assert ops.amax_tag(out) is None and float(rec.max()) == 0.0
    # a tag dies with an in-place update of the tensor (autograd adds gradients in place)
    t = torch.ones(4, 4, device=DEV)
    ops.tag_amax(t, rec)
    assert ops.amax_tag(t) is rec
    t.add_(1.0)
    assert ops.amax_tag(t) is None


@pytest.mark.parametrize('sa,sb', [(1e-7, 1.0), (1e5, 1e-4), (3e-9, 2e-5), (1e3, 1e3)])
def test_gemm_ps_scaled_operands_any_magnitude(ops, sa, sb):
    """Gradient-like (1e-7..1e-9), tiny-weight (1e-4) and huge-activation (1e5) operands: with the per-tensor power-of-two scale
    the result stays fp32-equivalent; WITHOUT it the same operands lose precision or overflow (which is why it exists)."""
    M, N, K = 512, 256, 512
    a, b = rnd(M, K, seed=11, scale=sa), rnd(N, K, seed=12, scale=sb)
    ref = a.double() @ b.double().t()
    ia, ib = ops.split_image(a.to(DEV), scaled=True), ops.split_image(b.to(DEV), scaled=True)
    out = ops.gemm_ps(ia, ib)
    e = rel(out, ref)
    e32 = rel(ops.gemm(a.to(DEV), b.to(DEV), mode='fp32'), ref)
    print(f'scales {sa:g} x {sb:g}: ps(scaled) {e:.2e}, exact-fp32 kernel {e32:.2e}')
    assert e < 2e-6 and e < 4 * e32 + 5e-7
    raw = ops.gemm_ps(ops.split_image(a.to(DEV)), ops.split_image(b.to(DEV)))
    e_raw = rel(raw, ref) if bool(torch.isfinite(raw).all()) else float('inf')
    if min(sa, sb) < 1e-3 or max(sa, sb) > 6e4:
        assert e_raw > 10 * e


def test_gemm_ps_refuses_what_it_cannot_tile(ops):
    a, b = rnd(128, 64, seed=13), rnd(128, 64, seed=14)
    assert ops.gemm_ps(ops.split_image(a.to(DEV)), ops.split_image(b.to(DEV))) is None      # M < 256: the caller falls back


# --------------------------------------------------------------------------- #
# BASELINE configs[4]: e4m3 images + the GEMM on v_mfma_scale_f32_32x32x64_f8f6f4
# --------------------------------------------------------------------------- #
def _e4m3(x):
    """Round to OCP e4m3fn on the CPU (torch's float8_e4m3fn is that format)."""
    return x.to(torch.float8_e4m3fn).float()


@pytest.mark.parametrize('M,N,K', [(256, 128, 128), (512, 256, 768), (300, 384, 1024), (1024, 256, 2304)])
def test_gemm_f8_is_exact_on_fp8_representable_operands(ops, M, N, K):
    """Small-integer operands are exact in e4m3 and their products/sums exact in fp32: the kernel must reproduce the integer
    result bit for bit — this pins the lane <-> (row, k) assignment of the fp8 MFMA fragments."""
    g = torch.Generator().manual_seed(31)
    a = torch.randint(-4, 5, (M, K), generator=g).float()
    b = torch.randint(-3, 4, (N, K), generator=g).float()
    b[:, 0] = torch.arange(N).float() % 5 - 2                    # asymmetric structure
    ia, ib = ops.quant_fp8(a.to(DEV)), ops.quant_fp8(b.to(DEV))
    out = ops.gemm_f8(ia, ib)
    assert out is not None
    assert torch.equal(out.cpu(), (a.double() @ b.double().t()).float())


def test_quant_fp8_matches_torch_e4m3(ops):
    x = rnd(64, 256, seed=41, scale=3.0)
    img = ops.quant_fp8(x.to(DEV))
    s = 2.0 ** (7 - int(np.floor(np.log2(float(x.abs().max())))))
    got = img.data.cpu().view(torch.float8_e4m3fn).float()
    assert torch.equal(got, _e4m3(x * s))
    assert 128 <= float(got.abs().max()) <= 256


def test_gemm_f8_random_operands_and_epilogue(ops):
    M, N, K = 512, 256, 768
    a, b = rnd(M, K, seed=42, scale=0.7), rnd(N, K, seed=43, scale=0.05)
    bias, res = rnd(N, seed=44), rnd(M, N, seed=45)
    ia, ib = ops.quant_fp8(a.to(DEV)), ops.quant_fp8(b.to(DEV))
    out = ops.gemm_f8(ia, ib, bias=bias.to(DEV), add=res.to(DEV), relu=1)
    # exact reference of the mode's definition: operands rounded to e4m3 after the power-of-two scale, fp32+ accumulation
    sa = 2.0 ** (7 - int(np.floor(np.log2(float(a.abs().max()))))); sb = 2.0 ** (7 - int(np.floor(np.log2(float(b.abs().max())))))
    ref = torch.relu((_e4m3(a * sa).double() @ _e4m3(b * sb).double().t()) / (sa * sb) + bias.double() + res.double())
    assert rel(out, ref) < 5e-5                            # (measured 1.1e-5: the fp8 MFMA aligns its 64 products before the fp32 add)
    full = torch.relu(a.double() @ b.double().t() + bias.double() + res.double())
    assert 1e-3 < rel(out, full) < 0.2                     # really 3-bit-significand operands, and not garbage
