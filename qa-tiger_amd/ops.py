"""Host-side operators of the QA-TIGER fusion path: thin wrappers that hand raw
device pointers to libqatiger_hip.so, and the `torch.autograd.Function`s built on
them.  PyTorch supplies device memory, the current HIP stream and the autograd
graph; every FLOP runs in the hand-written kernels.  There is no CPU fallback:
a non-CUDA tensor raises.
"""
from __future__ import annotations

import ctypes as C
import math
from typing import List, Optional, Sequence, Tuple

import torch
from torch import Tensor

from . import _lib

LN_EPS = 1e-5

# GEMM arithmetic.  'fp32'  : v_mfma_f32_32x32x2_f32 (exact fp32 products)
#                   'f16x3' : three fp16 MFMAs per product on hi/lo-split operands (~fp32 accuracy, fp16 range)
#                   'bf16x3': three bf16 MFMAs per product (~1.5e-5 relative, fp32 range)
# Forward GEMMs see activations/weights (well inside fp16 range); backward GEMMs see gradients.
#                   'bf16x1': ONE bf16 MFMA per product (operands rounded to bf16, fp32 accumulate): the
#                             reduced-precision mode of BASELINE config 5 — outside the 1e-3 parity gate, judged by
#                             arg-max agreement; never the default and never the benchmarked metric.
# QT_GEMM=fp32 forces the exact-fp32 kernels everywhere; QT_GEMM=bf16x1 selects the reduced-precision mode.
import os as _os
if _os.environ.get('QT_GEMM', '').lower() == 'fp32':
    precision = {'fwd': 'fp32', 'bwd': 'fp32'}
elif _os.environ.get('QT_GEMM', '').lower() == 'bf16x1':
    precision = {'fwd': 'bf16x1', 'bwd': 'bf16x1'}
else:
    # bwd: fp16 parts too, on dY scaled by its max|dY| (fp32-equivalent products in both directions); QT_BWD=bf16x3 selects round 1's
    # cheaper backward (bf16 parts of the unscaled gradient: ~1.5e-5 per product, ~8 % faster train step)
    precision = {'fwd': 'f16x3', 'bwd': 'bf16x3' if _os.environ.get('QT_BWD', '').lower() == 'bf16x3' else 'f16x3'}
# Pre-split operand pipeline (csrc/gemm_plane.hip): Linear layers whose shapes fit run on split images — the input is split once
# (and reused by every layer that reads it and by the dW GEMM of the backward), weights are split once per optimizer step, the
# gradient dY is measured (max|dY|), scaled into fp16's normal range and split once for the dX and dW GEMMs: fp32-equivalent
# products (3 fp16 MFMAs on 11+11-bit parts) in the forward AND the backward.  QT_PLANE=0 keeps every GEMM on the
# split-in-the-consumer kernels (gemm_split.hip, with the same per-tensor scales applied inside the kernel).
plane = _os.environ.get('QT_PLANE', '1') != '0'      # and precision['fwd'] == 'f16x3', checked per call: bench.py switches modes at run time
weights_epoch = 0          # bumped whenever parameter memory is rewritten behind autograd's back (train.FusedAdam.step)
_HALF_TYPE = {'f16x3': 0, 'bf16x3': 1, 'bf16x1': 3, 'f16x3_t128': 16, 'bf16x3_t128': 17, 'f16x3_t256': 32, 'bf16x3_t256': 33,
              'f16x3_sk': 64, 'bf16x3_sk': 65, 'f16x3_nosk': 128, 'bf16x3_nosk': 129}   # _tNNN: force a tile (A/B testing)


# --------------------------------------------------------------------------- #
# plumbing
# --------------------------------------------------------------------------- #
_raw_stream = getattr(torch._C, '_cuda_getCurrentRawStream', None)
_dev_index: Optional[int] = None


def _stream() -> C.c_void_p:
    """hipStream_t of torch's current stream.  `torch.cuda.current_stream()` builds a Stream object (9 us a call, ~1 ms per
    train step of 2000 launches — the whole step is host-bound at B = 32); the raw-handle query is ~20x cheaper."""
    global _dev_index
    if _raw_stream is None:
        return C.c_void_p(torch.cuda.current_stream().cuda_stream)
    if _dev_index is None:
        _dev_index = torch.cuda.current_device()       # one process drives one GPU
    return C.c_void_p(_raw_stream(_dev_index))


def _ptr(t: Optional[Tensor]) -> Optional[C.c_void_p]:
    return None if t is None else C.c_void_p(t.data_ptr())


def _dev(*ts: Optional[Tensor]) -> None:
    for t in ts:
        if t is None:
            continue
        if not t.is_cuda:
            raise _lib.QTError('qa-tiger_amd kernels need tensors on an MI355X (cuda) device; got a '
                               f'{t.device.type} tensor and there is no CPU fallback')
        if t.dtype != torch.float32:
            raise _lib.QTError(f'fp32 tensor expected, got {t.dtype}')


def _on_gpu(*ts: Optional[Tensor]) -> None:
    """Device check for integer operands (token ids, labels, question types)."""
    for t in ts:
        if t is not None and not t.is_cuda:
            raise _lib.QTError('qa-tiger_amd kernels need tensors on an MI355X (cuda) device; got a '
                               f'{t.device.type} tensor and there is no CPU fallback')


def as2d(t: Tensor) -> Tensor:
    """View as [rows, cols] with unit column stride (copy only if unavoidable)."""
    if t.dim() != 2:
        t = t.reshape(-1, t.shape[-1])
    if t.stride(1) != 1 or (t.shape[0] > 1 and t.stride(0) < t.shape[1]):
        t = t.contiguous()
    return t


def _ld(t: Tensor) -> int:
    return t.stride(0) if t.shape[0] > 1 else max(t.stride(0), t.shape[1])


class _Seeds:
    """Dropout seeds: one fresh 64-bit value per dropout site per call, derived
    from torch's seed so `torch.manual_seed` makes runs repeatable."""

    def __init__(self):
        self.base = None
        self.count = 0

    def next(self) -> int:
        if self.base is None:
            self.base = torch.initial_seed() & 0xFFFFFFFFFFFFFFFF
        self.count += 1
        return (self.base * 0x9E3779B97F4A7C15 + self.count * 0xD1342543DE82EF95) & 0xFFFFFFFFFFFFFFFF

    def reset(self, base: Optional[int] = None):
        self.base, self.count = base, 0


seeds = _Seeds()


class KernelTimer:
    """Optional per-launch timing with HIP events recorded on the stream the kernels
    are launched on (torch's current stream).  Used by bench.py for the roofline."""

    def __init__(self):
        self.records = []          # (family, work, detail, start_event, end_event)

    def launch(self, family: str, work: float, detail, fn):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        rc = fn()
        e1.record()
        self.records.append((family, work, detail, e0, e1))
        return rc

    def summary(self):
        torch.cuda.synchronize()
        out = {}
        for fam, work, detail, e0, e1 in self.records:
            d = out.setdefault(fam, dict(ms=0.0, work=0.0, launches=0, by_detail={}))
            ms = e0.elapsed_time(e1)
            d['ms'] += ms; d['work'] += work; d['launches'] += 1
            bd = d['by_detail'].setdefault(detail, [0.0, 0.0, 0])
            bd[0] += ms; bd[1] += work; bd[2] += 1
        return out


timer: Optional[KernelTimer] = None


def _launch(family: str, work: float, detail, fn):
    return fn() if timer is None else timer.launch(family, work, detail, fn)


# --------------------------------------------------------------------------- #
# raw kernel calls (no autograd)
# --------------------------------------------------------------------------- #
def gemm(a: Tensor, b: Tensor, *, trans_a: bool = False, trans_b: bool = True, out: Optional[Tensor] = None,
         bias: Optional[Tensor] = None, add: Optional[Tensor] = None, relu: bool = False,
         mask_src: Optional[Tensor] = None, mask_scale: float = 1.0, drop_p: float = 0.0, seed: int = 0,
         accumulate: bool = False, alpha: float = 1.0, split_k: int = 0, mode: str = 'fp32',
         a_colsum: Optional[Tensor] = None, batch: Optional[Tuple[int, int, int, int]] = None,
         a_amax: Optional[Tensor] = None, b_amax: Optional[Tensor] = None, c_amax: Optional[Tensor] = None) -> Tensor:
    """out[M,N] = epilogue(alpha * op(a) @ op(b)); see include/qatiger.h qt_gemm_f32 / qt_gemm_x3.
    a_amax / b_amax (split modes only): device max|a| / max|b|; that operand is scaled into fp16's normal range before its split (gradient operands)."""
    a, b = as2d(a), as2d(b)
    _dev(a, b, bias, add, mask_src, out)
    if mode == 'fp8':                                   # configs[4]: only the Linear projections run on fp8 images (LinearFn);
        mode = 'f16x3'                                  # head-batched and odd-shaped products stay fp32-equivalent
    M, K = (a.shape[1], a.shape[0]) if trans_a else (a.shape[0], a.shape[1])
    N, Kb = (b.shape[0], b.shape[1]) if trans_b else (b.shape[1], b.shape[0])
    if K != Kb:
        raise _lib.QTError(f'gemm: inner dimensions differ ({K} vs {Kb})')
    if out is None:
        assert batch is None, 'batched gemm writes into a caller-provided view of the first problem'
        out = torch.empty((M, N), device=a.device, dtype=torch.float32)
    else:
        out = out if out.dim() == 2 else out.reshape(-1, out.shape[-1])
        assert out.shape == (M, N) and out.stride(1) == 1
    nb, a_bs, b_bs, c_bs = batch if batch is not None else (1, 0, 0, 0)
    if add is not None:
        add = as2d(add)
        assert add.shape == (M, N)
    if mask_src is not None:
        mask_src = as2d(mask_src)
        assert mask_src.shape == (M, N)
    tail = (int(trans_a), int(trans_b), M, N, K, float(alpha),
            _ptr(a), _ld(a), _ptr(b), _ld(b), _ptr(out), _ld(out),
            _ptr(bias), _ptr(add), _ld(add) if add is not None else 0, int(relu),
            _ptr(mask_src), _ld(mask_src) if mask_src is not None else 0, float(mask_scale),
            float(drop_p), C.c_ulonglong(seed), int(accumulate), int(split_k))
    btail = (int(nb), int(a_bs), int(b_bs), int(c_bs))
    if mode != 'fp32':
        ws = _splitk_workspace(a.device) if nb == 1 else None          # K splits of one problem meet in the workspace, not through atomics
        rc = _launch('gemm', 2.0 * M * N * K * nb, (M, N, K, int(trans_a), int(trans_b), mode) + ((nb,) if nb > 1 else ()),
                     lambda: _lib.load().qt_gemm_x3(_stream(), _HALF_TYPE[mode], *tail, _ptr(a_colsum), *btail, _ptr(a_amax), _ptr(b_amax), _ptr(c_amax),
                                                    _ptr(ws), ws.numel() if ws is not None else 0))
        if rc == 0:
            return out
        if rc == 2:                                    # done, but a split-K launch could not measure max|C|: one explicit pass
            _lib.check(_launch('split', 4.0 * out.numel(), ('amax',) + tuple(out.shape), lambda: _lib.load().qt_amax_f32(
                _stream(), _ptr(out), _ld(out), out.shape[0], out.shape[1], _ptr(c_amax), 1)), 'qt_amax_f32')
            return out
        if rc != -4:                                   # -4: shape not handled by the split kernel -> exact fp32 kernel
            _lib.check(rc, 'qt_gemm_x3')
        if timer is not None:
            timer.records.pop()

    ws = _splitk_workspace(a.device) if nb == 1 else None
    rc = _launch('gemm', 2.0 * M * N * K * nb, (M, N, K, int(trans_a), int(trans_b), 'fp32') + ((nb,) if nb > 1 else ()), lambda: _lib.load().qt_gemm_f32_ws(
        _stream(), int(trans_a), int(trans_b), M, N, K, float(alpha),
        _ptr(a), _ld(a), _ptr(b), _ld(b), _ptr(out), _ld(out),
        _ptr(bias), _ptr(add), _ld(add) if add is not None else 0, int(relu),
        _ptr(mask_src), _ld(mask_src) if mask_src is not None else 0, float(mask_scale),
        float(drop_p), C.c_ulonglong(seed), int(accumulate), int(split_k), *btail, _ptr(ws), ws.numel() if ws is not None else 0))
    _lib.check(rc, 'qt_gemm_f32')
    if a_colsum is not None:                           # the fp32 kernel has no fused column sum
        # (accumulate: C and a_colsum both hold earlier contributions — a shared weight's second use adds to its claimed slices)
        _lib.check(_lib.load().qt_colsum(_stream(), _ptr(a), _ld(a), a.shape[0], a.shape[1], _ptr(a_colsum), int(accumulate)), 'qt_colsum')
    if c_amax is not None:                             # ... and does not measure max|C|: one explicit pass into the caller's record
        _lib.check(_launch('split', 4.0 * out.numel(), ('amax',) + tuple(out.shape), lambda: _lib.load().qt_amax_f32(
            _stream(), _ptr(out), _ld(out), out.shape[0], out.shape[1], _ptr(c_amax), 1)), 'qt_amax_f32')
    return out


# --------------------------------------------------------------------------- #
# pre-split operands (csrc/gemm_plane.hip)
# --------------------------------------------------------------------------- #
class SplitImage:
    """A GEMM operand split once into fp16 (hi, lo) lines (include/qatiger.h qt_split_f16).  `data` has the shape and
    byte size of the fp32 tensor it was made from ([rows, cols] fp32-typed storage: an opaque byte image); `amax` is the
    device float the image was scaled by (None = unscaled)."""
    __slots__ = ('data', 'rows', 'cols', 'amax')

    def __init__(self, data: Tensor, rows: int, cols: int, amax: Optional[Tensor]):
        self.data, self.rows, self.cols, self.amax = data, rows, cols, amax


AMAX_SLOTS = 64          # a max|x| record is 64 floats; its value is their maximum (csrc/qt_common.h QT_AMAX_SLOTS)


class _AmaxPool:
    """Zeroed max|x| records without a launch per record: 2 x 1024 records, the half about to be handed out is zeroed by ONE kernel.
    A record stays readable for at least 1024 further takes — enough for a producer -> consumer hand-over inside one backward pass
    (a train step takes ~250), NOT for anything cached: a weight's maximum, an activation's maximum saved for the backward and
    every other record that outlives the pass is allocated on its own (`amax_of(..., keep=True)`)."""
    HALF = 1024

    def __init__(self):
        self.buf = None
        self.cur = 0

    def take(self, device) -> Tensor:
        if self.buf is None or self.buf.device != torch.device(device):
            self.buf = torch.empty((2 * self.HALF, AMAX_SLOTS), device=device, dtype=torch.float32)
            self.cur = 0
        if self.cur % self.HALF == 0:
            half = self.buf[self.cur:self.cur + self.HALF]
            _lib.check(_lib.load().qt_zero_f32(_stream(), _ptr(half), half.numel()), 'qt_zero_f32')
        rec = self.buf[self.cur]
        self.cur = (self.cur + 1) % (2 * self.HALF)
        return rec


    def zero_all(self) -> None:
        """Zero both halves now (after a graph capture: the capture recorded the zeroing kernel without running it, and the half beyond
        the captured range still holds warm-up values)."""
        if self.buf is not None:
            _lib.check(_lib.load().qt_zero_f32(_stream(), _ptr(self.buf), self.buf.numel()), 'qt_zero_f32')


_amax_pool = _AmaxPool()

# While a HIP graph is being captured (train.GraphedForward / GraphedTrainStep) nothing that is computed may come from, or go to, a
# host-side cache: a cached weight image would freeze the capture-time model state into the graph, and an image "made" during the
# capture holds no data until the first replay.  Operands made during a capture are shared within it through this dict (owned by the
# object that owns the graph, which also keeps them alive) and never touch the caches on parameters / tensors.
capture_keep: Optional[dict] = None


def _capturing() -> bool:
    return torch.cuda.is_current_stream_capturing()


def _captured(key, make):
    """`make()` once per capture and key (or every time when the capture is not one of ours)."""
    if capture_keep is None:
        return make()
    hit = capture_keep.get(key)
    if hit is None:
        hit = capture_keep[key] = make()
    return hit


_salt_word: Optional[Tensor] = None


def dropout_salt(device) -> Tensor:
    """The per-process dropout salt word (include/qatiger.h qt_set_dropout_salt): ONE int64 device word, allocated on first use,
    registered with the library and never freed or replaced, so the pointer the library keeps cannot dangle and every engine of the
    process bumps the same word.  Zero until a captured train step bumps it; eager steps then add its (constant between bumps) value to
    their seeds in forward and backward alike."""
    global _salt_word
    if _salt_word is None:
        _salt_word = torch.zeros(1, dtype=torch.int64, device=device)
    elif _salt_word.device != torch.device(device):
        raise _lib.QTError(f'dropout salt lives on {_salt_word.device}; one process drives one GPU')
    _lib.check(_lib.load().qt_set_dropout_salt(_ptr(_salt_word)), 'qt_set_dropout_salt')
    return _salt_word


def new_amax(device) -> Tensor:
    """A ZEROED max|x| record for a kernel that measures a maximum as a by-product (pooled: see _AmaxPool)."""
    return _amax_pool.take(device)


def amax_of(x: Tensor, out: Optional[Tensor] = None, keep: bool = False) -> Tensor:
    """max |x| as a device record of 64 floats (no host sync; take `.max()` to read it); `out` folds into an earlier one.
    keep=True: the record is cached by the caller beyond the current pass, so it gets memory of its own instead of a pooled slot."""
    x = as2d(x)
    _dev(x)
    if out is None:
        out = torch.zeros(AMAX_SLOTS, device=x.device, dtype=torch.float32) if keep else _amax_pool.take(x.device)
    _lib.check(_launch('split', 4.0 * x.numel(), ('amax',) + tuple(x.shape), lambda: _lib.load().qt_amax_f32(
        _stream(), _ptr(x), _ld(x), x.shape[0], x.shape[1], _ptr(out), 0)), 'qt_amax_f32')
    return out


def split_image(x: Tensor, scaled: bool = False, amax: Optional[Tensor] = None, colsum: Optional[Tensor] = None, keep: bool = False,
                colsum_accumulate: bool = False, out: Optional[Tensor] = None) -> SplitImage:
    """fp32 [rows, cols] -> SplitImage.  scaled=True measures max|x| first (one extra read of x) and scales the parts into
    fp16's normal range; pass `amax` when a producer has already measured it."""
    x = as2d(x)
    _dev(x, colsum)
    R, Cc = x.shape
    if Cc % 32:
        raise _lib.QTError(f'split_image: {Cc} columns is not a multiple of 32')
    if scaled and amax is None:
        amax = amax_of(x, keep=keep)
    img = out if (out is not None and out.shape == (R, Cc) and out.is_contiguous()) else torch.empty((R, Cc), device=x.device, dtype=torch.float32)
    rc = _launch('split', 8.0 * R * Cc, (R, Cc), lambda: _lib.load().qt_split_f16(
        _stream(), _ptr(x), _ld(x), R, Cc, _ptr(amax), _ptr(img), 4 * Cc, _ptr(colsum), int(colsum_accumulate)))
    _lib.check(rc, 'qt_split_f16')
    return SplitImage(img, R, Cc, amax)


def tag_amax(t: Tensor, rec: Tensor) -> None:
    """Attach a producer-measured max|x| record to the tensor OBJECT autograd will hand to the consumer.  The tag remembers the
    tensor's version counter: when autograd sums several gradients it may add into the first one in place (same object, larger
    values) — the bumped version makes the tag void and the consumer measures again."""
    t._qt_amax = (rec, t._version)


def _retag(t: Optional[Tensor], rec: Optional[Tensor]) -> None:
    """After an in-place accumulation into `t` by a raw kernel: replace its max|x| tag by the new record, or void it."""
    if t is None:
        return
    if rec is not None:
        tag_amax(t, rec)
    elif hasattr(t, '_qt_amax'):
        del t._qt_amax


use_amax_tags = True     # False: every consumer measures max|dY| itself (A/B and debugging of the tag protocol)


def amax_tag(t: Tensor) -> Optional[Tensor]:
    tag = getattr(t, '_qt_amax', None) if use_amax_tags else None
    if tag is None or tag[1] != t._version:
        return None
    return tag[0]


class _ReshapeFn(torch.autograd.Function):
    """x.reshape(shape) whose backward keeps the max|dY| tag on the gradient (torch's own view node hands on a new, untagged tensor)."""

    @staticmethod
    def forward(ctx, x: Tensor, shape):
        ctx.in_shape = x.shape
        return x.reshape(shape)

    @staticmethod
    def backward(ctx, g: Tensor):
        g2 = g.reshape(ctx.in_shape)
        rec = amax_tag(g)
        if rec is not None and g2 is not g:
            tag_amax(g2, rec)
        return g2, None


def _carry_image(src: Tensor, dst: Tensor) -> None:
    """A producer-written operand image (gemm_ps emit_image) belongs to the ROWS, whatever the leading dimensions are folded into."""
    cached = getattr(src, '_qt_img', None)
    if cached is not None and dst is not src and cached[0] == (src._version, src.data_ptr(), pass_epoch) and not getattr(src, '_qt_is_image', False) \
            and src.is_contiguous() and dst.is_contiguous() and dst.shape[-1] == src.shape[-1] and dst.data_ptr() == src.data_ptr():
        dst._qt_img = ((dst._version, dst.data_ptr(), pass_epoch), cached[1])


class _SplitRowsFn(torch.autograd.Function):
    """x [2n, ...] -> (x[:n], x[n:]) as views.  Autograd's own slice backward builds, per half, a zero-filled tensor of the FULL size,
    copies the half's gradient in and then adds the two (at the primary config, for the [2B, T, D] output of AVQCrossAttn and the
    [2BT, D] output of PatchSelecter's MLP: two fills, two copies and an add of 94 MB each, ~0.1 ms per site); here the two gradients
    are copied into the halves of one buffer."""

    @staticmethod
    def forward(ctx, x: Tensor, n: int):
        ctx.n, ctx.shape = n, x.shape
        return x[:n], x[n:]

    @staticmethod
    def backward(ctx, ga, gb):
        n = ctx.n
        ref = ga if ga is not None else gb
        gx = torch.empty(ctx.shape, device=ref.device, dtype=ref.dtype)
        for half, g in ((gx[:n], ga), (gx[n:], gb)):
            if g is None:
                half.zero_()
            else:
                half.copy_(g)
        return gx, None


class _JoinRowsFn(torch.autograd.Function):
    """(a [n, ...], b [m, ...]) -> their concatenation along dim 0 WITHOUT a copy when b starts where a ends in the same storage (the two
    halves `split_rows` made of one tensor, or two LayerNorm outputs written into the halves of one buffer: add_layer_norm(out=));
    torch.cat otherwise.  Backward hands each input its rows of the gradient (views)."""

    @staticmethod
    def forward(ctx, a: Tensor, b: Tensor):
        ctx.n = a.shape[0]
        if (a.is_contiguous() and b.is_contiguous() and a.dtype == b.dtype and a.shape[1:] == b.shape[1:]
                and a.untyped_storage().data_ptr() == b.untyped_storage().data_ptr()
                and b.storage_offset() == a.storage_offset() + a.numel()):
            return torch.as_strided(a, (a.shape[0] + b.shape[0],) + tuple(a.shape[1:]), a.stride(), a.storage_offset())
        return torch.cat([a, b], dim=0)

    @staticmethod
    def backward(ctx, g: Tensor):
        return g[:ctx.n], g[ctx.n:]


def join_rows(a: Tensor, b: Tensor) -> Tensor:
    """cat([a, b], dim=0), free when the two already lie back to back in one storage (see _JoinRowsFn)."""
    if torch.is_grad_enabled() and (a.requires_grad or b.requires_grad):
        return _JoinRowsFn.apply(a, b)
    return _JoinRowsFn.forward(_NoCtx(), a, b)


class _NoCtx:
    pass


def split_rows(x: Tensor, n: int):
    """(x[:n], x[n:]) whose gradients meet in one buffer (see _SplitRowsFn)."""
    if x.requires_grad and torch.is_grad_enabled():
        return _SplitRowsFn.apply(x, n)
    return x[:n], x[n:]


def reshape(x: Tensor, shape) -> Tensor:
    out = _ReshapeFn.apply(x, tuple(shape)) if x.requires_grad and torch.is_grad_enabled() else x.reshape(shape)
    _carry_image(x, out)
    return out


_splitk_ws: Optional[Tensor] = None
SPLITK_WS_FLOATS = 32 << 20          # 128 MiB: split x M x N of every dW product of the path (2304 x 768 at split 14, 3072 x 768 at split 7)


def _splitk_workspace(device) -> Tensor:
    """One workspace per process for the K-split slabs of the dW GEMMs (include/qatiger.h qt_gemm_ps `ws`): launches are stream-ordered,
    so consecutive GEMMs reuse it; allocated on first use (outside any graph capture: the warm-up steps come first)."""
    global _splitk_ws
    if _splitk_ws is None or _splitk_ws.device != torch.device(device):
        _splitk_ws = torch.empty(SPLITK_WS_FLOATS, device=device, dtype=torch.float32)
    return _splitk_ws


def gemm_ps(a: SplitImage, b: SplitImage, *, a_kslow: bool = False, b_kslow: bool = False, out: Optional[Tensor] = None,
            bias: Optional[Tensor] = None, add: Optional[Tensor] = None, relu: int = 0,
            mask_src: Optional[Tensor] = None, mask_scale: float = 1.0, drop_p: float = 0.0, seed: int = 0,
            accumulate: bool = False, alpha: float = 1.0, split_k: int = 0, c_amax: Optional[Tensor] = None,
            emit_image: bool = False) -> Optional[Tensor]:
    """out[M,N] = epilogue(alpha * op(a) @ op(b)) on split images; returns None (out untouched) when the shape does not fit
    the kernel (the caller then uses `gemm` on the fp32 tensors).
    c_amax: a zeroed max|x| record; when the launch could fill it (no split-K atomics) the returned tensor carries it as `_qt_amax`.
    emit_image: the epilogue also writes `out` as an unscaled split image (qt_gemm_ps_img); the returned tensor then carries it as
    `_qt_img`, where the Linear layer that consumes it finds it (input_image) instead of running a split pass."""
    M, K = (a.cols, a.rows) if a_kslow else (a.rows, a.cols)
    N, Kb = (b.cols, b.rows) if b_kslow else (b.rows, b.cols)
    if K != Kb:
        raise _lib.QTError(f'gemm_ps: inner dimensions differ ({K} vs {Kb})')
    _dev(bias, add, mask_src, out)
    if out is None:
        out = torch.empty((M, N), device=a.data.device, dtype=torch.float32)
    else:
        out = out if out.dim() == 2 else out.reshape(-1, out.shape[-1])
        assert out.shape == (M, N) and out.stride(1) == 1
    if add is not None:
        add = as2d(add)
    if mask_src is not None:
        mask_src = as2d(mask_src)
    if emit_image and N % 32 == 0 and not (a_kslow or b_kslow) and split_k in (0, 1) and not accumulate:
        img = torch.empty((M, N), device=out.device, dtype=torch.float32)
        rc = _launch('gemm', 2.0 * M * N * K, (M, N, K, 0, 0, 'ps'), lambda: _lib.load().qt_gemm_ps_img(
            _stream(), 0, 0, M, N, K, float(alpha),
            _ptr(a.data), 4 * a.data.stride(0), _ptr(a.amax), _ptr(b.data), 4 * b.data.stride(0), _ptr(b.amax),
            _ptr(out), _ld(out), _ptr(bias), _ptr(add), _ld(add) if add is not None else 0, int(relu),
            _ptr(mask_src), _ld(mask_src) if mask_src is not None else 0, float(mask_scale),
            float(drop_p), C.c_ulonglong(seed), 0, 1, _ptr(c_amax), _ptr(img), 4 * N))
        if rc == -4:
            if timer is not None:
                timer.records.pop()
            return None
        if rc not in (2, 3):
            _lib.check(rc, 'qt_gemm_ps_img')
            if c_amax is not None:
                tag_amax(out, c_amax)
            out._qt_img = ((out._version, out.data_ptr(), pass_epoch), SplitImage(img, M, N, None))
        return out
    ws = _splitk_workspace(out.device) if (a_kslow and b_kslow) else None      # dW products: K splits meet in a workspace, not through atomics
    rc = _launch('gemm', 2.0 * M * N * K, (M, N, K, int(a_kslow), int(b_kslow), 'ps'), lambda: _lib.load().qt_gemm_ps(
        _stream(), int(a_kslow), int(b_kslow), M, N, K, float(alpha),
        _ptr(a.data), 4 * a.data.stride(0), _ptr(a.amax), _ptr(b.data), 4 * b.data.stride(0), _ptr(b.amax),       # (a column block of an image is an image)
        _ptr(out), _ld(out), _ptr(bias), _ptr(add), _ld(add) if add is not None else 0, int(relu),
        _ptr(mask_src), _ld(mask_src) if mask_src is not None else 0, float(mask_scale),
        float(drop_p), C.c_ulonglong(seed), int(accumulate), int(split_k), _ptr(c_amax), _ptr(ws), ws.numel() if ws is not None else 0))
    if rc == -4:
        if timer is not None:
            timer.records.pop()
        return None
    if rc != 2:
        _lib.check(rc, 'qt_gemm_ps')
        if c_amax is not None:
            tag_amax(out, c_amax)
    return out


def gemm_ps_batched(a: SplitImage, b: SplitImage, out: Tensor, batch: int, a_bs_bytes: int, b_bs_bytes: int, c_bs_floats: int, *,
                    a_kslow: bool = False, b_kslow: bool = False, alpha: float = 1.0) -> bool:
    """`batch` products of one shape in one launch (include/qatiger.h qt_gemm_ps_batched): problem z multiplies the images `a`, `b` moved on by
    z strides and writes `out` moved on by z * c_bs_floats.  `a`, `b`, `out` describe problem 0.  False (nothing written) when the shape
    does not fit the kernel."""
    M, K = (a.cols, a.rows) if a_kslow else (a.rows, a.cols)
    N, Kb = (b.cols, b.rows) if b_kslow else (b.rows, b.cols)
    if K != Kb:
        raise _lib.QTError(f'gemm_ps_batched: inner dimensions differ ({K} vs {Kb})')
    _dev(out)
    assert out.shape == (M, N) and out.stride(1) == 1
    rc = _launch('gemm', 2.0 * M * N * K * batch, (M, N, K, int(a_kslow), int(b_kslow), 'ps', batch), lambda: _lib.load().qt_gemm_ps_batched(
        _stream(), int(a_kslow), int(b_kslow), M, N, K, float(alpha),
        _ptr(a.data), 4 * a.data.stride(0), _ptr(a.amax), _ptr(b.data), 4 * b.data.stride(0), _ptr(b.amax),
        _ptr(out), _ld(out), int(batch), int(a_bs_bytes), int(b_bs_bytes), int(c_bs_floats)))
    if rc == -4:
        if timer is not None:
            timer.records.pop()
        return False
    _lib.check(rc, 'qt_gemm_ps_batched')
    return True


def quant_fp8(x: Tensor, amax: Optional[Tensor] = None, keep: bool = False) -> SplitImage:
    """fp32 [rows, cols] -> e4m3 image (one byte per element, per-tensor power-of-two scale from max|x|): BASELINE configs[4]."""
    x = as2d(x)
    _dev(x)
    R, Cc = x.shape
    if Cc % 128:
        raise _lib.QTError(f'quant_fp8: {Cc} columns is not a multiple of 128')
    if amax is None:
        amax = amax_of(x, keep=keep)
    img = torch.empty((R, Cc), device=x.device, dtype=torch.uint8)
    _lib.check(_launch('split', 5.0 * R * Cc, ('fp8', R, Cc), lambda: _lib.load().qt_quant_fp8(
        _stream(), _ptr(x), _ld(x), R, Cc, _ptr(amax), _ptr(img), Cc)), 'qt_quant_fp8')
    return SplitImage(img, R, Cc, amax)


def gemm_f8(a: SplitImage, b: SplitImage, *, out: Optional[Tensor] = None, bias: Optional[Tensor] = None, add: Optional[Tensor] = None,
            relu: int = 0, drop_p: float = 0.0, seed: int = 0, alpha: float = 1.0) -> Optional[Tensor]:
    """out[M,N] = epilogue(alpha * a @ b^T) on e4m3 images (fp8 MFMA, fp32 accumulate); None when the shape does not fit."""
    M, K, N = a.rows, a.cols, b.rows
    if K != b.cols:
        raise _lib.QTError(f'gemm_f8: inner dimensions differ ({K} vs {b.cols})')
    _dev(bias, add, out)
    if out is None:
        out = torch.empty((M, N), device=a.data.device, dtype=torch.float32)
    if add is not None:
        add = as2d(add)
    rc = _launch('gemm', 2.0 * M * N * K, (M, N, K, 0, 0, 'fp8'), lambda: _lib.load().qt_gemm_f8(
        _stream(), M, N, K, float(alpha), _ptr(a.data), a.cols, _ptr(a.amax), _ptr(b.data), b.cols, _ptr(b.amax),
        _ptr(out), _ld(out), _ptr(bias), _ptr(add), _ld(add) if add is not None else 0, int(relu),
        None, 0, 1.0, float(drop_p), C.c_ulonglong(seed), 0, 1))
    if rc == -4:
        if timer is not None:
            timer.records.pop()
        return None
    _lib.check(rc, 'qt_gemm_f8')
    return out


def colsum(x: Tensor) -> Tensor:
    x = as2d(x)
    _dev(x)
    out = torch.empty((x.shape[1],), device=x.device, dtype=torch.float32)
    _lib.check(_lib.load().qt_colsum(_stream(), _ptr(x), _ld(x), x.shape[0], x.shape[1], _ptr(out), 0), 'qt_colsum')
    return out


def add_rows(a: Tensor, b: Tensor, alpha: float = 1.0, beta: float = 1.0, rows: Optional[int] = None) -> Tensor:
    """out[r] = alpha*a[r % rows(a)] + beta*b[r % rows(b)]."""
    a, b = as2d(a), as2d(b)
    _dev(a, b)
    rows = max(a.shape[0], b.shape[0]) if rows is None else rows
    D = a.shape[1]
    out = torch.empty((rows, D), device=a.device, dtype=torch.float32)
    rc = _lib.load().qt_add_rows(_stream(), _ptr(a), _ld(a), a.shape[0], float(alpha), _ptr(b), _ld(b), b.shape[0],
                                 float(beta), _ptr(out), D, rows, D)
    _lib.check(rc, 'qt_add_rows')
    return out


def mask_grad(g: Tensor, y: Optional[Tensor], drop_p: float, seed: int, want_amax: bool = False) -> Tensor:
    """g * (ReLU / dropout mask); with want_amax the result is tagged with its max|.| record, measured while it is written."""
    g = as2d(g)
    _dev(g, y)
    out = torch.empty_like(g, memory_format=torch.contiguous_format)
    if y is not None:
        y = as2d(y)
    am = new_amax(g.device) if (want_amax and precision['bwd'] == 'f16x3') else None
    rc = _lib.load().qt_mask_grad(_stream(), _ptr(g), _ld(g), _ptr(y), _ld(y) if y is not None else 0, float(drop_p),
                                  C.c_ulonglong(seed), _ptr(out), _ld(out), g.shape[0], g.shape[1], _ptr(am))
    _lib.check(rc, 'qt_mask_grad')
    if am is not None:
        tag_amax(out, am)
    return out


# --------------------------------------------------------------------------- #
# Linear:  y = dropout(relu(x W^T + b))        (reference nn.Linear call sites)
# --------------------------------------------------------------------------- #
# Parameters that own a slice of the flat gradient buffer (train.FlatParams sets `_qt_gv`) get their dW / db written
# there by the producing GEMM instead of into a temporary that a hook then copies.  Only the FIRST producer of a
# parameter in a step may do so (a second use of a shared weight would overwrite it); GradSync.begin() clears the set.
#
# With train.GradSync driving the step (`direct_grads`): every parameter's `.grad` IS its slice of the zeroed flat buffer for the
# whole step, a producer that claimed (a row range of) it accumulates there, reports it (`_grad_written`) and hands autograd
# None — no AccumulateGrad clone, no hook copy, and for a row slice of a packed MHA in-projection (q | k | v blocks used by
# different layers) no zero-padded full-size temporary either.  GradSync learns that a parameter is complete from `grad_ready_cb`.
grad_claims = set()         # id(parameter) (whole) or (id(parameter), first row, rows) (leading-dimension slice)
grads_prezeroed = False     # train.GradSync.begin() zeroes the flat gradient buffer with ONE launch; the first producer of a claimed
                            # parameter then ACCUMULATES into its slice (no zeroing launch per dW GEMM for split-K / bias column sums)
direct_grads = False        # set by train.GradSync.begin() (device steps only), cleared by finish()
grad_ready_cb = None        # callable(parameter): all rows of the parameter's gradient have been written in place
_claimed_rows = {}          # id(parameter) -> [parameter, rows written in place so far, [(r0, r1), ...]]
_fwd_use_rows = {}          # id(parameter) -> rows of it this step's FORWARD consumed through producers that may claim its gradient
grad_write_cb = None        # callable(parameter): one producer has written (part of) the parameter's gradient in place (GradSync.trace)


def _note_use(*params: Optional[Tensor]) -> None:
    """Forward side of the in-place gradient protocol: count how many rows of each parameter this step's forward consumes.  A
    parameter is reported complete from `_grad_written` only when the rows written in place equal the rows consumed — a weight used
    twice (QstGrounding is called twice per forward, net.py:136-138; the second use cannot claim and its gradient arrives through
    autograd's AccumulateGrad) is then completed by the post-accumulate hook, which runs after ALL uses, never by the first writer."""
    if not direct_grads:
        return
    for t in params:
        if t is None or not t.requires_grad:
            continue
        pr = _param_rows(t)
        if pr is not None:
            _fwd_use_rows[id(pr[0])] = _fwd_use_rows.get(id(pr[0]), 0) + pr[2]


def _param_rows(t: Tensor):
    """(parameter, first row, rows) when `t` is a FlatParams parameter or a contiguous leading-dimension slice of one."""
    if getattr(t, '_qt_gv', None) is not None:
        return t, 0, (t.shape[0] if t.dim() > 0 else 1)
    base = t._base
    # (row ranges of matrices AND of vectors: the q | k | v blocks of a packed in-projection weight and of its bias.  A block without an
    # in-place producer — b_v of a folded attention goes through a small autograd expression, b_k of one gets no gradient at all —
    # arrives through AccumulateGrad or not at all; the post-accumulate hook, which runs once every use of the parameter has run its
    # backward, then completes the parameter: train.GradSync._make_hook)
    if base is None or getattr(base, '_qt_gv', None) is None or t.dim() != base.dim() or t.dim() not in (1, 2) \
            or t.shape[1:] != base.shape[1:] or t.stride() != base.stride() or not base.is_contiguous():
        return None
    inner = 1
    for d in base.shape[1:]:
        inner *= d
    off = t.storage_offset() - base.storage_offset()
    if off < 0 or off % inner:
        return None
    return base, off // inner, t.shape[0]


def _claim_grad_view(param: Tensor, additive: bool = False):
    """The slice of the flat gradient buffer a producer may write `param`'s gradient into, or None (not a FlatParams parameter,
    already claimed this step — a shared weight's second use —, or an overlapping row range).
    additive=True: the producer ACCUMULATES into the slice (the buffer was zeroed at the start of the step).  Such a producer may also
    take rows that an earlier producer of this step has written — the second use of a shared weight (QstGrounding runs twice per
    forward, net.py:136-138) adds its share in place instead of sending it through autograd's AccumulateGrad; an overwriting
    producer (additive=False) still needs rows nobody has touched."""
    pr = _param_rows(param)
    if pr is None:
        return None
    base, r0, rows = pr
    gv = base._qt_gv
    if not gv.is_cuda:
        return None
    whole = r0 == 0 and (base.dim() == 0 or rows == base.shape[0])
    if not whole and not direct_grads:           # a row range can only be handed over in place (autograd wants the full-size gradient)
        return None
    ent = _claimed_rows.get(id(base))
    shared_ok = additive and direct_grads and grads_prezeroed
    if not shared_ok and (id(base) in grad_claims or (ent is not None and any(r0 < b and a < r0 + rows for a, b in ent[2]))):
        return None
    if whole:
        grad_claims.add(id(base))
    if ent is None:
        ent = _claimed_rows[id(base)] = [base, 0, []]
    ent[2].append((r0, r0 + rows))
    return gv if whole else gv[r0:r0 + rows]


def _unclaim(param: Tensor) -> None:
    pr = _param_rows(param)
    if pr is not None:
        base, r0, rows = pr
        ent = _claimed_rows.get(id(base))
        if ent is not None and (r0, r0 + rows) in ent[2]:
            ent[2].remove((r0, r0 + rows))
        n_rows = base.shape[0] if base.dim() > 0 else 1
        if ent is None or not any(a == 0 and b == n_rows for a, b in ent[2]):      # (an additive claim may sit on top of an earlier whole claim)
            grad_claims.discard(id(base))


def _grad_written(param: Tensor, view: Tensor):
    """The producer has launched the kernels that accumulate `param`'s gradient into `view` (from _claim_grad_view).
    Returns what the autograd Function should return for it: None in the direct mode, the view otherwise."""
    if not direct_grads:
        return view[:]
    base, r0, rows = _param_rows(param)
    ent = _claimed_rows[id(base)]
    ent[1] += rows
    if grad_write_cb is not None:
        grad_write_cb(base)
    # complete only if no other producer can still deliver: every forward use of the parameter has written in place (a use that could
    # not claim hands its gradient to autograd, whose AccumulateGrad + post-accumulate hook then complete the parameter after all uses)
    if ent[1] >= (base.shape[0] if base.dim() > 0 else 1) and _fwd_use_rows.get(id(base)) == ent[1] and grad_ready_cb is not None:
        grad_ready_cb(base)
    return None


class GradSink:
    """One gradient buffer for a tensor with several consumers.  Autograd sums the gradients of a multi-consumer tensor with one
    elementwise add per extra consumer ([2B T, D] rows in AVQCrossAttn: the stacked input feeds three in-projections and the residual of
    norm1 — three adds of 94 MB each at the primary config).  With `share_grad(x)` the consumers that know how (LinearFn's dX GEMM,
    the residual + LayerNorm backward) cooperate instead: the FIRST of them to run backward hands autograd its gradient tensor as
    usual and parks it here; every later one ACCUMULATES into that memory (GEMM epilogue `accumulate`) and hands autograd None.
    Autograd only moves the first tensor into the producer's input buffer and reads it when the producer's backward runs — after all
    consumers, by its own dependency count — so the sum is complete by then.  The generalisation of GradLink to any consumer order."""
    __slots__ = ('buf', 'holder')

    def __init__(self):
        self.buf = None          # [rows, cols] view of the parked gradient
        self.holder = None       # the tensor object autograd carries (re-tagged with max|sum| by accumulating producers)


def share_grad(x: Tensor) -> Tensor:
    """Mark an activation whose gradient contributions should meet in one buffer (see GradSink); returns x."""
    if x.requires_grad and torch.is_grad_enabled():
        sink = x._qt_sink = GradSink()

        def done(_g, sink=sink):                     # runs when x's gradient is complete (all consumers have run), before its producer's backward:
            sink.buf = sink.holder = None            # a second backward over a retained graph starts from an empty sink, and the buffer is not kept alive
        x.register_hook(done)
    return x


class GradLink:
    """Joins the two gradient contributions of a tensor `x` that feeds a residual block `x + f(x)`:
    `linear(o, W, b, residual=x, link=L)` (the block's last layer) parks its residual gradient in L, and
    `linear(x, W_in, b_in, link=L)` (the block's first layer, whose backward runs later) accumulates its dX GEMM into
    that buffer and hands autograd no gradient of its own — instead of autograd adding two [rows, D] tensors
    (0.32 ms per step on the 660 MB patch activations)."""
    __slots__ = ('buf', 'res')

    def __init__(self):
        self.buf = None
        self.res = None          # the tensor object autograd carries to the residual's producer: the accumulating dX GEMM tags it with max|sum|


# below this many rows a Linear stays on the split-in-the-consumer kernels: a handful of 256x256 tiles leaves the chip idle and
# the operand passes are pure launch latency there (measured: 256-row layers 28 us on qt_gemm_ps vs 17 us on qt_gemm_x3's 64-row tiles)
plane_min_rows = int(_os.environ.get('QT_PLANE_MIN_ROWS', '2048'))
plane_dw_min_rows = int(_os.environ.get('QT_PLANE_DW_MIN_ROWS', '2048'))    # below: forward and dX on the pre-split kernel, dW on the small-tile kernel
# (round 2: 8192 — the pre-split kernel's split-K atomics bound the short-K dW products; with the K splits meeting in a workspace it wins
# from 2048 rows on: B = 32 shard 8.67 -> 8.50 ms, B = 64 12.59 -> 12.13 ms per captured step)


def _plane_ok(rows: int, n_out: int, k_in: int) -> bool:
    """Shapes the 256x256x32 tile of qt_gemm_ps covers for all three products of a Linear layer (y, dX, dW)."""
    return plane and precision['fwd'] == 'f16x3' and precision['bwd'] == 'f16x3' and rows >= max(256, plane_min_rows) and rows % 32 == 0 and n_out >= 128 and n_out % 32 == 0 and k_in >= 128 and k_in % 32 == 0


def weight_image(weight: Tensor) -> Optional[SplitImage]:
    """Split image of a Linear weight [n_out, k_in] (or of a row slice of one, e.g. the q / kv blocks of a packed MHA
    in-projection), scaled by the parameter's max|w| and cached on the parameter until its memory changes."""
    base = weight._base if weight._base is not None else weight
    if base.dim() != 2 or weight.dim() != 2 or not base.is_contiguous() or weight.stride() != base.stride():
        return None
    rows_b, cols = base.shape
    if cols % 32 or weight.shape[1] != cols:
        return None
    flat = getattr(base, '_qt_flat', None)
    if flat is not None and base.data_ptr() == flat.flat.data_ptr() + 4 * base._qt_flat_off:
        # train.FlatParams: every weight's image is refreshed by ONE launch per optimizer step and lives in flat.img at the weight's own offset
        if flat.img_epoch != weights_epoch or getattr(base, '_qt_img_ver', -1) != base._version:
            with torch.no_grad():
                flat.refresh_images()
        o = base._qt_flat_off
        img = SplitImage(flat.img[o:o + base.numel()].view(rows_b, cols), rows_b, cols, base._qt_amax_view)
    elif base.is_cuda and _capturing():              # rebuilt inside the graph: replays follow the weights
        with torch.no_grad():
            img = _captured(('wimg', id(base)), lambda: split_image(base.detach(), scaled=True, keep=True))
    else:
        key = (base._version, weights_epoch, base.data_ptr())
        cached = getattr(base, '_qt_wimg', None)
        if cached is None or cached[0] != key:
            with torch.no_grad():
                cached = (key, split_image(base.detach(), scaled=True, keep=True))
            base._qt_wimg = cached
        img = cached[1]
    r0 = (weight.storage_offset() - base.storage_offset()) // cols
    if weight.shape[0] == rows_b and r0 == 0:
        return img
    return SplitImage(img.data[r0:r0 + weight.shape[0]], weight.shape[0], cols, img.amax)


def weight_image_fp8(weight: Tensor) -> Optional[SplitImage]:
    """e4m3 image of a Linear weight (or of a row slice of one), cached on the parameter like `weight_image`."""
    base = weight._base if weight._base is not None else weight
    if base.dim() != 2 or weight.dim() != 2 or not base.is_contiguous() or weight.stride() != base.stride():
        return None
    rows_b, cols = base.shape
    if cols % 128 or weight.shape[1] != cols:
        return None
    if _capturing():
        with torch.no_grad():
            img = _captured(('wimg8', id(base)), lambda: quant_fp8(base.detach(), keep=True))
    else:
        key = (base._version, weights_epoch, base.data_ptr())
        cached = getattr(base, '_qt_wimg8', None)
        if cached is None or cached[0] != key:
            with torch.no_grad():
                cached = (key, quant_fp8(base.detach(), keep=True))
            base._qt_wimg8 = cached
        img = cached[1]
    r0 = (weight.storage_offset() - base.storage_offset()) // cols
    if weight.shape[0] == rows_b and r0 == 0:
        return img
    return SplitImage(img.data[r0:r0 + weight.shape[0]], weight.shape[0], cols, img.amax)


pass_epoch = 0             # bumped at the start of every model forward (new_pass): activation-side caches never outlive one pass


def new_pass() -> None:
    """Start of a model forward.  Operand images / max records cached on activation tensors are shared by the layers of ONE pass (the three
    in-projections that read the same rows, ...) and must not survive it: a caller that feeds the same input tensor object to every step
    (a benchmark loop on one synthetic batch) would otherwise skip the split of its 880 MB patch features from the second step on —
    work a real training loop, whose batches are new tensors, does every step."""
    global pass_epoch
    pass_epoch += 1


def input_image_fp8(x: Tensor, x2: Tensor) -> SplitImage:
    if _capturing():                               # in the graph, shared within the capture only
        return _captured(('ximg8', id(x), x._version, tuple(x2.shape)), lambda: (quant_fp8(x2.detach(), keep=True), x))[0]
    cached = getattr(x, '_qt_img8', None)
    if cached is not None and cached[0] == (x._version, x.data_ptr(), pass_epoch) and cached[1].data.shape == x2.shape:
        return cached[1]
    img = quant_fp8(x2.detach(), keep=True)        # cached on the tensor object for the rest of this pass
    x._qt_img8 = ((x._version, x.data_ptr(), pass_epoch), img)
    return img


def input_image(x: Tensor, x2: Tensor) -> SplitImage:
    """Split image of an activation (x2 = the [rows, k_in] view of x), made once per tensor object: the layers that read the
    same tensor (the three in-projections of AVQCrossAttn, ...) and the dW GEMM of the backward share it."""
    if getattr(x, '_qt_is_image', False):           # the producer wrote the image itself (attention output): x IS its image, in or out of a capture
        return x._qt_img[1]
    if _capturing():
        # the split must be IN the graph (a cached image would freeze the warm-up batch into it) and is shared only within the capture
        # (the tuple keeps x alive, so id(x) stays unique for the capture)
        def make():
            with torch.no_grad():
                am = act_amax(x, x2)
                return split_image(x2.detach(), scaled=am is not None, amax=am), x
        return _captured(('ximg', id(x), x._version, tuple(x2.shape)), make)[0]
    cached = getattr(x, '_qt_img', None)
    if cached is not None and cached[0] == (x._version, x.data_ptr(), pass_epoch) and cached[1].data.shape == x2.shape:
        return cached[1]
    # A stale image on the same tensor object (an input fed to pass after pass) lends its memory to the new one when no autograd graph can
    # still hold it (inference): re-allocating a 2 GB image per pass next to its predecessor made the caching allocator fall back to
    # hipMalloc at the long-video config (host stalls of tens of ms).  Launches are stream-ordered: the old image's readers are ahead.
    # (only for the SAME contents — version and address unchanged, an earlier pass of this very tensor: a graph that still holds the old
    # image then finds identical bytes in it)
    reuse = cached[1].data if (cached is not None and cached[0][:2] == (x._version, x.data_ptr()) and not torch.is_grad_enabled()
                               and not getattr(x, '_qt_is_image', False)) else None
    with torch.no_grad():
        # robust-range mode: activations beyond fp16's range are scaled by their max|x| first (act_amax); by default the parts are unscaled
        am = act_amax(x, x2)
        img = split_image(x2.detach(), scaled=am is not None, amax=am, out=reuse)
    x._qt_img = ((x._version, x.data_ptr(), pass_epoch), img)
    return img


emit_images = _os.environ.get('QT_EMIT_IMAGE', '1') != '0'       # GEMM epilogues write the next Linear layer's operand image (A/B switch)
robust_range = _os.environ.get('QT_ROBUST_RANGE', '0') != '0'   # also measure max|x| of every Linear INPUT (one extra read of it per layer)


def weight_amax(weight: Tensor) -> Optional[Tensor]:
    """Device max|w| of the parameter a Linear weight (or a row slice of one) belongs to: the weight operand is scaled by the
    matching power of two before its fp16 split, so weights of any magnitude (1e-4 after weight decay, ...) keep 11 + 11 bits.
    With train.FlatParams every parameter's maximum comes from ONE launch per optimizer step (`_qt_amax_view`); otherwise it is
    measured once per parameter version and cached on the parameter."""
    if precision['fwd'] != 'f16x3':
        return None
    base = weight._base if weight._base is not None else weight
    view = getattr(base, '_qt_amax_view', None)
    if view is not None:
        flat = base._qt_flat
        if flat.amax_epoch != weights_epoch or getattr(base, '_qt_amax_ver', -1) != base._version:
            flat.refresh_amax()
        return view
    def measure():
        with torch.no_grad():
            return amax_of(base.detach().reshape(-1, base.shape[-1]) if base.dim() > 1 else base.detach().reshape(1, -1), keep=True)
    _dev(base)                                      # (a CPU tensor fails here, loudly, before any device query)
    if _capturing():
        return _captured(('wamax', id(base)), measure)
    key = (base._version, weights_epoch, base.data_ptr())
    cached = getattr(base, '_qt_wamax', None)
    if cached is None or cached[0] != key:
        cached = (key, measure())
        base._qt_wamax = cached
    return cached[1]


def act_amax(x: Tensor, x2: Tensor) -> Optional[Tensor]:
    """max|x| of a Linear input, only in the robust-range mode (activations beyond fp16's range, e.g. |x| ~ 1e5)."""
    if not robust_range or precision['fwd'] != 'f16x3':
        return None
    if _capturing():
        return _captured(('xamax', id(x), x._version), lambda: (amax_of(x2.detach(), keep=True), x))[0]
    cached = getattr(x, '_qt_xamax', None)
    if cached is not None and cached[0] == (x._version, x.data_ptr(), pass_epoch):
        return cached[1]
    am = amax_of(x2.detach(), keep=True)           # read again by the backward's dW GEMM: not a pooled slot
    x._qt_xamax = ((x._version, x.data_ptr(), pass_epoch), am)
    return am


def grad_amax(dy: Tensor, g2: Optional[Tensor] = None) -> Optional[Tensor]:
    """Device max|dY| for the scaled fp16 split of a backward GEMM's gradient operand (None in the modes that do not need it).
    A producer that has already measured it tags the tensor (`dy._qt_amax`); otherwise one streaming pass (qt_amax_f32)."""
    if precision['bwd'] != 'f16x3':
        return None
    tag = amax_tag(dy)
    if tag is not None:
        return tag
    return amax_of(dy if g2 is None else g2)


class LinearFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x: Tensor, weight: Tensor, bias: Optional[Tensor], relu: bool, drop_p: float, seed: int,
                residual: Optional[Tensor], grad_premasked: bool, link: Optional[GradLink] = None, emit_image: bool = False):
        ctx.link = link
        ctx.sink = getattr(x, '_qt_sink', None) if link is None else None
        shape = x.shape
        x2 = as2d(x)
        y = None
        ximg = wimg = None
        _note_use(weight, bias)
        if precision['fwd'] == 'fp8':
            # BASELINE configs[4]: projection GEMMs on fp8 MFMA (forward only; the router / Gaussian / softmax / LayerNorm
            # statistics and every small GEMM stay in fp32 — SURVEY.md Appendix E)
            if x2.shape[0] >= 256 and weight.shape[0] >= 128 and weight.shape[1] % 128 == 0:
                wq = weight_image_fp8(weight)
                if wq is not None:
                    y = gemm_f8(input_image_fp8(x, x2), wq, bias=bias, add=residual, relu=int(relu), drop_p=drop_p, seed=seed)
            if y is None:
                y = gemm(x2, weight, trans_b=True, bias=bias, add=residual, relu=relu, drop_p=drop_p, seed=seed, mode='f16x3')
            return y.reshape(*shape[:-1], weight.shape[0])
        if _plane_ok(x2.shape[0], weight.shape[0], weight.shape[1]):
            wimg = weight_image(weight)
            if wimg is not None:
                ximg = input_image(x, x2)
                y = gemm_ps(ximg, wimg, bias=bias, add=residual, relu=int(relu), drop_p=drop_p, seed=seed,
                            emit_image=emit_image and emit_images and not robust_range and x2.shape[0] >= plane_dw_min_rows)
        if getattr(x, '_qt_is_image', False) and (y is None or x2.shape[0] < plane_dw_min_rows):
            raise _lib.QTError('a tensor that holds a split image (attention output written for the pre-split GEMM) reached a Linear layer '
                               'that reads fp32 rows')
        if y is None:
            ximg = wimg = None
            wam = weight_amax(weight) if weight.shape[1] % 4 == 0 else None
            xam = act_amax(x, x2)
            y = gemm(x2, weight, trans_b=True, bias=bias, add=residual, relu=relu, drop_p=drop_p, seed=seed, mode=precision['fwd'],
                     a_amax=xam, b_amax=wam)
            ctx.wam = wam
            ctx.xam = xam                                # robust-range mode: the dW GEMM scales the saved input by the same record
        # the plane path saves the input's IMAGE (same bytes) for the dW GEMM, not the fp32 rows — unless the layer has so few rows
        # that its dW (a 768-wide output over a short K) is quicker on the small-tile split-K kernel, which reads fp32 (measured at the
        # 8-GPU shard, K = 3840: 49 us against 67-74 us; tools/sweep_splitk.py)
        ctx.dw_on_image = ximg is not None and x2.shape[0] >= plane_dw_min_rows
        ctx.save_for_backward(x2 if (ximg is None or not ctx.dw_on_image) else ximg.data, weight, y if relu else None)
        ctx.used_plane = ximg is not None
        if ximg is not None:
            ctx.xam = ximg.amax                          # None unless the robust-range mode scaled the input's image
        ctx.wimg = wimg                             # the weight's image (weights do not change between forward and backward)
        ctx.params = (weight, bias)                 # the Parameter objects themselves (saved tensors come back as new objects)
        ctx.cfg = (relu, drop_p, seed, bias is not None, residual is not None, grad_premasked, shape)
        out = y.reshape(*shape[:-1], weight.shape[0])
        _carry_image(y, out)
        return out

    @staticmethod
    def backward(ctx, dy: Tensor):
        x2, weight, y = ctx.saved_tensors
        relu, drop_p, seed, has_bias, has_res, premasked, shape = ctx.cfg
        g = as2d(dy)
        if (relu or drop_p > 0.0) and not premasked:
            g = dy = mask_grad(g, y if relu else None, drop_p, seed, want_amax=True)      # tagged with max|g| for grad_amax below
        dx = dw = db = dres = None
        link = ctx.link
        want_db = has_bias and ctx.needs_input_grad[2]
        w_param, b_param = ctx.params
        if ctx.used_plane:
            # dY: one max|dY| pass, one split pass (which also sums the columns = the bias gradient); both GEMMs then stream the image
            n_out, k_in = weight.shape
            bgv = None
            if want_db:
                bgv = _claim_grad_view(b_param, additive=True)       # (the column sums are accumulated whenever the slice is a claimed one)
                db = bgv[:] if bgv is not None else torch.empty((n_out,), device=g.device, dtype=torch.float32)
            gam = grad_amax(dy, g)                          # measured by dY's producer when it could (attention / LayerNorm / mask kernels)
            gimg = split_image(g, scaled=gam is None, amax=gam, colsum=db, colsum_accumulate=bgv is not None and grads_prezeroed)
            if bgv is not None:
                db = _grad_written(b_param, bgv)
            wimg = ctx.wimg
            ximg = SplitImage(x2, x2.shape[0], x2.shape[1], ctx.xam) if ctx.dw_on_image else None
            sink = ctx.sink
            if ctx.needs_input_grad[0]:
                # max|dX| is measured by the epilogue that stores it and rides on the tensor autograd hands to dX's consumer
                cam = new_amax(g.device)
                if link is not None and not has_res and link.buf is not None and link.buf.shape == (g.shape[0], k_in):
                    acc = gemm_ps(gimg, wimg, b_kslow=True, out=link.buf, accumulate=True, c_amax=cam)
                    if acc is None:
                        raise _lib.QTError('qt_gemm_ps refused a dX shape its forward accepted')
                    if link.res is not None and amax_tag(acc) is not None:
                        tag_amax(link.res, cam)         # dx stays None: the sum lives in the residual gradient
                    link.buf = link.res = None
                elif sink is not None and sink.buf is not None and sink.buf.shape == (g.shape[0], k_in):
                    acc = gemm_ps(gimg, wimg, b_kslow=True, out=sink.buf, accumulate=True, c_amax=cam)     # dx stays None: an earlier consumer's gradient holds the sum
                    if acc is None:
                        raise _lib.QTError('qt_gemm_ps refused a dX shape its forward accepted')
                    _retag(sink.holder, cam if amax_tag(acc) is not None else None)
                else:
                    dx2 = gemm_ps(gimg, wimg, b_kslow=True, c_amax=cam)
                    if dx2 is None:
                        raise _lib.QTError('qt_gemm_ps refused a dX shape its forward accepted')
                    dx = dx2.reshape(shape)
                    if amax_tag(dx2) is not None:
                        tag_amax(dx, cam)
                    if sink is not None and sink.buf is None:
                        sink.buf, sink.holder = dx2, dx
            if ctx.needs_input_grad[1]:
                wgv = _claim_grad_view(w_param, additive=True)
                # a claimed slice of the zeroed flat gradient buffer is accumulated into: no zeroing launch in front of a split-K GEMM
                if ctx.dw_on_image:
                    dw = gemm_ps(gimg, ximg, a_kslow=True, b_kslow=True, out=None if wgv is None else wgv[:],
                                 accumulate=wgv is not None and grads_prezeroed)
                    if dw is None:
                        raise _lib.QTError('qt_gemm_ps refused a dW shape its forward accepted')
                else:
                    dw = gemm(g, x2, trans_a=True, trans_b=False, mode=precision['bwd'], a_amax=gimg.amax, b_amax=ctx.xam,
                              out=None if wgv is None else wgv[:], accumulate=wgv is not None and grads_prezeroed)
                if wgv is not None:
                    dw = _grad_written(w_param, wgv)
        else:
            gam = grad_amax(dy, g)                          # max|dY|: the gradient operand's scale for both GEMMs
            sink = ctx.sink
            if ctx.needs_input_grad[0]:
                if link is not None and not has_res and link.buf is not None and link.buf.shape == (g.shape[0], weight.shape[1]):
                    # dx stays None; the sum's max|.| is measured by the epilogue that writes it and rides on the residual gradient
                    cam = new_amax(g.device) if gam is not None else None
                    gemm(g, weight, trans_b=False, mode=precision['bwd'], out=link.buf, accumulate=True, a_amax=gam, b_amax=ctx.wam, c_amax=cam)
                    if cam is not None and link.res is not None:
                        tag_amax(link.res, cam)
                    link.buf = link.res = None
                elif sink is not None and sink.buf is not None and sink.buf.shape == (g.shape[0], weight.shape[1]):
                    gemm(g, weight, trans_b=False, mode=precision['bwd'], out=sink.buf, accumulate=True, a_amax=gam, b_amax=ctx.wam)
                    _retag(sink.holder, None)               # (the parked gradient grew: whoever consumes it measures it again)
                else:
                    dx2 = gemm(g, weight, trans_b=False, mode=precision['bwd'], a_amax=gam, b_amax=ctx.wam)
                    dx = dx2.reshape(shape)
                    if sink is not None and sink.buf is None:
                        sink.buf, sink.holder = dx2, dx
            if ctx.needs_input_grad[1]:
                bgv = None
                # (claims are additive only when BOTH gradients can be accumulated in place: one GEMM launch has one `accumulate` flag)
                if want_db:                                 # bias gradient rides on the dW GEMM (it streams g anyway)
                    bgv = _claim_grad_view(b_param, additive=True)
                    db = bgv[:] if bgv is not None else torch.empty((g.shape[1],), device=g.device, dtype=torch.float32)
                wgv = _claim_grad_view(w_param, additive=(db is None or bgv is not None))
                if wgv is None and bgv is not None:         # the weight's slice was not available: the bias goes through autograd too
                    _unclaim(b_param)
                    bgv = None
                    db = torch.empty((g.shape[1],), device=g.device, dtype=torch.float32)
                inplace = grads_prezeroed and wgv is not None and (db is None or bgv is not None)
                dw = gemm(g, x2, trans_a=True, trans_b=False, mode=precision['bwd'], a_colsum=db, a_amax=gam,
                          b_amax=ctx.xam if precision['bwd'] == 'f16x3' else None,
                          out=None if wgv is None else wgv[:], accumulate=inplace)                                   # [N_out, K_in] = g^T x
                if wgv is not None:                         # written in place: autograd gets None in the direct mode
                    dw = _grad_written(w_param, wgv)
                if bgv is not None:
                    db = _grad_written(b_param, bgv)
            elif want_db:
                db = colsum(g)
        if has_res and ctx.needs_input_grad[6]:
            dres = g.reshape(*shape[:-1], weight.shape[0]) if not (relu or drop_p > 0.0) else None
            if dres is None:
                raise _lib.QTError('residual with relu/dropout epilogue is not differentiable here')
            if link is not None and g.dim() == 2 and g.stride(1) == 1:
                link.buf = g                            # same memory as dres: the linked first layer adds its dX here
                link.res = dres
        return dx, dw, db, None, None, None, dres, None, None, None


def linear(x: Tensor, weight: Tensor, bias: Optional[Tensor] = None, relu: bool = False, drop_p: float = 0.0,
           seed: int = 0, residual: Optional[Tensor] = None, grad_premasked: bool = False,
           link: Optional[GradLink] = None, emit_image: bool = False) -> Tensor:
    """emit_image: the result feeds another large Linear layer — let the GEMM epilogue write its operand image as well (the fp32 result is
    still written: residual adds, attention and LayerNorm kernels read it), so that layer needs no split pass."""
    if precision['fwd'] == 'fp8' and torch.is_grad_enabled() and (x.requires_grad or weight.requires_grad):
        raise _lib.QTError("precision 'fp8' (BASELINE configs[4]) is a forward-only mode: run it under torch.no_grad()")
    return LinearFn.apply(x, weight, bias, relu, drop_p, seed, residual, grad_premasked, link, emit_image)


# --------------------------------------------------------------------------- #
# attention core
# --------------------------------------------------------------------------- #
class AttnSpec:
    """Geometry of one attention call (see include/qatiger.h qt_attn_fwd).

    q lives in columns [q_col, q_col+E) of `qbuf`; k / v in columns [k_col, ..)
    and [v_col, ..) of `kvbuf` (which may be `qbuf` itself for packed QKV)."""

    def __init__(self, n_items, n_kv, kv_shift, H, Lq, Lk, E, q_col, k_col, v_col,
                 q_is=None, q_rs=1, o_is=None, o_rs=1, o_rows=None):
        self.n_items, self.n_kv, self.kv_shift, self.H = n_items, n_kv, kv_shift, H
        self.Lq, self.Lk, self.E = Lq, Lk, E
        self.q_col, self.k_col, self.v_col = q_col, k_col, v_col
        self.q_is = Lq if q_is is None else q_is
        self.q_rs = q_rs
        self.o_is = self.q_is if o_is is None else o_is
        self.o_rs = q_rs if o_is is None else o_rs
        self.o_rows = n_items * Lq if o_rows is None else o_rows
        self.dh = E // H
        self.scale = 1.0 / math.sqrt(self.dh)


def _attn_fwd(qbuf: Tensor, kvbuf: Tensor, s: AttnSpec, drop_p: float, seed: int, want_img: bool = False):
    """Returns (o, lse, is_image): with want_img the kernel may write o as a split image (same shape and bytes as the fp32 tensor)."""
    _dev(qbuf, kvbuf)
    o = torch.empty((s.o_rows, s.E), device=qbuf.device, dtype=torch.float32)
    lse = torch.empty((s.n_items * s.H * s.Lq,), device=qbuf.device, dtype=torch.float32)   # used by the streaming path
    f4 = 4
    nbytes = 4.0 * s.E * (2 * s.n_items * s.Lq + 2 * s.n_kv * s.Lk)
    head = lambda: (_stream(), C.c_void_p(qbuf.data_ptr() + s.q_col * f4), _ld(qbuf), s.q_is, s.q_rs,
                    C.c_void_p(kvbuf.data_ptr() + s.k_col * f4), _ld(kvbuf),
                    C.c_void_p(kvbuf.data_ptr() + s.v_col * f4), _ld(kvbuf), s.Lk,
                    _ptr(o), s.E, s.o_is, s.o_rs,
                    s.n_items, s.n_kv, s.kv_shift, s.H, s.Lq, s.Lk, s.dh, s.scale, float(drop_p), C.c_ulonglong(seed), _ptr(lse))
    if want_img and s.E % 32 == 0:
        # one buffer serves both outcomes: the image has the pitch and size of the fp32 tensor (rc 1: it holds the image)
        rc = _launch('attn_fwd', nbytes, (s.n_items, s.Lq, s.Lk, s.E), lambda: _lib.load().qt_attn_fwd_img(*head(), _ptr(o), 4 * s.E))
        _lib.check(0 if rc == 1 else rc, 'qt_attn_fwd_img')
        return o, lse, rc == 1
    rc = _launch('attn_fwd', nbytes, (s.n_items, s.Lq, s.Lk, s.E), lambda: _lib.load().qt_attn_fwd(*head()))
    _lib.check(rc, 'qt_attn_fwd')
    return o, lse, False


class AttentionFn(torch.autograd.Function):
    """softmax(q k^T / sqrt(d_h)) -> dropout -> v, reading q/k/v in place from the
    projection outputs.  `kvbuf is None` means packed QKV in `qbuf`."""

    @staticmethod
    def forward(ctx, qbuf: Tensor, kvbuf: Optional[Tensor], spec: AttnSpec, drop_p: float, seed: int, out_image: bool = False):
        qb = as2d(qbuf)
        kb = qb if kvbuf is None else as2d(kvbuf)
        o, lse, is_img = _attn_fwd(qb, kb, spec, drop_p, seed, want_img=out_image)
        # (the fp32 o is saved for the streaming backward's delta = rowsum(dO * O) only; the kernels that write an image do not need it)
        ctx.save_for_backward(qb, kb if kvbuf is not None else None, None if is_img else o, lse)
        ctx.cfg = (spec, drop_p, seed, qbuf.shape, None if kvbuf is None else kvbuf.shape)
        if is_img:
            # `o` holds the split image of the attention output, not fp32 rows: it may only feed a Linear on the pre-split path, which
            # finds the image here (input_image) and checks the flag
            # (the image record holds a detached alias of the buffer, not `o` itself: `o -> attribute -> SplitImage -> o` would be a
            # reference cycle that keeps the tensor AND the autograd graph behind it alive until Python's cyclic collector happens to
            # run — +38 MB per step at the primary config over a 400-step soak before this was found)
            img = SplitImage(o.detach(), o.shape[0], o.shape[1], None)
            o._qt_img = ((o._version, o.data_ptr(), pass_epoch), img)
            o._qt_is_image = True
        return o

    @staticmethod
    def backward(ctx, d_o: Tensor):
        qb, kb_saved, o_fwd, lse = ctx.saved_tensors
        s, drop_p, seed, qshape, kvshape = ctx.cfg
        packed = kb_saved is None
        kb = qb if packed else kb_saved
        d_o = as2d(d_o.contiguous())
        # gradients are written in the same packed layout as the inputs.  Columns
        # that hold no q/k/v of this call (none on this path) would stay zero.
        dqb = torch.zeros_like(qb) if (not packed and qb.shape[1] != s.E) else torch.empty_like(qb, memory_format=torch.contiguous_format)
        dkb = dqb if packed else torch.empty_like(kb, memory_format=torch.contiguous_format)
        f4 = 4
        delta_ws = torch.empty_like(lse)
        # packed QKV with no foreign columns: ONE gradient tensor leaves this node, so its max can be measured while it is written
        gam = new_amax(qb.device) if (packed and precision['bwd'] == 'f16x3' and qb.shape[1] == 3 * s.E) else None
        nbytes = 4.0 * s.E * (3 * s.n_items * s.Lq + 4 * s.n_kv * s.Lk)      # q, dO, dq + k, v, dk, dv
        rc = _launch('attn_bwd', nbytes, (s.n_items, s.Lq, s.Lk, s.E), lambda: _lib.load().qt_attn_bwd(
            _stream(), C.c_void_p(qb.data_ptr() + s.q_col * f4), _ld(qb), s.q_is, s.q_rs,
            C.c_void_p(kb.data_ptr() + s.k_col * f4), _ld(kb),
            C.c_void_p(kb.data_ptr() + s.v_col * f4), _ld(kb), s.Lk,
            _ptr(d_o), _ld(d_o), s.o_is, s.o_rs,
            C.c_void_p(dqb.data_ptr() + s.q_col * f4), _ld(dqb), s.q_is, s.q_rs,
            C.c_void_p(dkb.data_ptr() + s.k_col * f4), _ld(dkb),
            C.c_void_p(dkb.data_ptr() + s.v_col * f4), _ld(dkb),
            s.n_items, s.n_kv, s.kv_shift, s.H, s.Lq, s.Lk, s.dh, s.scale, float(drop_p), C.c_ulonglong(seed),
            _ptr(o_fwd), _ptr(lse), _ptr(delta_ws), _ptr(gam)))
        if rc == 1:                                    # the MFMA kernels measured max|dq,dk,dv| while writing the packed gradient
            rc = 0
        else:
            gam = None
        _lib.check(rc, 'qt_attn_bwd')
        dq_out = dqb.reshape(qshape)
        if gam is not None:
            tag_amax(dq_out, gam)                      # the in-projection's backward GEMMs scale dY by it (grad_amax)
        return dq_out, (None if packed else dkb.reshape(kvshape)), None, None, None, None


def attention(qbuf: Tensor, kvbuf: Optional[Tensor], spec: AttnSpec, drop_p: float = 0.0, seed: int = 0, for_linear: Optional[Tuple[int, int]] = None) -> Tensor:
    """for_linear = (n_out, k_in) of the Linear layer that is the ONLY consumer of the result: when that layer will run on the pre-split
    GEMM (forward, dX and dW on images) the attention kernel writes its output as the split image directly and the fp32 rows never exist."""
    img = False
    if for_linear is not None:
        rows = spec.o_rows
        img = _plane_ok(rows, for_linear[0], for_linear[1]) and rows >= plane_dw_min_rows and not robust_range and spec.E == for_linear[1]
    return AttentionFn.apply(qbuf, kvbuf, spec, drop_p, seed, img)


# --------------------------------------------------------------------------- #
# few-query attention with the K/V projections folded away (csrc/xattn.hip)
# --------------------------------------------------------------------------- #
fold_kv = True      # PatchSelecter.crs_attn (Lq=2) and TempMoE.qst_attn (Lq=1) skip the K/V projection GEMMs
fold_slab_min_rows = 8192   # query rows from which the H head slabs of the fold GEMMs go to the pre-split kernel one launch each (below: one head-batched launch)


head_slabs_batched = _os.environ.get('QT_HEAD_SLABS_BATCHED', '1') != '0'


def _head_slabs(ximg: SplitImage, wimg: SplitImage, out: Tensor, M: int, dh: int, D: int, H: int) -> None:
    """out[:, h*D:(h+1)*D] = x[:, h-th dh columns] @ W[h-th dh rows, :] for the H heads, on the pre-split kernel: A = the head's column block
    of x's image, B = the head's row block of W's image read K-slow.  Each slab is bound by writing its [M, D] floats (K = d_h is three K
    steps): ONE batched launch walks all H slabs' tiles (QT_HEAD_SLABS_BATCHED=0: one launch per head, round 2)."""
    a = SplitImage(ximg.data[:, :dh], M, dh, ximg.amax)
    b = SplitImage(wimg.data[:dh], dh, D, wimg.amax)
    if head_slabs_batched and H > 1:
        if not gemm_ps_batched(a, b, out[:, :D], H, 4 * dh, 4 * dh * wimg.data.stride(0), D, b_kslow=True):
            raise _lib.QTError('qt_gemm_ps_batched refused the head slabs of the fold GEMM')
        return
    for h in range(H):
        a = SplitImage(ximg.data[:, h * dh:(h + 1) * dh], M, dh, ximg.amax)
        b = SplitImage(wimg.data[h * dh:(h + 1) * dh], dh, D, wimg.amax)
        if gemm_ps(a, b, b_kslow=True, out=out[:, h * D:(h + 1) * D], split_k=1) is None:
            raise _lib.QTError('qt_gemm_ps refused a head slab of the fold GEMM')


class HeadExpandFn(torch.autograd.Function):
    """U[:, h, :] = q[:, h-th d_h slice] @ W[h-th d_h rows, :]  — q [M, H*dh], W [H*dh, D] -> U [M, H*D].
    With W = the key block of an MHA in-projection this is u_h = W_k,h^T q_h: the query moved to input space."""

    @staticmethod
    def forward(ctx, q: Tensor, w: Tensor, H: int):
        q2, w2 = as2d(q), as2d(w)
        _note_use(w)
        M, E = q2.shape
        D, dh = w2.shape[1], E // H
        u = torch.empty((M, H * D), device=q2.device, dtype=torch.float32)
        done = False
        if M >= fold_slab_min_rows and _plane_ok(M, D, E) and dh % 32 == 0 and D % 4 == 0:
            # per head on the pre-split kernel: A = the head's column block of q's image, B = the head's row block of W's image read
            # K-slow; the launch is bound by writing its [M, D] slab of u (94 MB at the primary config), which the float4 epilogue does
            wimg = weight_image(w)
            if wimg is not None:
                qimg = input_image(q, q2)
                _head_slabs(qimg, wimg, u, M, dh, D, H)
                done = True
        if not done:
            gemm(q2[:, :dh], w2[:dh], trans_b=False, out=u[:, :D], batch=(H, dh, dh * D, D), mode=precision['fwd'])
        ctx.save_for_backward(q2, w2)
        ctx.H = H
        ctx.w_obj = w                               # the caller's tensor object (a row block of a packed in-projection): its gradient slot
        return u

    @staticmethod
    def backward(ctx, du: Tensor):
        q2, w2 = ctx.saved_tensors
        H = ctx.H
        M, E = q2.shape
        D, dh = w2.shape[1], E // H
        du_in = du
        du = as2d(du.contiguous())
        gam = grad_amax(du_in, du) if du_in.is_contiguous() else grad_amax(du)       # FoldAttnFn.backward tags dU with its max
        dq = dw = None
        if ctx.needs_input_grad[0]:
            dq = torch.empty_like(q2)
            gemm(du[:, :D], w2[:dh], trans_b=True, out=dq[:, :dh], batch=(H, D, dh * D, dh), mode=precision['bwd'], a_amax=gam)
        if ctx.needs_input_grad[1]:
            wgv = _claim_grad_view(ctx.w_obj) if w2.is_contiguous() else None
            dw = wgv[:] if wgv is not None else torch.empty_like(w2)
            gemm(q2[:, :dh], du[:, :D], trans_a=True, trans_b=False, out=dw[:dh], batch=(H, dh, D, dh * D), mode=precision['bwd'], b_amax=gam)
            if wgv is not None:
                dw = _grad_written(ctx.w_obj, wgv)
        return dq, dw, None


class HeadContractFn(torch.autograd.Function):
    """o[:, h-th d_h slice] = ctx[:, h, :] @ W[h-th d_h rows, :]^T  — ctx [M, H*D], W [H*dh, D] -> o [M, H*dh]
    (the value block of the in-projection applied AFTER the attention-weighted sum of raw rows)."""

    @staticmethod
    def forward(ctx_, c: Tensor, w: Tensor, H: int):
        c2, w2 = as2d(c), as2d(w)
        _note_use(w)
        M = c2.shape[0]
        D = w2.shape[1]
        dh = w2.shape[0] // H
        o = torch.empty((M, H * dh), device=c2.device, dtype=torch.float32)
        gemm(c2[:, :D], w2[:dh], trans_b=True, out=o[:, :dh], batch=(H, D, dh * D, dh), mode=precision['fwd'])
        ctx_.save_for_backward(c2, w2)
        ctx_.H = H
        ctx_.w_obj = w
        return o

    @staticmethod
    def backward(ctx_, do: Tensor):
        c2, w2 = ctx_.saved_tensors
        H = ctx_.H
        D = w2.shape[1]
        dh = w2.shape[0] // H
        do_in = do
        do = as2d(do.contiguous())
        gam = grad_amax(do_in, do) if do_in.is_contiguous() else grad_amax(do)
        dc = dw = None
        if ctx_.needs_input_grad[0]:
            dc = torch.empty_like(c2)
            done = False
            M = do.shape[0]
            if M >= fold_slab_min_rows and _plane_ok(M, D, H * dh) and dh % 32 == 0 and D % 4 == 0 and gam is not None:
                wimg = weight_image(ctx_.w_obj)
                if wimg is not None:                    # same slab structure as HeadExpandFn.forward: dC_h = dO_h W_v,h
                    gimg = split_image(do, amax=gam)
                    _head_slabs(SplitImage(gimg.data, M, H * dh, gam), wimg, dc, M, dh, D, H)
                    done = True
            if not done:
                gemm(do[:, :dh], w2[:dh], trans_b=False, out=dc[:, :D], batch=(H, dh, dh * D, D), mode=precision['bwd'], a_amax=gam)
        if ctx_.needs_input_grad[1]:
            wgv = _claim_grad_view(ctx_.w_obj) if w2.is_contiguous() else None
            dw = wgv[:] if wgv is not None else torch.empty_like(w2)
            gemm(do[:, :dh], c2[:, :D], trans_a=True, trans_b=False, out=dw[:dh], batch=(H, dh, D, dh * D), mode=precision['bwd'], a_amax=gam)
            if wgv is not None:
                dw = _grad_written(ctx_.w_obj, wgv)
        return dc, dw, None


class FoldAttnFn(torch.autograd.Function):
    """(u [rows, H*D], x [n_items*Lk, D]) -> ctx [rows, H*D], asum [rows, H]; see include/qatiger.h qt_xattn_fwd."""

    @staticmethod
    def forward(ctx_, u: Tensor, x: Tensor, n_items: int, H: int, Lq: int, Lk: int, u_is: int, u_rs: int,
                scale: float, drop_p: float, seed: int):
        u2, x2 = as2d(u).contiguous(), as2d(x).contiguous()
        _dev(u2, x2)
        D = x2.shape[1]
        rows = u2.shape[0]
        assert u2.shape[1] == H * D and x2.shape[0] == n_items * Lk
        c = torch.empty_like(u2)
        lse = torch.empty((rows * H,), device=u2.device, dtype=torch.float32)
        asum = torch.empty((rows, H), device=u2.device, dtype=torch.float32)
        nbytes = 4.0 * D * (2 * rows * H + n_items * Lk)
        rc = _launch('attn_fwd', nbytes, ('fold', n_items, Lq, Lk, D), lambda: _lib.load().qt_xattn_fwd(
            _stream(), _ptr(u2), u_is, u_rs, _ptr(x2), n_items, H, Lq, Lk, D, float(scale), float(drop_p), C.c_ulonglong(seed),
            _ptr(c), _ptr(lse), _ptr(asum)))
        _lib.check(rc, 'qt_xattn_fwd')
        ctx_.save_for_backward(u2, x2, c, lse, asum)
        ctx_.cfg = (n_items, H, Lq, Lk, D, u_is, u_rs, scale, drop_p, seed, u.shape, x.shape)
        return c, asum

    @staticmethod
    def backward(ctx_, dc: Tensor, dasum: Optional[Tensor]):
        u2, x2, c, lse, asum = ctx_.saved_tensors
        n_items, H, Lq, Lk, D, u_is, u_rs, scale, drop_p, seed, ushape, xshape = ctx_.cfg
        dc = as2d(dc.contiguous())
        dasum = None if dasum is None else dasum.contiguous()
        du, dx = torch.empty_like(u2), torch.empty_like(x2)
        xam = new_amax(x2.device) if precision['bwd'] == 'f16x3' else None
        uam = new_amax(x2.device) if precision['bwd'] == 'f16x3' else None
        nbytes = 4.0 * D * (4 * u2.shape[0] * H + 2 * n_items * Lk)
        rc = _launch('attn_bwd', nbytes, ('fold', n_items, Lq, Lk, D), lambda: _lib.load().qt_xattn_bwd(
            _stream(), _ptr(u2), u_is, u_rs, _ptr(x2), _ptr(c), _ptr(lse), _ptr(asum), _ptr(dc), _ptr(dasum),
            n_items, H, Lq, Lk, D, float(scale), float(drop_p), C.c_ulonglong(seed), _ptr(du), _ptr(dx), _ptr(xam), _ptr(uam)))
        _lib.check(rc, 'qt_xattn_bwd')
        dx_out = dx.reshape(xshape)
        if xam is not None:
            tag_amax(dx_out, xam)                      # max|dx| measured while dx was written: the producer layer's dY scale
        du_out = du.reshape(ushape)
        if uam is not None:
            tag_amax(du_out, uam)                      # the head-batched W_k GEMMs (HeadExpandFn.backward) scale dU by it
        return du_out, dx_out, None, None, None, None, None, None, None, None, None


class HeadBiasFn(torch.autograd.Function):
    """out[m, e] = o[m, e] + asum[m, head(e)] * b_v[e]  (include/qatiger.h qt_headbias_fwd): the value bias of a folded attention."""

    @staticmethod
    def forward(ctx, o: Tensor, asum: Tensor, bv: Tensor, H: int):
        o2 = as2d(o)
        asum2 = asum.contiguous()
        _dev(o2, asum2, bv)
        _note_use(bv)
        rows, E = o2.shape
        out = torch.empty((rows, E), device=o2.device, dtype=torch.float32)
        _lib.check(_lib.load().qt_headbias_fwd(_stream(), _ptr(o2), _ld(o2), _ptr(asum2), _ptr(bv), rows, E, H, _ptr(out), E), 'qt_headbias_fwd')
        ctx.save_for_backward(asum2, bv)
        ctx.H, ctx.bv_obj, ctx.o_shape = H, bv, o.shape
        return out.reshape(o.shape)

    @staticmethod
    def backward(ctx, dout: Tensor):
        asum2, bv = ctx.saved_tensors
        d2 = as2d(dout)
        rows, E = d2.shape
        dasum = torch.empty_like(asum2)
        bgv = _claim_grad_view(ctx.bv_obj, additive=True) if (ctx.needs_input_grad[2] and grads_prezeroed) else None
        dbv = bgv[:] if bgv is not None else torch.empty((E,), device=d2.device, dtype=torch.float32)
        _lib.check(_lib.load().qt_headbias_bwd(_stream(), _ptr(d2), _ld(d2), _ptr(asum2), _ptr(bv), rows, E, ctx.H, _ptr(dasum), _ptr(dbv),
                                               int(bgv is not None)), 'qt_headbias_bwd')
        if bgv is not None:
            dbv = _grad_written(ctx.bv_obj, bgv)
        # d o = d out: the same tensor object goes on (its max|dY| tag with it)
        return dout, (dasum if ctx.needs_input_grad[1] else None), (dbv if ctx.needs_input_grad[2] else None), None


def folded_attention(q: Tensor, x: Tensor, mha, n_items: int, Lq: int, Lk: int, u_is: int, u_rs: int,
                     drop_p: float, seed: int) -> Tensor:
    """MHA(query rows q (already q-projected, [rows, E]), key = value = raw rows x [n_items*Lk, E]) WITHOUT the
    K/V in-projection GEMMs; returns the attention output before the out-projection, [rows, E].
    `mha` is the nn.MultiheadAttention parameter container."""
    E, H = mha.embed_dim, mha.num_heads
    dh = E // H
    W, b = mha.in_proj_weight, mha.in_proj_bias
    u = HeadExpandFn.apply(q, W[E:2 * E], H)                                    # W_k folded into the query
    c, asum = FoldAttnFn.apply(u, x, n_items, H, Lq, Lk, u_is, u_rs, 1.0 / math.sqrt(dh), drop_p, seed)
    o = HeadContractFn.apply(c, W[2 * E:], H)                                   # W_v after the weighted sum
    # + b_v,h * sum_p a_h[p]   (sum = 1 unless attention-probability dropout is active)
    if dh % 4 == 0 and H <= 32 and E <= 2048:
        return HeadBiasFn.apply(o, asum, b[2 * E:], H)
    sel = getattr(mha, '_qt_head_sel', None)                              # generic head widths: a K = H GEMM on wb[e, h] = b_v[e] * [e belongs to head h]
    if sel is None or sel.device != q.device or sel.shape != (E, H):
        sel = (torch.arange(E, device=q.device)[:, None] // dh == torch.arange(H, device=q.device)[None, :]).to(b.dtype)
        mha._qt_head_sel = sel
    wb = b[2 * E:, None] * sel                                            # [E, H]; differentiable in b_v
    return linear(asum, wb, None, residual=o)


# --------------------------------------------------------------------------- #
# y = LayerNorm(sum_i s_i * dropout_i(x_i))
# --------------------------------------------------------------------------- #
def _host_arrays(n, ptrs, lds, scales, drops, sds):
    P = (C.c_void_p * n)(*ptrs)
    L = (C.c_long * n)(*lds)
    S = (C.c_float * n)(*scales)
    Dr = (C.c_int * n)(*drops)
    Sd = (C.c_ulonglong * n)(*sds)
    return P, L, S, Dr, Sd


class AddLayerNormFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, gamma: Tensor, beta: Tensor, relu_out: bool, drop_p: float, scales, drops, sds, out, *xs: Tensor):
        n = len(xs)
        x2 = [as2d(x) for x in xs]
        _dev(gamma, beta, *x2)
        _note_use(gamma, beta)
        rows, D = x2[0].shape
        for x in x2:
            assert x.shape == (rows, D)
        if out is not None:
            out = out[0]
            assert out.is_contiguous() and out.numel() == rows * D and out.dtype == torch.float32 and out.device == x2[0].device
            y = out.view(rows, D)
        else:
            y = torch.empty((rows, D), device=x2[0].device, dtype=torch.float32)
        xhat = torch.empty_like(y)
        rstd = torch.empty((rows,), device=y.device, dtype=torch.float32)
        P, L, S, Dr, Sd = _host_arrays(n, [x.data_ptr() for x in x2], [_ld(x) for x in x2], scales, [int(d) for d in drops], sds)
        rc = _lib.load().qt_add_ln_fwd(_stream(), rows, D, n, P, L, S, Dr, Sd, float(drop_p), _ptr(gamma), _ptr(beta),
                                       LN_EPS, _ptr(y), D, int(relu_out), _ptr(xhat), _ptr(rstd))
        _lib.check(rc, 'qt_add_ln_fwd')
        ctx.save_for_backward(gamma, xhat, rstd, y if relu_out else None)
        ctx.sinks = [getattr(x, '_qt_sink', None) for x in xs]
        ctx.params = (gamma, beta)                  # the Parameter objects (for the flat-gradient claim in backward)
        ctx.cfg = (relu_out, drop_p, tuple(scales), tuple(drops), tuple(sds), [x.shape for x in xs])
        return y.reshape(xs[0].shape)

    @staticmethod
    def backward(ctx, dy: Tensor):
        gamma, xhat, rstd, y = ctx.saved_tensors
        relu_out, drop_p, scales, drops, sds, shapes = ctx.cfg
        n = len(shapes)
        g = as2d(dy)
        if relu_out:
            g = mask_grad(g, y, 0.0, 0)
        rows, D = xhat.shape
        dxs = [torch.empty((rows, D), device=g.device, dtype=torch.float32) if ctx.needs_input_grad[8 + i] else None
               for i in range(n)]
        acc_params = 0
        ggv = bgv = None
        if grads_prezeroed:                          # accumulate straight into the parameters' slices of the (zeroed) flat gradient buffer
            ggv = _claim_grad_view(ctx.params[0], additive=True)
            bgv = _claim_grad_view(ctx.params[1], additive=True) if ggv is not None else None
        if ggv is not None and bgv is not None:
            dgamma, dbeta, acc_params = ggv[:], bgv[:], 1
        else:
            if ggv is not None:
                _unclaim(ctx.params[0])
            dgb = torch.empty((2, D), device=g.device, dtype=torch.float32)      # side by side: zeroed by one launch
            dgamma, dbeta = dgb[0], dgb[1]
        # every dx_i is somebody's dY: its max|.| is measured while it is written (the producing layer's backward GEMMs scale by it)
        ams = [new_amax(g.device) if (d is not None and precision['bwd'] == 'f16x3') else None for d in dxs]
        AM = (C.c_void_p * n)(*[a.data_ptr() if a is not None else None for a in ams])
        P = (C.c_void_p * n)(*[d.data_ptr() if d is not None else None for d in dxs])
        L = (C.c_long * n)(*[D] * n)
        Acc = (C.c_int * n)(*[0] * n)
        S = (C.c_float * n)(*scales)
        Dr = (C.c_int * n)(*[int(d) for d in drops])
        Sd = (C.c_ulonglong * n)(*sds)
        rc = _lib.load().qt_add_ln_bwd(_stream(), rows, D, n, _ptr(g), _ld(g), _ptr(xhat), _ptr(rstd), _ptr(gamma),
                                       S, Dr, Sd, float(drop_p), P, L, Acc, _ptr(dgamma), _ptr(dbeta), AM, acc_params)
        _lib.check(rc, 'qt_add_ln_bwd')
        outs = [d.reshape(shp) if d is not None else None for d, shp in zip(dxs, shapes)]
        for o_, a_ in zip(outs, ams):
            if o_ is not None and a_ is not None:
                tag_amax(o_, a_)
        for d, o_, sk in zip(dxs, outs, ctx.sinks):  # the residual's gradient is usually the first contribution to its tensor: park it
            if sk is not None and sk.buf is None and d is not None:
                sk.buf, sk.holder = d, o_
        if acc_params:
            dgamma, dbeta = _grad_written(ctx.params[0], ggv), _grad_written(ctx.params[1], bgv)
        return (dgamma, dbeta, None, None, None, None, None, None, *outs)


def add_layer_norm(xs: Sequence[Tensor], gamma: Tensor, beta: Tensor, scales: Optional[Sequence[float]] = None,
                   drops: Optional[Sequence[bool]] = None, drop_p: float = 0.0, relu_out: bool = False, out: Optional[Tensor] = None) -> Tensor:
    """out (optional): a contiguous [rows, D] buffer (e.g. one half of a larger one: see join_rows) that receives the result."""
    n = len(xs)
    scales = [1.0] * n if scales is None else list(scales)
    drops = [False] * n if drops is None else list(drops)
    sds = [seeds.next() if (d and drop_p > 0.0) else 0 for d in drops]
    return AddLayerNormFn.apply(gamma, beta, relu_out, drop_p, scales, drops, sds, None if out is None else (out,), *xs)      # (boxed: a plain buffer, not an autograd input)


# --------------------------------------------------------------------------- #
# Gaussian-expert pieces (TempMoE)
# --------------------------------------------------------------------------- #
class MoECoefFn(torch.autograd.Function):
    """(router_logits [B,K], gauss_raw [B,2K]) -> coef [B,T,K], csum [B,K]
    plus non-differentiable diagnostics (topk_inds, topk_probs, gauss)."""

    @staticmethod
    def forward(ctx, logits: Tensor, raw: Tensor, T: int, K: int, k: int, sigma: float = 9.0):
        logits, raw = logits.contiguous(), raw.contiguous()
        _dev(logits, raw)
        B = logits.shape[0]
        dev = logits.device
        probs = torch.empty((B, K), device=dev)
        ti = torch.empty((B, k), device=dev, dtype=torch.int64)
        tp = torch.empty((B, k), device=dev)
        gauss = torch.empty((B, k, T), device=dev)
        coef = torch.empty((B, T, K), device=dev)
        csum = torch.empty((B, K), device=dev)
        lib = _lib.load()
        _lib.check(lib.qt_moe_router_fwd(_stream(), _ptr(logits), _ptr(raw), B, T, K, k, float(sigma), _ptr(probs), _ptr(ti), _ptr(tp),
                                         _ptr(gauss)), 'qt_moe_router_fwd')
        _lib.check(lib.qt_moe_coef_fwd(_stream(), _ptr(ti), _ptr(tp), _ptr(gauss), B, T, K, k, _ptr(coef), _ptr(csum)),
                   'qt_moe_coef_fwd')
        ctx.save_for_backward(ti, tp, gauss, probs, raw)
        ctx.dims = (B, T, K, k, float(sigma))
        ctx.mark_non_differentiable(ti, tp, gauss)
        return coef, csum, ti, tp, gauss

    @staticmethod
    def backward(ctx, dcoef, dcsum, _a, _b, _c):
        ti, tp, gauss, probs, raw = ctx.saved_tensors
        B, T, K, k, sigma = ctx.dims
        if dcoef is None:
            dcoef = torch.zeros((B, T, K), device=raw.device)
        dcoef = dcoef.contiguous()
        dcsum = None if dcsum is None else dcsum.contiguous()
        dlogits = torch.empty((B, K), device=raw.device)
        draw = torch.empty((B, 2 * K), device=raw.device)
        rc = _lib.load().qt_moe_router_bwd(_stream(), _ptr(dcoef), _ptr(ti), _ptr(tp), _ptr(gauss), _ptr(probs), _ptr(raw),
                                           _ptr(dcsum), B, T, K, k, sigma, _ptr(dlogits), _ptr(draw))
        _lib.check(rc, 'qt_moe_router_bwd')
        return dlogits, draw, None, None, None, None


class StackParamsFn(torch.autograd.Function):
    """K same-shape parameters -> one tensor: 'rows' = cat(dim=0) ([n, m] -> [K n, m]; [n] -> [K n]), 'cols' = cat(dim=1)
    ([d, n] -> [d, K n]), 'stack1' = stack(dim=1) ([d] -> [d, K]).  The experts of TempMoE are separate modules in the reference
    (modules.py:386-393, and so in the state_dict); the kernels want them stacked.  torch's own cat costs one kernel forward but
    K AccumulateGrad additions backward; under train.GradSync the K gradient slots lie at a constant pitch in the flat gradient
    buffer, so the backward is ONE strided accumulation into all of them (and autograd gets None)."""

    @staticmethod
    def forward(ctx, mode: str, *params: Tensor):
        ctx.mode, ctx.plist = mode, params
        _note_use(*params)
        if mode == 'rows':
            return torch.cat(params, dim=0)
        if mode == 'cols':
            return torch.cat(params, dim=1)
        return torch.stack(params, dim=1)

    @staticmethod
    def backward(ctx, g: Tensor):
        mode, params = ctx.mode, ctx.plist
        K, shp = len(params), params[0].shape
        if mode == 'rows':
            gk = g.reshape(K, *shp)
        elif mode == 'cols':
            gk = g.reshape(shp[0], K, shp[1]).permute(1, 0, 2)
        else:
            gk = g.t()
        if direct_grads:
            flat = getattr(params[0], '_qt_flat', None)
            offs = [getattr(p, '_qt_flat_off', None) for p in params]
            ok = flat is not None and all(o is not None for o in offs) and all(getattr(p, '_qt_flat', None) is flat and p.shape == shp for p in params)
            pitch = offs[1] - offs[0] if ok and K > 1 else 0
            ok = ok and K > 1 and all(offs[e] - offs[0] == e * pitch for e in range(K))
            if ok:
                views = [_claim_grad_view(p) for p in params]
                if all(v is not None for v in views):
                    inner = [1] * len(shp)
                    for d in range(len(shp) - 2, -1, -1):
                        inner[d] = inner[d + 1] * shp[d + 1]
                    slots = torch.as_strided(flat.grad, (K, *shp), (pitch, *inner), offs[0])
                    slots.add_(gk)                                   # the buffer was zeroed at the start of the step (grads_prezeroed)
                    for p, v in zip(params, views):
                        _grad_written(p, v)
                    return (None,) * (K + 1)
                for p, v in zip(params, views):
                    if v is not None:
                        _unclaim(p)
        return (None, *gk.unbind(0))


def stack_params(mode: str, params: Sequence[Tensor]) -> Tensor:
    return StackParamsFn.apply(mode, *params)


class MoEReduceFn(torch.autograd.Function):
    """R[s*B+b, e, :] = Σ_t coef[b,t,e] · H[(s*B+b)*T+t, e, :].  The returned dH
    already carries the ReLU mask of H (H is a ReLU output)."""

    @staticmethod
    def forward(ctx, H: Tensor, coef: Tensor, S: int, K: int):
        H = as2d(H)
        coef = coef.contiguous()
        _dev(H, coef)
        B, T, _ = coef.shape
        Dh = H.shape[1] // K
        assert H.shape[0] == S * B * T and H.is_contiguous()
        R = torch.empty((S * B, K * Dh), device=H.device)
        _lib.check(_lib.load().qt_moe_treduce_fwd(_stream(), _ptr(H), _ptr(coef), S, B, T, K, Dh, _ptr(R)), 'qt_moe_treduce_fwd')
        ctx.save_for_backward(H, coef)
        ctx.dims = (S, B, T, K, Dh)
        return R

    @staticmethod
    def backward(ctx, dR: Tensor):
        H, coef = ctx.saved_tensors
        S, B, T, K, Dh = ctx.dims
        dR = dR.contiguous()
        dH = torch.empty_like(H)
        dcoef = torch.empty_like(coef)
        ham = new_amax(H.device) if precision['bwd'] == 'f16x3' else None
        rc = _lib.load().qt_moe_treduce_bwd(_stream(), _ptr(H), _ptr(coef), _ptr(dR), S, B, T, K, Dh, _ptr(dH), _ptr(dcoef), _ptr(ham))
        _lib.check(rc, 'qt_moe_treduce_bwd')
        if ham is not None:
            tag_amax(dH, ham)                           # the stacked expert layer's backward GEMMs scale dH by it
        return dH, dcoef, None, None


# --------------------------------------------------------------------------- #
# loss
# --------------------------------------------------------------------------- #
class CrossEntropyFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, logits: Tensor, label: Tensor):
        logits = as2d(logits)
        _dev(logits)
        label = label.contiguous()
        assert label.dtype == torch.int64 and label.is_cuda
        B, A = logits.shape
        rows = torch.empty((B,), device=logits.device)
        mean = torch.empty((), device=logits.device)
        dlog = torch.empty((B, A), device=logits.device)
        rc = _lib.load().qt_xent_fwd_bwd(_stream(), _ptr(logits), _ld(logits), _ptr(label), B, A, _ptr(rows), _ptr(mean),
                                         _ptr(dlog), A)
        _lib.check(rc, 'qt_xent_fwd_bwd')
        ctx.save_for_backward(dlog)
        return mean

    @staticmethod
    def backward(ctx, g: Tensor):
        (dlog,) = ctx.saved_tensors
        return dlog * g, None      # scalar chain-rule factor (1.0 for loss.backward())


def cross_entropy(logits: Tensor, label: Tensor) -> Tensor:
    return CrossEntropyFn.apply(logits, label)


def adam_step_(param: Tensor, grad: Tensor, exp_avg: Tensor, exp_avg_sq: Tensor, step: int, lr: float,
               betas: Tuple[float, float] = (0.9, 0.999), eps: float = 1e-8, grad_scale: float = 1.0,
               weight_decay: float = 0.0) -> None:
    """In-place torch.optim.Adam update on flat fp32 buffers."""
    _dev(param, grad, exp_avg, exp_avg_sq)
    for t in (param, grad, exp_avg, exp_avg_sq):
        assert t.is_contiguous() and t.numel() == param.numel()
    rc = _lib.load().qt_adam_step(_stream(), _ptr(param), _ptr(grad), _ptr(exp_avg), _ptr(exp_avg_sq), param.numel(), int(step),
                                  float(lr), float(betas[0]), float(betas[1]), float(eps), float(grad_scale), float(weight_decay))
    _lib.check(rc, 'qt_adam_step')


# --------------------------------------------------------------------------- #
# frozen text tower + evaluation counters (csrc/text.hip) — forward only
# --------------------------------------------------------------------------- #
def text_embed(text: Tensor, token_embedding: Tensor, positional_embedding: Tensor) -> Tensor:
    """x[b,l,:] = token_embedding[text[b,l]] + positional_embedding[l]  (reference encoders.py:31-32)."""
    _on_gpu(text)
    _dev(token_embedding, positional_embedding)
    if text.dtype != torch.int32:
        text = text.to(torch.int32)
    text = text.contiguous()
    B, L = text.shape
    vocab, W = token_embedding.shape
    if L > positional_embedding.shape[0]:
        raise _lib.QTError(f'text_embed: {L} tokens but only {positional_embedding.shape[0]} positions')
    out = torch.empty((B, L, W), device=text.device, dtype=torch.float32)
    _lib.check(_lib.load().qt_text_embed_fwd(_stream(), _ptr(text), B, L, W, vocab, _ptr(token_embedding.contiguous()),
                                             _ptr(positional_embedding.contiguous()), _ptr(out)), 'qt_text_embed_fwd')
    return out


def text_eot_rows(text: Tensor, x: Tensor) -> Tuple[Tensor, Tensor]:
    """x[b, argmax_l text[b,l], :] and the chosen positions (reference encoders.py:37)."""
    _on_gpu(text)
    _dev(x)
    if text.dtype != torch.int32:
        text = text.to(torch.int32)
    text = text.contiguous()
    B, L = text.shape
    x = x.contiguous()
    W = x.shape[-1]
    out = torch.empty((B, W), device=x.device, dtype=torch.float32)
    idx = torch.empty((B,), device=x.device, dtype=torch.int32)
    _lib.check(_lib.load().qt_text_eot_rows(_stream(), _ptr(text), B, L, W, _ptr(x), _ptr(out), _ptr(idx)), 'qt_text_eot_rows')
    return out, idx


def causal_attention(qkv: Tensor, n_items: int, L: int, H: int) -> Tensor:
    """Causal self-attention on packed QKV rows [n_items*L, 3E] (item-major) -> [n_items*L, E]."""
    qkv = as2d(qkv)
    _dev(qkv)
    E = qkv.shape[1] // 3
    dh = E // H
    o = torch.empty((n_items * L, E), device=qkv.device, dtype=torch.float32)
    f4 = 4
    rc = _launch('attn_fwd', 4.0 * E * 4 * n_items * L, ('causal', n_items, L, L, E), lambda: _lib.load().qt_attn_causal_fwd(
        _stream(), _ptr(qkv), _ld(qkv), L, 1,
        C.c_void_p(qkv.data_ptr() + E * f4), _ld(qkv), C.c_void_p(qkv.data_ptr() + 2 * E * f4), _ld(qkv), L,
        _ptr(o), E, L, 1, n_items, H, L, dh, 1.0 / math.sqrt(dh)))
    _lib.check(rc, 'qt_attn_causal_fwd')
    return o


def eval_counts(logits: Tensor, label: Tensor, qtype: Optional[Tensor] = None, n_types: int = 9,
                totals: Optional[Tensor] = None, correct_by_type: Optional[Tensor] = None,
                total_by_type: Optional[Tensor] = None):
    """Device-side argmax + accuracy counters (reference trainutils.py:338-346, 417-423).

    Returns (predicted [B] int64, totals [2] int64 = (samples, correct), correct_by_type [n_types], total_by_type [n_types]);
    pass the three counter tensors back in to keep accumulating over batches without a host sync."""
    _on_gpu(label, qtype)
    _dev(logits)
    logits = as2d(logits)
    B, Cn = logits.shape
    dev = logits.device
    label = label.to(torch.int64).contiguous()
    if qtype is not None:
        qtype = qtype.to(torch.int32).contiguous()
    totals = torch.zeros(2, dtype=torch.int64, device=dev) if totals is None else totals
    correct_by_type = torch.zeros(n_types, dtype=torch.int64, device=dev) if correct_by_type is None else correct_by_type
    total_by_type = torch.zeros(n_types, dtype=torch.int64, device=dev) if total_by_type is None else total_by_type
    pred = torch.empty(B, dtype=torch.int64, device=dev)
    _lib.check(_lib.load().qt_eval_counts(_stream(), _ptr(logits), _ld(logits), B, Cn, _ptr(label), _ptr(qtype), n_types,
                                          _ptr(pred), _ptr(totals), _ptr(correct_by_type), _ptr(total_by_type)), 'qt_eval_counts')
    return pred, totals, correct_by_type, total_by_type


# --------------------------------------------------------------------------- #
# single-head attention with a large head dimension, built from batched GEMMs + row softmax (csrc/softmax.hip)
# (TSPM's nn.MultiheadAttention(512, nhead=1), reference src/models/tspm.py:20-21, 194), and the head-averaged
# attention weights nn.MultiheadAttention returns (tspm.py:99-100)
# --------------------------------------------------------------------------- #
def _batched(a: Tensor, b: Tensor, out: Tensor, nb: int, a_bs: int, b_bs: int, c_bs: int, trans_a=False, trans_b=True, mode='fp32',
             a_amax=None, b_amax=None):
    """`nb` equal problems; `a`, `b`, `out` are views of the FIRST one (unit column stride), the others follow at the
    given element strides."""
    return gemm(a, b, trans_a=trans_a, trans_b=trans_b, out=out, mode=mode, batch=(nb, a_bs, b_bs, c_bs), a_amax=a_amax, b_amax=b_amax)


class BmmAttentionFn(torch.autograd.Function):
    """softmax(q k^T / sqrt(E)) -> dropout -> v for ONE head of width E (any multiple of 4), items of Lq / Lk consecutive
    rows; query item n reads kv item (n + kv_shift) % n_items.  q in columns [q_col, q_col+E) of `qbuf`, k / v in columns
    [k_col, ..) / [v_col, ..) of `kvbuf` (None: packed in `qbuf`).  Scores and probabilities are materialised
    ([n_items*Lq, Lk]); every product is a batched GEMM over the items."""

    @staticmethod
    def forward(ctx, qbuf: Tensor, kvbuf: Optional[Tensor], n_items: int, Lq: int, Lk: int, E: int, q_col: int, k_col: int,
                v_col: int, kv_shift: int, drop_p: float, seed: int):
        qb = as2d(qbuf)
        kb = qb if kvbuf is None else as2d(kvbuf)
        _dev(qb, kb)
        scale = 1.0 / math.sqrt(E)
        S = torch.empty((n_items * Lq, Lk), device=qb.device, dtype=torch.float32)
        parts = [(0, n_items, kv_shift % n_items)] if kv_shift % n_items == 0 else \
            [(0, n_items - kv_shift % n_items, kv_shift % n_items), (n_items - kv_shift % n_items, n_items, -(n_items - kv_shift % n_items))]
        mode_f = precision['fwd']
        for lo, hi, sh in parts:                                   # items [lo, hi) read kv items [lo+sh, hi+sh)
            _batched(qb[lo * Lq:lo * Lq + Lq, q_col:q_col + E], kb[(lo + sh) * Lk:(lo + sh) * Lk + Lk, k_col:k_col + E],
                     S[lo * Lq:lo * Lq + Lq], hi - lo, Lq * _ld(qb), Lk * _ld(kb), Lq * Lk, trans_b=True, mode=mode_f)
        P = torch.empty_like(S)
        Y = S                                                       # dropped probabilities overwrite the scores
        _lib.check(_lib.load().qt_softmax_fwd(_stream(), _ptr(S), Lk, n_items * Lq, Lk, scale, float(drop_p), C.c_ulonglong(seed),
                                              _ptr(P), Lk, _ptr(Y), Lk), 'qt_softmax_fwd')
        O = torch.empty((n_items * Lq, E), device=qb.device, dtype=torch.float32)
        for lo, hi, sh in parts:
            _batched(Y[lo * Lq:lo * Lq + Lq], kb[(lo + sh) * Lk:(lo + sh) * Lk + Lk, v_col:v_col + E], O[lo * Lq:lo * Lq + Lq],
                     hi - lo, Lq * Lk, Lk * _ld(kb), Lq * E, trans_b=False, mode=mode_f)
        ctx.save_for_backward(qb, kb if kvbuf is not None else None, P)
        ctx.cfg = (n_items, Lq, Lk, E, q_col, k_col, v_col, parts, drop_p, seed, scale, qbuf.shape, None if kvbuf is None else kvbuf.shape)
        return O

    @staticmethod
    def backward(ctx, dO: Tensor):
        qb, kb_saved, P = ctx.saved_tensors
        n_items, Lq, Lk, E, q_col, k_col, v_col, parts, drop_p, seed, scale, qshape, kvshape = ctx.cfg
        packed = kb_saved is None
        kb = qb if packed else kb_saved
        dO = as2d(dO.contiguous())
        mode_b = precision['bwd']
        dqb = torch.zeros_like(qb, memory_format=torch.contiguous_format)
        dkb = dqb if packed else torch.zeros_like(kb, memory_format=torch.contiguous_format)
        # the dropped probabilities again (Y = P * mask, regenerated from the seed) instead of a second saved [rows, Lk] tensor
        dY = torch.empty_like(P)
        lib = _lib.load()
        if drop_p > 0.0:
            Y = torch.empty_like(P)
            _lib.check(lib.qt_mask_grad(_stream(), _ptr(P), Lk, None, 0, float(drop_p), C.c_ulonglong(seed), _ptr(Y), Lk, n_items * Lq, Lk, None), 'qt_mask_grad')
        else:
            Y = P
        gam_o = grad_amax(dO)                                        # gradient operands are scaled by their max before the fp16 split
        for lo, hi, sh in parts:
            nb = hi - lo
            # dV[kv item] = Y^T dO ;  dY = dO V^T
            _batched(Y[lo * Lq:lo * Lq + Lq], dO[lo * Lq:lo * Lq + Lq], dkb[(lo + sh) * Lk:(lo + sh) * Lk + Lk, v_col:v_col + E],
                     nb, Lq * Lk, Lq * E, Lk * _ld(dkb), trans_a=True, trans_b=False, mode=mode_b, b_amax=gam_o)
            _batched(dO[lo * Lq:lo * Lq + Lq], kb[(lo + sh) * Lk:(lo + sh) * Lk + Lk, v_col:v_col + E], dY[lo * Lq:lo * Lq + Lq],
                     nb, Lq * E, Lk * _ld(kb), Lq * Lk, trans_b=True, mode=mode_b, a_amax=gam_o)
        dS = dY
        _lib.check(lib.qt_softmax_bwd(_stream(), _ptr(dY), Lk, _ptr(P), Lk, n_items * Lq, Lk, scale, float(drop_p), C.c_ulonglong(seed),
                                      _ptr(dS), Lk), 'qt_softmax_bwd')
        gam_s = grad_amax(dS)
        for lo, hi, sh in parts:
            nb = hi - lo
            # dQ = dS K ;  dK[kv item] = dS^T Q
            _batched(dS[lo * Lq:lo * Lq + Lq], kb[(lo + sh) * Lk:(lo + sh) * Lk + Lk, k_col:k_col + E], dqb[lo * Lq:lo * Lq + Lq, q_col:q_col + E],
                     nb, Lq * Lk, Lk * _ld(kb), Lq * _ld(dqb), trans_b=False, mode=mode_b, a_amax=gam_s)
            _batched(dS[lo * Lq:lo * Lq + Lq], qb[lo * Lq:lo * Lq + Lq, q_col:q_col + E], dkb[(lo + sh) * Lk:(lo + sh) * Lk + Lk, k_col:k_col + E],
                     nb, Lq * Lk, Lq * _ld(qb), Lk * _ld(dkb), trans_a=True, trans_b=False, mode=mode_b, a_amax=gam_s)
        return (dqb.reshape(qshape), None if packed else dkb.reshape(kvshape)) + (None,) * 10


def bmm_attention(qbuf: Tensor, kvbuf: Optional[Tensor], n_items: int, Lq: int, Lk: int, E: int, q_col: int, k_col: int, v_col: int,
                  kv_shift: int = 0, drop_p: float = 0.0, seed: int = 0) -> Tensor:
    return BmmAttentionFn.apply(qbuf, kvbuf, n_items, Lq, Lk, E, q_col, k_col, v_col, kv_shift, drop_p, seed)


def attention_avg_weights(qbuf: Tensor, kvbuf: Tensor, s: AttnSpec, drop_p: float = 0.0, seed: int = 0) -> Tensor:
    """Head-averaged attention weights [n_items, Lq, Lk] of the attention `attention(qbuf, kvbuf, s, drop_p, seed)` applies
    (no gradient: the reference only takes arg-sorts of them, tspm.py:118-125)."""
    qb, kb = as2d(qbuf.detach()), as2d(kvbuf.detach())
    _dev(qb, kb)
    out = torch.empty((s.n_items, s.Lq, s.Lk), device=qb.device, dtype=torch.float32)
    f4 = 4
    _lib.check(_lib.load().qt_attn_avg_weights(
        _stream(), C.c_void_p(qb.data_ptr() + s.q_col * f4), _ld(qb), s.q_is, s.q_rs,
        C.c_void_p(kb.data_ptr() + s.k_col * f4), _ld(kb), s.Lk, s.n_items, s.n_kv, s.kv_shift, s.H, s.Lq, s.Lk, s.dh,
        s.scale, float(drop_p), C.c_ulonglong(seed), _ptr(out)), 'qt_attn_avg_weights')
    return out
