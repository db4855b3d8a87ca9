"""The `train()` step of the reference (src/trainutils.py:269-287) re-built for
one-process-per-GPU data parallelism on MI355X:

    zero_grad -> model(batch) -> CrossEntropyLoss -> backward -> [grad all-reduce] -> Adam.step

* `CrossEntropyLoss`  — nn.CrossEntropyLoss(mean) on the HIP loss kernel (train.py:41).
* `FlatParams`        — all trainable parameters live in ONE flat fp32 buffer (each
                        parameter a 16-byte-aligned view), with a matching flat gradient buffer.
* `GradSync`          — replaces DDP's bucketed gradient all-reduce (trainutils.py:103-106):
                        gradients are copied into the flat buffer as autograd finishes them and
                        each contiguous bucket is all-reduced asynchronously (RCCL over xGMI when
                        the backend is "nccl") while the rest of backward still runs.
* `FusedAdam`         — torch.optim.Adam(lr, betas, eps, weight_decay=0) (trainutils.py:141-144)
                        as one kernel over the flat buffers; the 1/world averaging is folded in.
* `DataParallelStep`  — the step above, end to end.
"""
from __future__ import annotations

from typing import Dict, Iterable, List, Optional, Sequence

import os as _os

import torch
import torch.distributed as dist
import torch.nn as nn
from torch import Tensor

from . import ops

_ALIGN = 32         # elements: every parameter view starts on a 128-byte boundary (one line of a split image, ops.weight_image)


class CrossEntropyLoss(nn.Module):
    """Drop-in for the `criterion` of reference train.py:41 (mean reduction, int64 targets)."""

    def forward(self, logits: Tensor, target: Tensor) -> Tensor:
        return ops.cross_entropy(logits, target)


class FlatParams:
    def __init__(self, params: Iterable[nn.Parameter]):
        self.params: List[nn.Parameter] = [p for p in params if p.requires_grad]
        if not self.params:
            raise ValueError('no trainable parameters')
        dev, dt = self.params[0].device, self.params[0].dtype
        self.offsets, off = [], 0
        for p in self.params:
            if p.device != dev or p.dtype != dt:
                raise ValueError('all parameters must share one device and dtype')
            self.offsets.append(off)
            off += (p.numel() + _ALIGN - 1) // _ALIGN * _ALIGN
        self.total = off
        self.flat = torch.zeros(self.total, device=dev, dtype=dt)
        self.grad = torch.zeros(self.total, device=dev, dtype=dt)
        with torch.no_grad():
            for p, o in zip(self.params, self.offsets):
                view = self.flat[o:o + p.numel()].view(p.shape)
                view.copy_(p.data)
                p.data = view                     # the module now computes from the flat buffer
        self.grad_views = [self.grad[o:o + p.numel()].view(p.shape) for p, o in zip(self.params, self.offsets)]
        # max|w| of every parameter (the fp16 split of a weight operand is scaled by it): one launch per optimizer step
        self.amax = torch.zeros((len(self.params), ops.AMAX_SLOTS), device=dev, dtype=dt)
        self._seg_off = torch.tensor(self.offsets, dtype=torch.int64, device=dev)
        self._seg_cnt = torch.tensor([p.numel() for p in self.params], dtype=torch.int64, device=dev)
        self.amax_epoch = None
        # split images of all weight matrices (the operands of the pre-split GEMMs), refreshed in one launch per optimizer step; made on first use
        self.img: Optional[torch.Tensor] = None
        self.img_epoch = None
        for i, (p, gv) in enumerate(zip(self.params, self.grad_views)):
            p._qt_gv = gv                          # ops.LinearFn writes dW / db straight into the flat gradient buffer
            if dev.type == 'cuda':
                p._qt_amax_view = self.amax[i]
                p._qt_flat = self
                p._qt_flat_off = self.offsets[i]

    def numel(self) -> int:
        return sum(p.numel() for p in self.params)

    def refresh_amax(self) -> None:
        """max|w| of every parameter in one launch (include/qatiger.h qt_amax_segments)."""
        lib = ops._lib.load()
        ops._lib.check(lib.qt_amax_segments(ops._stream(), ops._ptr(self.flat), ops._ptr(self._seg_off), ops._ptr(self._seg_cnt),
                                            len(self.params), ops._ptr(self.amax)), 'qt_amax_segments')
        self.amax_epoch = ops.weights_epoch
        for p in self.params:
            p._qt_amax_ver = p._version             # (a load_state_dict / copy_ into the parameters bumps the version: re-measure)

    def refresh_images(self) -> None:
        """max|w| and the split image of every weight matrix: two launches per optimizer step (qt_amax_segments, qt_split_f16_segments)."""
        if self.amax_epoch != ops.weights_epoch or any(getattr(p, '_qt_amax_ver', -1) != p._version for p in self.params):
            self.refresh_amax()
        if self.img is None:
            self.img = torch.empty_like(self.flat)
        lib = ops._lib.load()
        ops._lib.check(lib.qt_split_f16_segments(ops._stream(), ops._ptr(self.flat), ops._ptr(self._seg_off), ops._ptr(self._seg_cnt),
                                                 len(self.params), ops._ptr(self.amax), ops._ptr(self.img)), 'qt_split_f16_segments')
        self.img_epoch = ops.weights_epoch
        for p in self.params:
            p._qt_img_ver = p._version

    def check(self) -> None:
        """The parameters must still BE views of the flat buffer.  `model.to(...)`, `model.float()`, `load_state_dict(...,
        assign=True)` and similar re-allocate `p.data`; the optimizer would then update memory the model no longer reads.
        Checked (first and last parameter: two pointer reads) at the start of every step instead of failing silently."""
        for i in (0, len(self.params) - 1):
            p, o = self.params[i], self.offsets[i]
            if p.data_ptr() != self.flat.data_ptr() + o * self.flat.element_size():
                raise RuntimeError('FlatParams: a parameter no longer lives in the flat buffer (was the model moved or cast after '
                                   'the optimizer was built?); move the model first, then build FlatParams / DataParallelStep')


class GradSync:
    """Copy finished gradients into `flat.grad` and all-reduce them bucket by bucket.

    Buckets are contiguous slices of the flat buffer, filled from the END of the
    parameter list (autograd finishes the last-registered parameters first), so
    the first bucket is on the wire while most of backward is still to run.
    """

    def __init__(self, flat: FlatParams, process_group=None, bucket_bytes: int = 32 << 20):
        self.flat = flat
        self.pg = process_group
        self.world = dist.get_world_size(process_group) if dist.is_available() and dist.is_initialized() else 1
        # a one-rank group normally skips the collective; tests set this to drive the RCCL path on one GPU
        self.force_collective = False
        n = len(flat.params)
        self.bucket_of = [0] * n
        self.buckets: List[List[int]] = []        # [lo, hi, n_params]
        cur_hi, cur_lo, cur_n, b = flat.total, flat.total, 0, 0
        elem = flat.flat.element_size()
        for i in range(n - 1, -1, -1):
            cur_lo = flat.offsets[i]
            cur_n += 1
            self.bucket_of[i] = b
            if (cur_hi - cur_lo) * elem >= bucket_bytes or i == 0:
                self.buckets.append([cur_lo, cur_hi, cur_n])
                cur_hi, cur_n, b = cur_lo, 0, b + 1
        self._pending = [0] * len(self.buckets)
        self._seen = [False] * n
        self._works = []
        self._fired = [False] * len(self.buckets)
        self.defer = False                        # True: hooks only place gradients; every bucket is fired by finish() (captured train step)
        self.trace = None                         # set to a list to record ('grad', param index) / ('fire', bucket) / ('finish',) in order
        for i, p in enumerate(flat.params):
            p.register_post_accumulate_grad_hook(self._make_hook(i))
            p._qt_idx = i
        self.begin()

    def _make_hook(self, i: int):
        def hook(p: Tensor):
            gv = self.flat.grad_views[i]
            if p.grad is not gv:
                if p.grad.data_ptr() != gv.data_ptr():   # already in place when the producing GEMM wrote into gv
                    gv.copy_(p.grad)
                p.grad = gv                        # later accumulations (shared weights) add in place
            # AccumulateGrad (and this hook) runs once EVERY use of the parameter has run its backward — autograd's dependency count
            # includes the producers that wrote their share in place and returned None — so the gradient is complete here whatever mix
            # of in-place row blocks and autograd-delivered parts it is made of
            self._mark(i)
        return hook

    def _mark(self, i: int):
        if not self._seen[i]:
            self._seen[i] = True
            if self.trace is not None:
                self.trace.append(('grad', i))
            b = self.bucket_of[i]
            self._pending[b] -= 1
            if self._pending[b] == 0 and not self.defer:
                self._fire(b)

    def _written(self, p: Tensor):
        """ops.grad_write_cb (tracing only): a producer has accumulated (part of) p's gradient in place."""
        if getattr(p, '_qt_flat', None) is self.flat:
            self.trace.append(('write', p._qt_idx))

    def _claimed_ready(self, p: Tensor):
        """ops.grad_ready_cb: every row of p's gradient has been accumulated in place by its producers (no autograd hand-over)."""
        if getattr(p, '_qt_flat', None) is self.flat or p._qt_gv is self.flat.grad_views[p._qt_idx]:
            self._mark(p._qt_idx)

    def _fire(self, b: int):
        self._fired[b] = True
        if self.trace is not None:
            self.trace.append(('fire', b))
        if self.world > 1 or self.force_collective:
            lo, hi, _ = self.buckets[b]
            self._works.append(dist.all_reduce(self.flat.grad[lo:hi], op=dist.ReduceOp.SUM, group=self.pg, async_op=True))

    def begin(self):
        """Call before backward: forget last step's gradients (grad=None => autograd
        hands over fresh tensors, no memset of the flat buffer needed)."""
        self.flat.check()
        ops.grad_claims.clear()
        ops._claimed_rows.clear()
        ops._fwd_use_rows.clear()
        if self.flat.grad.is_cuda:                # one launch; claimed dW / db producers accumulate into it (ops.grads_prezeroed)
            ops._lib.check(ops._lib.load().qt_zero_f32(ops._stream(), ops._ptr(self.flat.grad), self.flat.grad.numel()), 'qt_zero_f32')
            ops.grads_prezeroed = True                   # first producer of a parameter's gradient this step may write in place
            # direct mode: .grad IS the (zeroed) slice for the whole step; producers that claim it accumulate there and hand autograd
            # None, anything autograd still delivers is added in place by AccumulateGrad (no clone, no copy back)
            for p, gv in zip(self.flat.params, self.flat.grad_views):
                p.grad = gv
            ops.direct_grads = True
            ops.grad_ready_cb = self._claimed_ready
            ops.grad_write_cb = self._written if self.trace is not None else None
        else:
            for p in self.flat.params:
                p.grad = None
        self.reset_host_state()

    def reset_host_state(self):
        """The host-side bookkeeping of begin().  A captured train step replays begin()'s DEVICE work only (GraphedTrainStep calls
        this before every replay: without it the second replay would find every bucket marked as already reduced)."""
        self._pending = [bk[2] for bk in self.buckets]
        self._seen = [False] * len(self.flat.params)
        self._fired = [False] * len(self.buckets)
        self._works = []

    def finish(self):
        """Call after backward: parameters that got no gradient this step count as
        zero; flush unfinished buckets; wait for the collectives."""
        if self.trace is not None:
            self.trace.append(('finish',))
        if not (self.flat.grad.is_cuda and ops.grads_prezeroed):
            for i, seen in enumerate(self._seen):
                if not seen:
                    self.flat.grad_views[i].zero_()
        ops.grads_prezeroed = False
        ops.direct_grads = False
        ops.grad_ready_cb = None
        ops.grad_write_cb = None
        for b, fired in enumerate(self._fired):
            if not fired:
                self._fire(b)
        for w in self._works:
            w.wait()
        self._works = []


class FusedAdam(torch.optim.Optimizer):
    """`torch.optim.Adam(params, lr, betas, eps, weight_decay)` (reference trainutils.py:141-144) as ONE kernel over the
    flat parameter / gradient / moment buffers.  A real `torch.optim.Optimizer`: one param group whose `lr`, `betas`,
    `eps` and `weight_decay` are read at every `step()`, so the reference's schedulers (`StepLR`, `ReduceLROnPlateau`,
    trainutils.py:147-182) drive it unchanged; `state_dict()` / `load_state_dict()` round-trip the moments (kept as views
    of the flat moment buffers) and the step count in torch's own per-parameter layout."""

    def __init__(self, flat: FlatParams, lr: float = 1e-4, betas=(0.95, 0.999), eps: float = 1e-8, weight_decay: float = 0.0):
        if lr < 0 or eps < 0 or weight_decay < 0 or not (0 <= betas[0] < 1 and 0 <= betas[1] < 1):
            raise ValueError(f'FusedAdam: bad hyper-parameters lr={lr} betas={betas} eps={eps} weight_decay={weight_decay}')
        self.flat = flat
        self.exp_avg = torch.zeros_like(flat.flat)
        self.exp_avg_sq = torch.zeros_like(flat.flat)
        self.step_count = 0
        super().__init__(flat.params, dict(lr=lr, betas=tuple(betas), eps=eps, weight_decay=weight_decay))
        self._bind_state()

    def _bind_state(self):
        for p, o in zip(self.flat.params, self.flat.offsets):
            n = p.numel()
            self.state[p] = dict(step=torch.tensor(float(self.step_count)),
                                 exp_avg=self.exp_avg[o:o + n].view(p.shape), exp_avg_sq=self.exp_avg_sq[o:o + n].view(p.shape))

    # the attribute spellings the rest of the repo (and round-1 callers) use
    @property
    def lr(self) -> float:
        return self.param_groups[0]['lr']

    @lr.setter
    def lr(self, v: float):
        self.param_groups[0]['lr'] = v

    @property
    def betas(self):
        return self.param_groups[0]['betas']

    @property
    def eps(self) -> float:
        return self.param_groups[0]['eps']

    @torch.no_grad()
    def step(self, closure=None, grad_scale: float = 1.0):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        if len(self.param_groups) != 1:
            raise ValueError('FusedAdam updates one flat buffer: exactly one param group')
        g = self.param_groups[0]
        self.step_count += 1
        ops.adam_step_(self.flat.flat, self.flat.grad, self.exp_avg, self.exp_avg_sq, self.step_count,
                       g['lr'], g['betas'], g['eps'], grad_scale, g['weight_decay'])
        ops.weights_epoch += 1                          # parameter memory rewritten by a raw kernel: cached weight images are stale
        return loss

    def sync_dev(self, step_dev: Tensor, hp_dev: Tensor, grad_scale: float = 1.0) -> None:
        """Write the host's view of the optimizer — step count, lr, betas, eps, weight_decay of the param group as they are NOW — into
        the device words a captured update reads (include/qatiger.h qt_adam_set_dev).  GraphedTrainStep calls this before every replay,
        so `StepLR.step()`, `ReduceLROnPlateau`, a hand-set `param_groups[0]['lr']` or `load_state_dict` after the capture reach the
        replayed kernel exactly as they reach the eager `step()` (reference trainutils.py:147-182)."""
        if len(self.param_groups) != 1:
            raise ValueError('FusedAdam updates one flat buffer: exactly one param group')
        g = self.param_groups[0]
        rc = ops._lib.load().qt_adam_set_dev(ops._stream(), ops._ptr(step_dev), ops._ptr(hp_dev), int(self.step_count), float(g['lr']),
                                            float(g['betas'][0]), float(g['betas'][1]), float(g['eps']), float(g['weight_decay']),
                                            float(grad_scale))
        ops._lib.check(rc, 'qt_adam_set_dev')

    @torch.no_grad()
    def step_dev(self, step_dev: Tensor, hp_dev: Tensor):
        """The same update with the step count AND the hyper-parameters in device memory (`sync_dev` wrote them; the kernel increments
        `step_dev` int64[1] and derives the bias corrections): the form a captured train step needs, whose kernel arguments are
        frozen (include/qatiger.h qt_adam_step_dev)."""
        self.step_count += 1
        rc = ops._lib.load().qt_adam_step_dev(ops._stream(), ops._ptr(self.flat.flat), ops._ptr(self.flat.grad), ops._ptr(self.exp_avg),
                                             ops._ptr(self.exp_avg_sq), self.flat.flat.numel(), ops._ptr(step_dev), ops._ptr(hp_dev))
        ops._lib.check(rc, 'qt_adam_step_dev')
        ops.weights_epoch += 1

    def state_dict(self) -> dict:
        for st in self.state.values():
            st['step'] = torch.tensor(float(self.step_count))
        return super().state_dict()

    def load_state_dict(self, state_dict: dict) -> None:
        super().load_state_dict(state_dict)             # torch hands back COPIES of the moments, keyed by parameter
        steps = set()
        with torch.no_grad():
            for p, o in zip(self.flat.params, self.flat.offsets):
                st, n = self.state.get(p), p.numel()
                if not st:
                    continue
                self.exp_avg[o:o + n].view(p.shape).copy_(st['exp_avg'])
                self.exp_avg_sq[o:o + n].view(p.shape).copy_(st['exp_avg_sq'])
                steps.add(int(float(st['step'])))
        if len(steps) > 1:
            raise ValueError(f'FusedAdam keeps one step count for all parameters; the checkpoint holds {sorted(steps)}')
        if steps:
            self.step_count = steps.pop()
        self._bind_state()


class DataParallelStep:
    """One rank of the data-parallel train step (weights replicated, batch sharded
    along dim 0 exactly like `DistributedSampler` + `batch_size // world`,
    reference trainutils.py:191-194)."""

    def __init__(self, model: nn.Module, lr: float = 1e-4, betas=(0.95, 0.999), eps: float = 1e-8,
                 process_group=None, bucket_bytes: int = 32 << 20, weight_decay: float = 0.0):
        self.model = model
        self.flat = FlatParams(model.parameters())
        self.sync = GradSync(self.flat, process_group, bucket_bytes)
        self.optim = FusedAdam(self.flat, lr, betas, eps, weight_decay)
        self.criterion = CrossEntropyLoss()

    def forward_backward(self, batch: Dict[str, Tensor]) -> Tensor:
        self.sync.begin()
        out = self.model(batch)
        loss = self.criterion(out['out'], batch['label'])
        loss.backward()
        self.sync.finish()
        return loss.detach()

    def step(self, batch: Dict[str, Tensor]) -> Tensor:
        loss = self.forward_backward(batch)
        self.optim.step(grad_scale=1.0 / self.sync.world)
        return loss


def shard_batch(batch: Dict[str, Tensor], rank: int, world: int) -> Dict[str, Tensor]:
    """Contiguous split of the global batch along dim 0 (B_local = B / world)."""
    out = {}
    for k, v in batch.items():
        if isinstance(v, Tensor) and v.dim() >= 1:
            B = v.shape[0]
            if B % world:
                raise ValueError(f'global batch {B} is not divisible by world size {world}')
            per = B // world
            out[k] = v[rank * per:(rank + 1) * per]
        else:
            out[k] = v
    return out



class GraphedTrainStep:
    """The whole train step of one rank — zero the gradient buffer, forward (dropout on), cross-entropy, backward, Adam — captured ONCE
    into a HIP graph and replayed.  A step is ~700 kernel launches; at the per-GPU batch of the 8-GPU strong-scaling point (B = 32) the
    GPU needs ~5 ms for them and Python/ctypes/autograd ~9 ms to issue them, so eager steps are host-bound there; a replay is one launch.

    What makes the step capturable (include/qatiger.h): no runtime memset/memcpy nodes from the library (every zeroing is a kernel), no
    allocation or synchronisation inside an entry point, dropout seeds salted from a DEVICE word that a one-thread kernel bumps inside the
    captured step (qt_set_dropout_salt / qt_bump_dropout_salt: fresh masks at every replay, forward and backward of one step still
    agree), the Adam step count, bias corrections AND hyper-parameters in device memory (qt_adam_set_dev / qt_adam_step_dev: the param
    group's lr / betas / eps / weight_decay and the host's step count are written before every replay, so schedulers and
    `load_state_dict` act on replays as they do on eager steps), max|x| records from a pool whose zeroing kernel is part of the
    capture, and the weights' max|w| records / split images refreshed INSIDE the captured step.  Only shapes are frozen at capture.

    world > 1: forward + backward are captured, the gradient all-reduce (RCCL) and the optimizer run eagerly after the replay
    (one replay + one collective per bucket + two launches per step)."""

    SALT_INC = 0x9E3779B97F4A7C15

    def __init__(self, engine: 'DataParallelStep', example: Dict[str, Tensor], warmup: int = 3):
        self.eng = engine
        dev = engine.flat.flat.device
        if dev.type != 'cuda':
            raise ops._lib.QTError('GraphedTrainStep needs an MI355X (cuda) device')
        self.distributed = engine.sync.world > 1 or engine.sync.force_collective
        self.lib = ops._lib.load()
        self.salt = ops.dropout_salt(dev)               # ONE word per process, owned by ops for the process lifetime (never freed or re-pointed)
        self.step_dev = torch.tensor([engine.optim.step_count], dtype=torch.int64, device=dev)
        self.hp = torch.zeros(8, dtype=torch.float32, device=dev)
        self.static_in = {k: (v.clone() if isinstance(v, Tensor) else v) for k, v in example.items()}
        engine.sync.defer = True
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(warmup):                     # lazy initialisation and allocator warm-up outside the capture
                self._sync_optim()
                self._captured_part()
                self._eager_part()
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        steps_before = engine.optim.step_count
        ops._amax_pool.cur = 0                          # the pool's zeroing kernel must be inside the capture
        engine.flat.amax_epoch = engine.flat.img_epoch = None   # ... and so must the refresh of the weights' max|w| records and images
        self.graph = torch.cuda.CUDAGraph()
        self._keep: dict = {}                           # operands made during the capture (weight images of parameters outside FlatParams, ...)
        ops.capture_keep = self._keep
        try:
            with torch.cuda.graph(self.graph):
                self.static_loss = self._captured_part()
        finally:
            ops.capture_keep = None
        engine.optim.step_count = steps_before          # the capture pass records work, it does not run it
        # ... so whatever the capture "refreshed" holds no data yet: host-side caches must not claim otherwise (an eager step between
        # capture and first replay re-measures), and the record pool's other half still carries warm-up values
        engine.flat.amax_epoch = engine.flat.img_epoch = None
        ops._amax_pool.zero_all()
        ops.grads_prezeroed = False                     # the captured begin() left its host flags set; each replay / eager step sets its own
        ops.direct_grads = False
        ops.grad_ready_cb = None

    def _sync_optim(self) -> None:
        self.eng.optim.sync_dev(self.step_dev, self.hp, grad_scale=1.0 / self.eng.sync.world if self.distributed else 1.0)

    def _captured_part(self) -> Tensor:
        eng = self.eng
        ops._lib.check(self.lib.qt_bump_dropout_salt(ops._stream(), ops._ptr(self.salt), self.SALT_INC), 'qt_bump_dropout_salt')
        eng.sync.begin()
        out = eng.model(self.static_in)
        loss = eng.criterion(out['out'], self.static_in['label'])
        loss.backward()
        if not self.distributed:
            eng.sync.finish()
            eng.optim.step_dev(self.step_dev, self.hp)
        return loss.detach()

    def _eager_part(self) -> None:
        if self.distributed:
            eng = self.eng
            eng.sync.finish()                           # fires every bucket's all-reduce now, waits
            eng.optim.step_dev(self.step_dev, self.hp)

    def __call__(self, batch: Dict[str, Tensor]) -> Tensor:
        for k, v in self.static_in.items():
            if isinstance(v, Tensor) and batch[k] is not v:
                v.copy_(batch[k], non_blocking=True)
        self.eng.flat.check()
        self._sync_optim()                              # host step count + current lr / betas / eps / weight_decay -> device (outside the graph)
        self.eng.sync.reset_host_state()                # (begin() ran at capture time only)
        self.graph.replay()
        if self.distributed:
            ops.grads_prezeroed = True                  # the replayed begin() zeroed the gradient buffer: finish() must not zero "unseen" slices
            self._eager_part()
        else:
            self.eng.optim.step_count += 1
            ops.weights_epoch += 1                      # parameter memory changed: host-side caches (weight scales) are stale for eager callers
        self.eng.optim._opt_called = True               # (torch's lr schedulers warn when they see no optimizer.step() before their own)
        return self.static_loss


class GraphedForward:
    """Forward pass (eval, no_grad) captured once into a HIP graph and replayed: the ~250 kernel launches of a forward
    make small batches host-bound (B = 32: 3 ms of Python/ctypes issue time), a replay is one launch.  Shapes are frozen
    at capture; `__call__` copies the new batch into the captured input buffers and returns the captured output dict
    (valid until the next call).  Every kernel of the library enqueues on torch's current stream without allocating or
    synchronising (include/qatiger.h conventions), so the capture itself needs nothing special.

    Round-1 history: replays of the whole-model graph ended in a GPU memory-access fault whenever a runtime blit ran between
    two replays, and the class demanded DEBUG_CLR_GRAPH_PACKET_CAPTURE=0.  The cause was in this library: its capturable
    entry points issued `hipMemsetAsync` (split-K output zeroing, bias-gradient and LayerNorm-gradient accumulators), which
    become runtime BLIT NODES inside the captured graph — the very class of operation the fault followed.  Every zeroing is
    now an ordinary kernel of the library (csrc/qt_common.h qt_zero_1d / qt_zero_2d); with that, `tools/graph_probe.py
    full_clone` replays the whole-model graph with a blit between replays at the runtime's DEFAULT settings
    (gpurun_out/r2_t14_graph_probe.log), and no environment variable is needed any more."""

    def __init__(self, model: nn.Module, example: Dict[str, Tensor], warmup: int = 2):
        if model.training:
            raise ValueError('GraphedForward captures the eval forward; call model.eval() first (train-mode dropout seeds '
                             'are host-side kernel arguments and would be frozen into the graph)')
        self.model = model
        self.static_in = {k: (v.clone() if isinstance(v, Tensor) else v) for k, v in example.items()}
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side), torch.no_grad():
            for _ in range(warmup):                     # lazy initialisation (library load, kernel attributes) outside the capture
                model(self.static_in)
        torch.cuda.current_stream().wait_stream(side)
        self.graph = torch.cuda.CUDAGraph()
        ops._amax_pool.cur = 0                          # (any pooled max|x| record the forward takes is zeroed inside the capture)
        # The weights' max|w| records and split images are REBUILT inside the capture, so a replay after an optimizer step, a
        # `load_state_dict` or any other in-place change of the parameters computes from the weights as they are then, like the eager
        # forward (cached images from the warm-up would freeze the model state of the capture into the graph).  FlatParams buffers
        # are persistent; images of other parameters are kept alive here for as long as the graph is.
        flats = {id(f): f for f in (getattr(p, '_qt_flat', None) for p in model.parameters()) if f is not None}
        for f in flats.values():
            f.amax_epoch = f.img_epoch = None
        self._keep: dict = {}
        ops.capture_keep = self._keep
        try:
            with torch.no_grad(), torch.cuda.graph(self.graph):
                self.static_out = model(self.static_in)
        finally:
            ops.capture_keep = None
        for f in flats.values():                        # the capture recorded the refresh, it did not run it
            f.amax_epoch = f.img_epoch = None
        ops._amax_pool.zero_all()

    def __call__(self, batch: Dict[str, Tensor]) -> Dict[str, Tensor]:
        for k, v in self.static_in.items():
            if isinstance(v, Tensor):
                v.copy_(batch[k], non_blocking=True)
        self.graph.replay()
        return self.static_out
