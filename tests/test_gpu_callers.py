"""The callers either side of the hot path, run as the reference runs them (GPU box only; each case is a child process so
that process groups and wrapped models never leak into the test runner):

* the reference's own train loop objects around this repo's `QA_TIGER` — `nn.DataParallel` / single-rank
  `DistributedDataParallel`, stock `optim.Adam` + `StepLR`, stock `nn.CrossEntropyLoss`, the `best.pt` round trip
  (tests/caller_loop_driver.py; reference src/trainutils.py:82-113, 116-182, 273-287, src/train.py:72-86);
* the repo's data-parallel step over RCCL (`backend='nccl'`), world size 1 on the one GPU of the test box and world
  size 2 where two GPUs are visible (tests/rccl_driver.py; reference src/utils.py:82-87, src/trainutils.py:47-79, 103-106).
"""
import json
import os
import socket
import subprocess
import sys

import pytest
import torch

from conftest import REPO

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _run(args, env=None, timeout=600):
    e = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY='0')
    e.update(env or {})
    res = subprocess.run([sys.executable] + args, cwd=REPO, env=e, capture_output=True, text=True, timeout=timeout)
    lines = [ln for ln in res.stdout.strip().splitlines() if ln.startswith('{')]
    rep = json.loads(lines[-1]) if lines else None
    return res, rep


@pytest.mark.parametrize('wrap', ['dp', 'ddp'])
def test_reference_caller_loop(wrap):
    res, rep = _run([os.path.join(REPO, 'tests', 'caller_loop_driver.py'), wrap], env=dict(MASTER_PORT=str(_free_port())))
    assert res.returncode == 0 and rep and rep['ok'], (res.stdout[-3000:], res.stderr[-3000:])
    assert len(rep['steps']) == 3 and all(s['rel'] < 2e-5 for s in rep['steps'])
    assert rep['lr_after_sched'] == pytest.approx(1e-4)            # StepLR(8, 0.1) after one epoch: unchanged; it ran against Adam
    print(rep)


def test_data_parallel_step_over_rccl_world1():
    res, rep = _run([os.path.join(REPO, 'tests', 'rccl_driver.py'), '0', '1', str(_free_port())])
    assert res.returncode == 0 and rep and rep['ok'], (res.stdout[-3000:], res.stderr[-3000:])
    assert rep['buckets_fired_during_backward'] >= rep['buckets'] - 1 and rep['first_fire_at'] < rep['last_grad_at']
    print(rep)


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason='needs two GPUs (the round-end box has one)')
def test_data_parallel_step_over_rccl_world2():
    port = str(_free_port())
    e = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY='0')
    procs = [subprocess.Popen([sys.executable, os.path.join(REPO, 'tests', 'rccl_driver.py'), str(r), '2', port], cwd=REPO, env=e,
                              stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True) for r in range(2)]
    outs = [p.communicate(timeout=600) for p in procs]
    assert all(p.returncode == 0 for p in procs), outs
    rep = json.loads([ln for ln in outs[0][0].strip().splitlines() if ln.startswith('{')][-1])
    assert rep['ok'] and rep['replica_weight_diff'] == 0.0


def test_fused_adam_is_a_torch_optimizer(qt, synth):
    """The repo's FusedAdam behind the reference's scheduler objects (trainutils.py:147-182): param_groups drive the kernel,
    weight decay follows torch.optim.Adam, state_dict round-trips."""
    torch.manual_seed(0)
    net = torch.nn.Sequential(torch.nn.Linear(8, 16), torch.nn.Linear(16, 4)).cuda()
    ref = torch.nn.Sequential(torch.nn.Linear(8, 16), torch.nn.Linear(16, 4)).cuda()
    ref.load_state_dict(net.state_dict())
    flat = qt.FlatParams(net.parameters())
    opt = qt.FusedAdam(flat, lr=1e-2, betas=(0.95, 0.999), weight_decay=0.1)
    ropt = torch.optim.Adam(ref.parameters(), lr=1e-2, betas=(0.95, 0.999), weight_decay=0.1)
    assert isinstance(opt, torch.optim.Optimizer) and len(opt.param_groups) == 1
    sch, rsch = torch.optim.lr_scheduler.StepLR(opt, 2, 0.1), torch.optim.lr_scheduler.StepLR(ropt, 2, 0.1)
    x = torch.randn(5, 8, device='cuda')
    for it in range(5):
        for o, n in ((opt, net), (ropt, ref)):
            o.zero_grad()
            n(x).square().sum().backward()
        for p, gv in zip(flat.params, flat.grad_views):
            gv.copy_(p.grad)
        opt.step(); ropt.step(); sch.step(); rsch.step()
        assert opt.param_groups[0]['lr'] == pytest.approx(ropt.param_groups[0]['lr'])
        if it == 2:                                                     # checkpoint / restore in the middle
            sd = opt.state_dict()
            opt2 = qt.FusedAdam(flat, lr=1.0)
            opt2.load_state_dict(sd)
            assert opt2.step_count == 3 and opt2.param_groups[0]['weight_decay'] == 0.1
            assert torch.equal(opt2.exp_avg, opt.exp_avg) and torch.equal(opt2.exp_avg_sq, opt.exp_avg_sq)
    for p, q in zip(net.parameters(), ref.parameters()):
        assert float((p - q).abs().max()) < 1e-6
    # a model moved after the flat buffer was built must fail loudly, not train a stale copy
    net.double()
    with pytest.raises(RuntimeError, match='flat buffer'):
        flat.check()


def test_integration_md_binding_stubs_run_as_written():
    """The ctypes stubs of INTEGRATION.md section 3 (`linear` on qt_gemm_x3, `image` + `linear_ps` on qt_split_f16 / qt_gemm_ps),
    executed verbatim from the document against libqatiger_hip.so with nothing of this repo's Python in between: a maintainer's
    binding of the C ABI gives torch's result."""
    import re
    text = open(os.path.join(REPO, 'INTEGRATION.md')).read()
    block = re.search(r"```python\nimport ctypes as C, torch\n(.*?)```", text, re.S)
    assert block, 'INTEGRATION.md: the binding stub moved'
    src = 'import ctypes as C, torch\n' + block.group(1).replace("C.CDLL('qa-tiger_amd/libqatiger_hip.so')", f"C.CDLL({os.path.join(REPO, 'qa-tiger_amd', 'libqatiger_hip.so')!r})")
    import importlib
    importlib.import_module('qa-tiger_amd._lib').build()            # (the library itself; the stub binds it on its own)
    ns = {}
    exec(compile(src, 'INTEGRATION.md', 'exec'), ns)
    g = torch.Generator().manual_seed(3)
    x, w, b = torch.randn(2304, 768, generator=g).cuda(), torch.randn(512, 768, generator=g).cuda(), torch.randn(512, generator=g).cuda()
    ref = (x.double() @ w.double().t() + b.double())
    y = ns['linear'](x, w, b)
    assert float((y.double() - ref).abs().max() / ref.abs().max()) < 2e-6
    y2 = ns['linear_ps'](ns['image'](x), ns['image'](w), b, 2304, 512, 768)
    assert float((y2.double() - ref).abs().max() / ref.abs().max()) < 2e-6
