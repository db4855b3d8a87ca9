"""world_size-2 test of the data-parallel gradient exchange (FlatParams + GradSync) on CPU
with the gloo backend.  The exchange is backend-agnostic host logic over torch.distributed;
on the GPU box the same code runs with backend "nccl" (= RCCL over xGMI)."""
import importlib
import os
import socket
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import REPO


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, bucket_bytes, q):
    try:
        sys.path.insert(0, REPO)
        os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
        dist.init_process_group('gloo', rank=rank, world_size=world)
        qt = importlib.import_module('qa-tiger_amd')
        train = importlib.import_module('qa-tiger_amd.train')
        torch.manual_seed(0)                                   # identical replicas
        net = torch.nn.Sequential(torch.nn.Linear(6, 16), torch.nn.ReLU(), torch.nn.Linear(16, 16), torch.nn.ReLU(),
                                  torch.nn.Linear(16, 4))
        unused = torch.nn.Parameter(torch.ones(5))              # never reached by the loss
        params = list(net.parameters()) + [unused]
        flat = qt.FlatParams(params)
        sync = train.GradSync(flat, bucket_bytes=bucket_bytes)
        assert sync.world == world and len(sync.buckets) >= (2 if bucket_bytes < 512 else 1)
        g = torch.Generator().manual_seed(100)
        x_all, y_all = torch.randn(8, 6, generator=g), torch.randn(8, 4, generator=g)
        shard = train.shard_batch(dict(x=x_all, y=y_all), rank, world)
        out = []
        for step in range(2):                                    # twice: begin() must reset state
            sync.begin()
            loss = ((net(shard['x']) - shard['y']) ** 2).sum()
            loss.backward()
            sync.finish()
            out.append(flat.grad.clone())
        # reference: sum over ranks of the local gradients == gradient of the summed loss on the global batch
        ref_net = torch.nn.Sequential(torch.nn.Linear(6, 16), torch.nn.ReLU(), torch.nn.Linear(16, 16), torch.nn.ReLU(),
                                      torch.nn.Linear(16, 4))
        ref_net.load_state_dict(net.state_dict())
        ((ref_net(x_all) - y_all) ** 2).sum().backward()
        al = train._ALIGN                                       # every parameter starts on an aligned offset of the flat buffer; padding stays zero
        ref = torch.cat([torch.nn.functional.pad(p.grad.reshape(-1), (0, (-p.numel()) % al)) for p in ref_net.parameters()]
                        + [torch.zeros(5 + (-5) % al)])
        ok = all(torch.allclose(o, ref, atol=1e-5, rtol=1e-5) for o in out)
        ok = ok and all(p.grad is v for p, v in zip(params[:-1], flat.grad_views))     # grads live in the flat buffer
        q.put((rank, ok, float((out[-1] - ref).abs().max())))
        dist.destroy_process_group()
    except Exception as e:  # noqa: BLE001
        q.put((rank, False, repr(e)))


@pytest.mark.parametrize('bucket_bytes', [64, 1 << 20])
def test_bucketed_grad_allreduce_world2(bucket_bytes):
    world = 2
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, bucket_bytes, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
    assert all(ok for _, ok, _ in results), results
