"""Shard -> pinned host -> HBM feeder throughput at the primary shapes (B=256, T=60, P=14, Dp=1024)."""
import importlib, os, sys, tempfile, time
import numpy as np, torch
sys.path.insert(0, '.')
fio = importlib.import_module('qa-tiger_amd.feature_io')
n, T = 512, 60
tmp = tempfile.mkdtemp(dir='/dev/shm' if os.path.isdir('/dev/shm') else None)
dirs = {k: os.path.join(tmp, k) for k in ('a', 'v', 'p')}
for d in dirs.values(): os.makedirs(d)
rng = np.random.default_rng(0)
names = [f'{i:08d}' for i in range(n)]
for nm in names:
    np.save(os.path.join(dirs['a'], nm + '.npy'), rng.standard_normal((T, 128), dtype=np.float32))
    np.save(os.path.join(dirs['v'], nm + '.npy'), rng.standard_normal((T, 768), dtype=np.float32))
    np.save(os.path.join(dirs['p'], nm + '.npy'), rng.standard_normal((T, 14, 1024), dtype=np.float32))
t = time.perf_counter()
for nm in names[:64]:                                   # the reference's per-sample path (page-cache warm)
    [np.load(os.path.join(dirs[k], nm + '.npy')) for k in 'avp']
per_file = (time.perf_counter() - t) / 64
print(f'per-sample np.load x3 (reference path, warm page cache): {per_file*1e3:.2f} ms/sample = {1/per_file:.0f} samples/s per worker')
for dtype in ('float32', 'float16'):
    path = os.path.join(tmp, f'f_{dtype}.qtfs')
    fio.pack_features(names, path, dirs['a'], dirs['v'], dirs['p'], dtype=dtype)
    shard = fio.FeatureShard(path)
    feeder = fio.DeviceFeeder(shard, 'cuda', 256)
    order = [names[i % n] for i in rng.permutation(256 * 12)]
    batches = [order[i:i + 256] for i in range(0, len(order), 256)]
    feeder.submit(batches[0]); feeder.submit(batches[1]); feeder.take(); feeder.submit(batches[2]); feeder.take()
    torch.cuda.synchronize(); t = time.perf_counter()
    for i in range(3, len(batches)):
        feeder.submit(batches[i]); feeder.take()
    feeder.take(); torch.cuda.synchronize()
    dt = (time.perf_counter() - t) / (len(batches) - 2)
    mb = sum(int(np.prod(s)) for s in shard.meta['shapes'].values()) * shard.dtype.itemsize * 256 / 1e6
    print(f'{dtype}: {dt*1e3:.1f} ms per batch of 256 = {256/dt:.0f} samples/s, {mb/dt/1e3:.1f} GB/s host-side bytes')
