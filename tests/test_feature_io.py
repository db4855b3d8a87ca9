"""Feature shards: the packed format must reproduce what the reference's per-sample
`np.load` path reads (src/dataset.py:134-159), bit for bit in fp32.  CPU tests cover the format and the gather;
GPU tests the pinned-staging feeder and the fp16 widening kernel."""
import importlib
import os

import numpy as np
import pytest
import torch

fio = importlib.import_module('qa-tiger_amd.feature_io')


def make_dataset(root, n=7, T=12, Da=8, Dv=16, P=3, Dp=20, seed=0):
    rng = np.random.default_rng(seed)
    names = [f'{i:08d}' for i in rng.permutation(1000)[:n]]
    dirs = {k: os.path.join(root, k) for k in ('vggish', 'frame_ViT-L14@336px', 'visual_tome14')}
    for d in dirs.values():
        os.makedirs(d)
    for nm in names:
        np.save(os.path.join(dirs['vggish'], nm + '.npy'), rng.standard_normal((T, Da), dtype=np.float32))
        np.save(os.path.join(dirs['frame_ViT-L14@336px'], nm + '.npy'), rng.standard_normal((T, Dv), dtype=np.float32))
        np.save(os.path.join(dirs['visual_tome14'], nm + '.npy'), rng.standard_normal((T, P, Dp), dtype=np.float32))
    return names, dirs


def reference_load(dirs, name, sample_rate):
    """What AVQA_dataset.load_samples reads for one sample (dataset.py:134-159)."""
    return dict(audio=np.load(os.path.join(dirs['vggish'], name + '.npy')),
                video=np.load(os.path.join(dirs['frame_ViT-L14@336px'], name + '.npy'))[::sample_rate],
                patch=np.load(os.path.join(dirs['visual_tome14'], name + '.npy'))[::sample_rate])


@pytest.mark.parametrize('dtype', ['float32', 'float16'])
def test_shard_reproduces_per_file_loads(tmp_path, dtype):
    names, dirs = make_dataset(str(tmp_path))
    path = str(tmp_path / 'feats.qtfs')
    meta = fio.pack_features(names, path, dirs['vggish'], dirs['frame_ViT-L14@336px'], dirs['visual_tome14'], dtype=dtype)
    assert all(meta['offsets'][k] % fio.ALIGN == 0 for k in meta['offsets'])
    shard = fio.FeatureShard(path)
    assert len(shard) == len(names) and shard.names == names
    order = [names[3], names[0], names[3], names[6]]                   # repeats and arbitrary order, like a sampler
    got = shard.gather(order)
    for j, nm in enumerate(order):
        ref = reference_load(dirs, nm, 1)
        for k in ('audio', 'video', 'patch'):
            want = ref[k].astype(dtype)
            assert got[k][j].dtype == np.dtype(dtype) and np.array_equal(got[k][j], want)
    staging = {k: np.empty((5, *shard.meta['shapes'][k]), dtype=dtype) for k in shard.arrays}
    got2 = shard.gather(order, out=staging)
    assert all(np.array_equal(got2[k], got[k]) for k in got) and got2['audio'].base is staging['audio']
    with pytest.raises(KeyError):
        shard.gather(['not-a-video'])


def test_shard_sample_rate_and_optional_patch(tmp_path):
    names, dirs = make_dataset(str(tmp_path), T=10)
    path = str(tmp_path / 'f.qtfs')
    fio.pack_features(names, path, dirs['vggish'], dirs['frame_ViT-L14@336px'], None, sample_rate=3)
    shard = fio.FeatureShard(path)
    assert 'patch' not in shard.arrays and shard.meta['shapes']['video'] == [4, 16]
    got = shard.gather(names[:2])
    for j in range(2):
        assert np.array_equal(got['video'][j], np.load(os.path.join(dirs['frame_ViT-L14@336px'], names[j] + '.npy'))[::3])
    bad = str(tmp_path / 'bad'); open(bad, 'wb').write(b'nope')
    with pytest.raises(ValueError):
        fio.FeatureShard(bad)


def test_feeder_needs_a_gpu(tmp_path):
    if torch.cuda.is_available():
        pytest.skip('GPU present')
    names, dirs = make_dataset(str(tmp_path), n=2)
    path = str(tmp_path / 'f.qtfs')
    fio.pack_features(names, path, dirs['vggish'], dirs['frame_ViT-L14@336px'], dirs['visual_tome14'])
    with pytest.raises(importlib.import_module('qa-tiger_amd._lib').QTError):
        fio.DeviceFeeder(fio.FeatureShard(path), 'cpu', 2)


@pytest.mark.gpu
@pytest.mark.parametrize('dtype', ['float32', 'float16'])
def test_device_feeder_matches_get_items(tmp_path, dtype):
    names, dirs = make_dataset(str(tmp_path), n=9, T=6, Da=128, Dv=96, P=14, Dp=64, seed=3)
    path = str(tmp_path / 'f.qtfs')
    fio.pack_features(names, path, dirs['vggish'], dirs['frame_ViT-L14@336px'], dirs['visual_tome14'], dtype=dtype)
    feeder = fio.DeviceFeeder(fio.FeatureShard(path), 'cuda', max_batch=4)
    batches = [names[0:4], names[4:8], names[8:9], names[2:5]]
    feeder.submit(batches[0])
    for i, cur in enumerate(batches):
        if i + 1 < len(batches):
            feeder.submit(batches[i + 1])                               # next batch in flight while this one is used
        got = feeder.take()
        for k in ('audio', 'video', 'patch'):
            ref = np.stack([reference_load(dirs, nm, 1)[k] for nm in cur])
            want = torch.from_numpy(ref.astype(dtype)).float()          # get_items: .to(device).float()
            assert got[k].dtype == torch.float32 and got[k].shape == want.shape
            assert torch.equal(got[k].cpu(), want)
    with pytest.raises(importlib.import_module('qa-tiger_amd._lib').QTError):
        feeder.take()


@pytest.mark.gpu
def test_device_feeder_slot_reuse_waits_for_the_consumer(tmp_path):
    """The submit-next-then-take loop with NO host sync per step: while the GPU is still busy with batch i (a long
    spin kernel queued in front of the kernel that reads it), batch i+2 is submitted into the same slot.  The read of
    batch i must still see batch i (the slot's `free` event is recorded by that submit on the consumer's stream)."""
    names, dirs = make_dataset(str(tmp_path), n=8, T=6, Da=128, Dv=96, P=14, Dp=64, seed=5)
    path = str(tmp_path / 'f.qtfs')
    fio.pack_features(names, path, dirs['vggish'], dirs['frame_ViT-L14@336px'], dirs['visual_tome14'])
    feeder = fio.DeviceFeeder(fio.FeatureShard(path), 'cuda', max_batch=2, depth=2)
    batches = [names[0:2], names[2:4], names[4:6], names[6:8]]
    seen = []
    feeder.submit(batches[0])
    for i, cur in enumerate(batches):
        if i + 1 < len(batches):
            feeder.submit(batches[i + 1])
        got = feeder.take()
        torch.cuda._sleep(int(2.0e8))                                   # ~0.1 s of GPU time in front of the consumer kernel
        seen.append({k: v.clone() for k, v in got.items()})            # the "train step": reads the batch late, no host sync
    torch.cuda.synchronize()
    for cur, got in zip(batches, seen):
        for k in ('audio', 'video', 'patch'):
            want = torch.from_numpy(np.stack([reference_load(dirs, nm, 1)[k] for nm in cur])).float()
            assert torch.equal(got[k].cpu(), want), k


@pytest.mark.gpu
def test_cast_f16_f32_kernel_tail():
    lib = importlib.import_module('qa-tiger_amd._lib')
    for n in (1, 7, 8, 1000003):
        src = torch.randn(n).half().cuda()
        dst = torch.empty(n, dtype=torch.float32, device='cuda')
        lib.check(lib.load().qt_cast_f16_f32(torch.cuda.current_stream().cuda_stream, src.data_ptr(), dst.data_ptr(), n), 'cast')
        assert torch.equal(dst, src.float())
