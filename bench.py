#!/usr/bin/env python3
"""Benchmark of the QA-TIGER fusion hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W
    (N>1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...)

A step = the reference's `train()` body (src/trainutils.py:273-287) on one batch
of synthetic features: zero_grad -> QA_TIGER.forward (train mode, dropout on) ->
CrossEntropyLoss -> backward -> [gradient all-reduce over RCCL] -> Adam.step.
Workload = BASELINE.json configs[1]: B=256 per GPU, T=60, D=768, K=8 experts
(k=8, P=14 patches, L=77 words, Da=128, Dp=1024), fp32, inputs resident in HBM.
Weak scaling: every rank processes its own B=256 batch; the only exchange is the
gradient all-reduce.

Prints ONE JSON line (rank 0) with the contract fields plus
  roofline      dominant kernel family (projection GEMMs, MFMA-bound)
  attention     the attention-core kernels against the HBM roofline
  cpu_baseline  the CPU oracle timed on this host's cores (rank 0, N=1 only)
"""
from __future__ import annotations

import argparse
import importlib
import json
import os
import sys
import time

import torch
import torch.distributed as dist

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)

# reference-executed GEMM FLOPs per sample at B=256,T=60,D=768,K=8 (torch FlopCounterMode)
GFLOP_FWD, GFLOP_FWD_BWD = 12.048, 34.647
ATTN_MB_FWD = 21.6                       # attention-core algorithmic bytes per sample, fwd
PEAK_F32_MFMA_TFLOPS = 157.3             # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, dense
PEAK_16BIT_MFMA_TFLOPS = 2500.0          # MI355X_MICROARCH.md: dense bf16/fp16 MFMA (never the 2:1-sparse figure)
PEAK_HBM_GBS = 8000.0


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=10)
    ap.add_argument('--warmup', type=int, default=3)
    ap.add_argument('--batch', type=int, default=256, help='per-GPU batch (weak scaling)')
    ap.add_argument('--strong', action='store_true', help='shard a GLOBAL batch of --batch over the ranks instead')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--cpu-sample', type=int, default=16, help='batch of the bounded CPU-baseline sample')
    ap.add_argument('--backend', default='nccl', help='torch.distributed backend (nccl = RCCL over xGMI; gloo only for rehearsals)')
    ap.add_argument('--skip-fwd-only', action='store_true', help='skip the forward-only passes (used by tools/pmc_step_traffic.sh so that every profiled launch belongs to a train step)')
    ap.add_argument('--reduced-precision', action='store_true', help='also time the bf16x1 GEMM mode (BASELINE config 5 direction: one bf16 MFMA per product); reported beside the metric with its arg-max agreement, never as the metric')
    ap.add_argument('--tspm', action='store_true', help="also time the reference's second model (TSPM, SURVEY 8f row 4) train step at the same batch and feature shapes; reported beside the metric")
    ap.add_argument('--text-tower', action='store_true', help='also time the frozen CLIP text tower (ViT-L/14 text side, forward, B x 77 tokens); reported beside the metric, never in it')
    ap.add_argument('--profile-steps', type=int, default=2, help='extra instrumented steps for the roofline (after the timed region)')
    return ap.parse_args()


def pmc_traffic(family):
    """Average HBM-side bytes per launch from the committed PMC summary (tools/pmc_step_traffic.sh), or None."""
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'profiles', 'r01_pmc_step_traffic.json')
    try:
        with open(path) as f:
            return round(json.load(f)[family]['hbm_bytes_per_launch'])
    except (OSError, KeyError, ValueError):
        return None


def cpu_baseline(qt, sample_b: int):
    """CPU oracle (port of the reference's execution plan) on this host: fwd+bwd, train mode."""
    oracle = importlib.import_module('oracle.qatiger_oracle')
    synth = qt.synth
    cfg = synth.PathConfig(**{**synth.CONFIG_2.asdict(), 'B': sample_b})
    cores = min(torch.get_num_threads(), len(os.sched_getaffinity(0)))
    torch.set_num_threads(cores)
    P = synth.make_state_dict(cfg, 713)
    g = torch.Generator().manual_seed(713)
    inp = dict(audio=torch.randn(cfg.B, cfg.T, cfg.Da, generator=g), video=torch.randn(cfg.B, cfg.T, cfg.Dv, generator=g),
               patch=torch.randn(cfg.B, cfg.T, cfg.P, cfg.Dp, generator=g), quest_feat=torch.randn(cfg.B, cfg.Dv, generator=g),
               words_feat=torch.randn(cfg.B, cfg.L, cfg.Dv, generator=g), label=torch.randint(0, cfg.A, (cfg.B,), generator=g))
    small = {k: v[:2] for k, v in inp.items()}
    oracle.forward_backward(P, small, cfg.K, cfg.k, train=True)          # thread-pool / allocator warm-up
    times = []
    for _ in range(2):
        t0 = time.perf_counter()
        oracle.forward_backward(P, inp, cfg.K, cfg.k, train=True)
        times.append(time.perf_counter() - t0)
    t = min(times)
    return dict(value=round(cfg.B / t, 3), unit='QA-pairs/s', cores=cores, kind='port',
                sample=f'oracle/qatiger_oracle.py fwd+bwd+CE, train mode, B={cfg.B} (T=60,D=768,K=8), best of 2, {t:.2f} s/iter, '
                       f'torch {torch.__version__} CPU fp32')


def main():
    args = parse()
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local = int(os.environ.get('LOCAL_RANK', '0'))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit(f'--gpus {args.gpus} needs a torch.distributed.run launch with {args.gpus} ranks')
    os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    ndev = torch.cuda.device_count()
    local = local % max(ndev, 1)                             # rehearsal: several ranks may share one GPU (gloo)
    torch.cuda.set_device(local)
    dev = torch.device('cuda', local)
    if world > 1:
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        if args.backend == 'nccl':
            dist.init_process_group(backend='nccl', init_method='env://', device_id=dev)   # "nccl" is RCCL on ROCm
        else:
            dist.init_process_group(backend=args.backend, init_method='env://')
    qt = importlib.import_module('qa-tiger_amd')
    synth, ops = qt.synth, qt.ops

    per_gpu = args.batch // world if args.strong else args.batch
    cfg = synth.PathConfig(**{**synth.CONFIG_2.asdict(), 'B': per_gpu})
    torch.manual_seed(713)                                    # reference seed (configs/qa-tiger/vitl14.py:4)
    model = qt.QA_TIGER(**cfg.model_kwargs())                 # reference initialisers, identical on every rank
    model = model.to(dev).train()
    eng = qt.DataParallelStep(model, lr=1e-4, betas=(0.95, 0.999))
    batch = synth.fast_device_inputs(cfg, dev, seed=713 + rank)

    def sync():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        eng.step(batch)
    sync()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss = eng.step(batch)
    sync()
    dt = time.perf_counter() - t0
    if world > 1:
        tmax = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax)
    ms_step = dt / args.steps * 1e3
    value = per_gpu * world * args.steps / dt

    # ---- forward-only (eval, no_grad): the north-star's ">= 10k QA-pairs/s forward" figure
    n_fwd = max(3, args.steps)
    fwd_value = 0.0
    if not args.skip_fwd_only:
        model.eval()
        with torch.no_grad():
            for _ in range(2):
                model(batch)
            sync()
            t1 = time.perf_counter()
            for _ in range(n_fwd):
                model(batch)
            sync()
            fwd_dt = time.perf_counter() - t1
        model.train()
        fwd_value = per_gpu * world * n_fwd / fwd_dt

    # ---- instrumented steps: HIP events around every kernel launch of the two dominant families
    roofline = attention = None
    if args.profile_steps > 0:
        ops.timer = ops.KernelTimer()
        for _ in range(args.profile_steps):
            eng.step(batch)
        summ = ops.timer.summary()
        ops.timer = None
        n = args.profile_steps
        scale = cfg.B / 256.0                                 # algorithmic figures are per sample
        gm = summ['gemm']
        gemm_ms = gm['ms'] / n
        algo_tf = GFLOP_FWD_BWD * 1e9 * cfg.B / (gemm_ms * 1e-3) / 1e12
        exec_tf = gm['work'] / n / (gemm_ms * 1e-3) / 1e12
        top = sorted(gm['by_detail'].items(), key=lambda kv: -kv[1][0])[:4]
        split = ops.precision['fwd'] != 'fp32'
        peak = PEAK_16BIT_MFMA_TFLOPS if split else PEAK_F32_MFMA_TFLOPS
        roofline = dict(bound='mfma',
                        kernel=('qt_gemm_x3_kernel' if split else 'qt_gemm_f32_kernel') +
                               ' (all projection/MLP GEMM launches of one train step, fwd+bwd)',
                        achieved=round(algo_tf, 2), peak=peak, unit='TFLOP/s', frac=round(algo_tf / peak, 4),
                        traffic=pmc_traffic('gemm_x3') if split else None,
                        note=('achieved = reference-executed fp32 GEMM FLOPs (34.647 GFLOP/sample) / summed GEMM kernel time; '
                              'each fp32 product is issued as 3 16-bit MFMA products, so the matrix pipes see 3x executed_tflops'
                              if split else 'fp32 MFMA'),
                        algorithmic_gflop_per_sample=GFLOP_FWD_BWD, executed_tflops=round(exec_tf, 2),
                        mfma_issue_frac=round((3.0 if split else 1.0) * exec_tf / peak, 4),
                        algorithmic_bytes_per_launch=round(sum(4.0 * (k[0] * k[2] + k[2] * k[1] + k[0] * k[1]) * (k[6] if len(k) > 6 else 1) * v[2]
                                                               for k, v in gm['by_detail'].items()) / gm['launches']),
                        traffic_source='profiles/r01_pmc_step_traffic.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of `bench.py --steps 2 --warmup 1 --skip-fwd-only`; '
                                       'FETCH_SIZE x2 per the gfx950 correction; average HBM-side bytes per qt_gemm_x3 launch of a train step, not measured in this run)',
                        launches_per_step=gm['launches'] // n, avg_launch_ms=round(gemm_ms / (gm['launches'] / n), 4),
                        # the qt_gemm_x3_kernel launches alone (the family also counts the small exact-fp32 launches):
                        # this is the figure to compare with the rocprofv3 kernel_stats average of qt_gemm_x3_kernel
                        x3_launches_per_step=sum(v[2] for k, v in gm['by_detail'].items() if k[5] != 'fp32') // n,
                        x3_avg_launch_ms=round(sum(v[0] for k, v in gm['by_detail'].items() if k[5] != 'fp32') /
                                               max(1, sum(v[2] for k, v in gm['by_detail'].items() if k[5] != 'fp32')), 4),
                        gemm_ms_per_step=round(gemm_ms, 3), share_of_step=round(gemm_ms / ms_step, 3),
                        top_shapes=[dict(MNK_tAtB=list(k), ms_per_step=round(v[0] / n, 3), launches=v[2] // n,
                                         tflops=round(v[1] / (v[0] * 1e-3) / 1e12, 1)) for k, v in top])
        a_ms = (summ.get('attn_fwd', {'ms': 0})['ms'] + summ.get('attn_bwd', {'ms': 0})['ms']) / n
        a_bytes = (summ.get('attn_fwd', {'work': 0})['work'] + summ.get('attn_bwd', {'work': 0})['work']) / n
        if a_ms > 0:
            gbs = a_bytes / (a_ms * 1e-3) / 1e9
            attention = dict(bound='hbm', kernel='qt_attn_fwd_kernel + qt_attn_bwd_kernel (12 + 12 launches per step)',
                             achieved=round(gbs, 1), peak=PEAK_HBM_GBS, unit='GB/s', frac=round(gbs / PEAK_HBM_GBS, 4),
                             algorithmic_mb_per_sample=round(a_bytes / cfg.B / 1e6, 2), ms_per_step=round(a_ms, 3),
                             share_of_step=round(a_ms / ms_step, 3))

    reduced = None
    if args.reduced_precision and world == 1:
        keep = dict(ops.precision)
        model.eval()
        with torch.no_grad():
            ref_logits = model(batch)['out'].clone()
        ops.precision.update(fwd='bf16x1', bwd='bf16x1')
        try:
            with torch.no_grad():
                low_logits = model(batch)['out'].clone()
                for _ in range(2):
                    model(batch)
                sync()
                t3 = time.perf_counter()
                for _ in range(n_fwd):
                    model(batch)
                sync()
                low_fwd = per_gpu * n_fwd / (time.perf_counter() - t3)
            model.train()
            for _ in range(2):
                eng.step(batch)
            sync()
            t3 = time.perf_counter()
            for _ in range(args.steps):
                eng.step(batch)
            sync()
            low_dt = (time.perf_counter() - t3) / args.steps
        finally:
            ops.precision.update(keep)
            model.train()
        reduced = dict(mode='bf16x1 (operands rounded to bf16, one MFMA per product, fp32 accumulate)',
                       value=round(per_gpu / low_dt, 1), unit='QA-pairs/s', ms_per_step=round(low_dt * 1e3, 3),
                       fwd_only=round(low_fwd, 1),
                       argmax_agreement=round(float((ref_logits.argmax(-1) == low_logits.argmax(-1)).float().mean()), 4),
                       max_rel_logit_diff=round(float((ref_logits - low_logits).abs().max() / ref_logits.abs().max()), 5),
                       note='outside the 1e-3 parity gate; not the benchmarked metric')

    tspm = None
    if args.tspm and world == 1:
        tcfg = synth.TspmConfig(**{**synth.TSPM_REF.asdict(), 'B': per_gpu})
        torch.manual_seed(713)
        tmodel = qt.TSPM(**tcfg.model_kwargs()).to(dev).train()
        teng = qt.DataParallelStep(tmodel, lr=1e-4, betas=(0.95, 0.999))
        g = torch.Generator(device=dev).manual_seed(713)
        tb = dict(audio=torch.randn(per_gpu, tcfg.T, tcfg.Da, device=dev, generator=g), video=torch.randn(per_gpu, tcfg.T, tcfg.Dv, device=dev, generator=g),
                  patch=torch.randn(per_gpu, tcfg.T, tcfg.P, tcfg.Dp, device=dev, generator=g), quest=torch.randn(per_gpu, 1, tcfg.Dq, device=dev, generator=g),
                  prompt=torch.randn(per_gpu, 1, tcfg.Dq, device=dev, generator=g), label=torch.randint(0, 42, (per_gpu,), device=dev, generator=g))
        for _ in range(args.warmup):
            teng.step(tb)
        sync()
        t4 = time.perf_counter()
        for _ in range(args.steps):
            teng.step(tb)
        sync()
        tdt = (time.perf_counter() - t4) / args.steps
        tspm = dict(value=round(per_gpu / tdt, 1), unit='QA-pairs/s', ms_per_step=round(tdt * 1e3, 3), batch=per_gpu,
                    config=dict(T=tcfg.T, P=tcfg.P, hidden=tcfg.C, topK=tcfg.topK, audio_dim=tcfg.Da, vis_dim=tcfg.Dv, patch_dim=tcfg.Dp),
                    note='train step (fwd + CE + bwd + Adam, dropout on) of src/models/tspm.py on the HIP kernels; not the benchmarked metric')
        del teng, tmodel

    text_tower = None
    if args.text_tower and rank == 0:
        # on the real (non-synthetic) path this runs once per batch in front of the fusion model
        enc = importlib.import_module('qa-tiger_amd.src.models.encoders').CLIP_TEncoder('ViT-L/14@336px', tower='vitl14').to(dev).eval()
        tokens = synth.make_text_tokens(synth.TEXT_VITL14, per_gpu, 713).to(dev)
        for _ in range(2):
            enc(tokens)
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        n_txt = max(3, args.steps)
        for _ in range(n_txt):
            enc(tokens)
        torch.cuda.synchronize()
        txt_dt = (time.perf_counter() - t2) / n_txt
        tc = synth.TEXT_VITL14
        gflop = (tc.layers * (2 * tc.context_length * 12 * tc.width ** 2 + 4 * tc.context_length ** 2 * tc.width) + 2 * tc.width * tc.embed_dim) / 1e9
        text_tower = dict(value=round(per_gpu / txt_dt, 1), unit='questions/s', ms_per_batch=round(txt_dt * 1e3, 3), batch=per_gpu,
                          tokens=tc.context_length, gflop_per_question=round(gflop, 3),
                          algorithmic_tflops=round(per_gpu / txt_dt * gflop / 1e3, 1), weights='random (CLIP initialiser scales); no checkpoint offline')
        del enc

    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        cpu = cpu_baseline(qt, args.cpu_sample)

    if rank == 0:
        out = {
            'metric': 'QA-pairs/sec fwd+bwd at B=256 T=60 D=768 K=8',
            'value': round(value, 2), 'unit': 'QA-pairs/s', 'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
            'ms_per_step': round(ms_step, 3), 'higher_is_better': True, 'scaling': 'strong' if args.strong else 'weak',
            'vs_baseline': None, 'dtype': 'f32',
            'dtype_note': 'fp32 in, fp32 out, fp32 accumulate; each GEMM product is evaluated as 3 16-bit MFMA products on hi+lo-split fp32 operands (fp16 parts forward, bf16 parts backward): logits within 1e-6 of the exact-fp32 kernels (QT_GEMM=fp32)',
            'data': 'synthetic',
            'config': {'workload': 'BASELINE configs[1]: train step (fwd + CE + bwd + grad all-reduce + Adam, dropout on) on synthetic '
                                   'CLIP/VGGish/ToMe-shaped features, reference initialisers',
                       'per_gpu_batch': per_gpu, 'global_batch': per_gpu * world, 'T': cfg.T, 'D': cfg.D, 'video_dim': cfg.Dv,
                       'audio_dim': cfg.Da, 'patch_dim': cfg.Dp, 'P': cfg.P, 'L': cfg.L, 'experts': cfg.K, 'topK': cfg.k,
                       'parallelism': f'dp{world}', 'backend': ('rccl' if args.backend == 'nccl' else args.backend) if world > 1 else None},
            'fwd_only': {'value': round(fwd_value, 2), 'unit': 'QA-pairs/s', 'mode': 'eval, no_grad', 'target': 10000,
                         'algorithmic_tflops': round(fwd_value * GFLOP_FWD * 1e9 / 1e12 / world, 1)},
            'loss': round(float(loss), 5),
            'roofline': roofline, 'attention': attention, 'cpu_baseline': cpu,
        }
        if text_tower is not None:
            out['text_tower'] = text_tower
        if reduced is not None:
            out['reduced_precision'] = reduced
        if tspm is not None:
            out['tspm'] = tspm
        print(json.dumps(out))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
