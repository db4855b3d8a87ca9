"""The reference's OWN caller loop around this repo's `QA_TIGER`, as a runnable driver (GPU box only).

Reproduces, with stock PyTorch objects only (nothing from qa-tiger_amd/train.py):

    get_model   (reference src/trainutils.py:82-113):  QA_TIGER(**cfg.hyper_params.model).to(device)
                -> load_state_dict(torch.load(weight), strict=False)
                -> DDP(model, device_ids=[rank], find_unused_parameters=False)  |  nn.DataParallel(model)
    get_optim   (trainutils.py:116-182):  optim.Adam(model.parameters(), lr, weight_decay, betas), StepLR(step_size 8, gamma 0.1)
    train body  (trainutils.py:273-287):  get_items dict (with its None entries) -> optimizer.zero_grad() -> model(dict)
                -> nn.CrossEntropyLoss()(output['out'], label) -> loss.backward() -> optimizer.step()
                -> gather_losses (trainutils.py:52-79: scalar all_reduce when a process group exists)
    checkpoint  (src/train.py:72-86):  model.module.state_dict() minus 'video_encoder' keys -> torch.save -> torch.load ->
                load_state_dict(strict=False) into a fresh model

in EVAL mode (dropout off: the CPU oracle cannot reproduce the device RNG) and compares every step's loss and the
final weights with the CPU oracle + `oracle.adam_step`.  The question goes in as int TOKENS through the frozen text tower
(`quest_encoder`), exactly as in the reference; the oracle gets the tower's outputs from oracle/clip_text_oracle.py.

    python tests/caller_loop_driver.py dp     # nn.DataParallel, one visible GPU (pass-through)
    python tests/caller_loop_driver.py ddp    # single-rank DistributedDataParallel over RCCL (backend 'nccl')

Prints one JSON line; exit code 0 = parity.
"""
import importlib
import io
import json
import os
import sys

import torch
import torch.distributed as dist
import torch.nn as nn
from torch.nn.parallel import DistributedDataParallel as DDP

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)


def main(wrap: str) -> int:
    os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    qt = importlib.import_module('qa-tiger_amd')
    oracle = importlib.import_module('oracle.qatiger_oracle')
    text_oracle = importlib.import_module('oracle.clip_text_oracle')
    synth = qt.synth
    device = torch.device('cuda', 0)
    torch.cuda.set_device(device)
    if wrap == 'ddp':                                                    # reference utils.py:82-87
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('MASTER_PORT', '29533')
        dist.init_process_group(backend='nccl', init_method='env://', rank=0, world_size=1)

    # words = the tower's hidden states [B,L,width], quest = EOT row @ text_projection [B,embed_dim]; QA_TIGER projects both
    # with Linear(video_dim, d_model) (net.py:122-123), so width == embed_dim == video_dim as in ViT-L/14 (768)
    tcfg = synth.TextConfig(width=128, layers=2, heads=2, context_length=77, vocab_size=512, embed_dim=128)
    cfg = synth.PathConfig(B=6, T=6, D=64, Dv=tcfg.embed_dim, Da=16, Dp=32, P=4, L=tcfg.context_length, K=4, k=4)
    seed = 21
    P = synth.make_state_dict(cfg, seed)
    raw = synth.make_inputs(cfg, seed)
    tokens = synth.make_text_tokens(tcfg, cfg.B, seed)

    # ---- get_model
    model = qt.QA_TIGER(**cfg.model_kwargs(), quest_dim=512, encoder_type='ViT-L/14@336px', text_tower=tcfg)
    model = model.to(device)
    buf = io.BytesIO()
    torch.save(P, buf)                                                    # the `weight` file of get_model: fusion weights only
    buf.seek(0)
    msg = model.load_state_dict(torch.load(buf), strict=False)
    assert not msg.unexpected_keys and all(k.startswith('quest_encoder.') for k in msg.missing_keys), msg
    if wrap == 'ddp':
        model = DDP(model, device_ids=[dist.get_rank()], find_unused_parameters=False)
    else:
        model = nn.DataParallel(model)
    TP = {n[len('quest_encoder.'):]: t.detach().cpu() for n, t in model.module.state_dict().items() if n.startswith('quest_encoder.')}

    # ---- get_optim
    optimizer = torch.optim.Adam(model.parameters(), lr=1e-4, weight_decay=0, betas=(0.95, 0.999))
    sched = torch.optim.lr_scheduler.StepLR(optimizer, step_size=8, gamma=0.1)
    criterion = nn.CrossEntropyLoss()

    # ---- get_items: tensors on the device, absent optional entries as None (DataParallel must scatter that dict)
    reshaped = dict(quest=tokens.to(device), audio=raw['audio'].to(device).float(), video=raw['video'].to(device).float(),
                    qtype_label=torch.zeros(cfg.B, dtype=torch.long, device=device), patch=raw['patch'].to(device).float(),
                    n_quest=None, n_video=None, n_audio=None, n_patch=None, prompt=None,
                    label=raw['label'].to(device).long().reshape(-1))

    # ---- oracle side
    with torch.no_grad():
        qf, wf = text_oracle.clip_text_forward(TP, tokens, tcfg.heads)
    oin = dict(audio=raw['audio'], video=raw['video'], patch=raw['patch'], quest_feat=qf, words_feat=wf, label=raw['label'])
    ref = {n: t.clone() for n, t in P.items()}
    m = {n: torch.zeros_like(t) for n, t in P.items()}
    v = {n: torch.zeros_like(t) for n, t in P.items()}

    model.eval()                                                         # train() body, dropout off (see the module docstring)
    report = dict(wrap=wrap, steps=[])
    ok = True
    for step in range(1, 4):
        optimizer.zero_grad()
        output = model(reshaped)
        loss = 0
        ce_loss = criterion(output['out'], reshaped['label'])
        loss += ce_loss
        for key in output:
            if 'loss' in key:
                loss += output[key]
        loss.backward()
        optimizer.step()
        if dist.is_initialized():                                        # gather_losses, trainutils.py:60-67
            gv = torch.tensor(float(loss)).to(device)
            dist.all_reduce(gv, op=dist.ReduceOp.SUM)
            shown = gv.item() / dist.get_world_size()
        else:
            shown = float(loss)
        loss_r, _, grads = oracle.forward_backward(ref, oin, cfg.K, cfg.k)
        if step == 1:
            solid = {n: g.abs() > 1e-3 * g.abs().max() for n, g in grads.items()}
        oracle.adam_step(ref, grads, m, v, step)
        rel = abs(shown - float(loss_r)) / abs(float(loss_r))
        report['steps'].append(dict(step=step, loss=shown, oracle_loss=float(loss_r), rel=rel))
        ok = ok and rel < 2e-5
    sched.step()                                                         # train.py: once per epoch
    report['lr_after_sched'] = optimizer.param_groups[0]['lr']

    # ---- final weights: compare the UPDATE (each weight moved ~3e-4), on elements whose gradient is not rounding noise
    worst = 0.0
    sd = model.module.state_dict()
    for n in P:
        upd, upd_r = sd[n].detach().cpu() - P[n], ref[n] - P[n]
        d = (upd - upd_r).abs()[solid[n]]
        if d.numel():
            worst = max(worst, float(d.max()))
    report['max_update_diff'] = worst
    ok = ok and worst < 2e-5

    # ---- best.pt round trip (train.py:72-86 save; trainutils.py:94-101 load with strict=False)
    new_sd = {k: t for k, t in sd.items() if 'video_encoder' not in k}
    keys = sorted(new_sd)
    want = sorted(list(synth.param_shapes(cfg)) + ['quest_encoder.' + k for k in synth.text_param_shapes(tcfg)])
    report['best_pt_keys'] = len(keys)
    ok = ok and keys == want
    buf = io.BytesIO()
    torch.save(new_sd, buf)
    buf.seek(0)
    fresh = qt.QA_TIGER(**cfg.model_kwargs(), text_tower=tcfg).to(device)
    msg = fresh.load_state_dict(torch.load(buf), strict=False)
    ok = ok and not msg.missing_keys and not msg.unexpected_keys
    fresh.eval()
    with torch.no_grad():
        a = model(reshaped)['out']
        b = fresh(reshaped)['out']
    report['reload_max_logit_diff'] = float((a - b).abs().max())
    ok = ok and report['reload_max_logit_diff'] < 1e-5
    report['ok'] = bool(ok)
    print(json.dumps(report))
    if dist.is_initialized():
        dist.barrier()
        dist.destroy_process_group()
    return 0 if ok else 1


if __name__ == '__main__':
    sys.exit(main(sys.argv[1] if len(sys.argv) > 1 else 'dp'))
