"""ctypes binding of libqatiger_hip.so (declared in include/qatiger.h).

No CPU fallback exists: if the shared library is missing or a kernel call
fails, the caller gets an exception (SURVEY.md §8b "errors: plain Python
exceptions").
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
import sys
from typing import Optional

_HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(_HERE, 'csrc')
LIB_PATH = os.path.join(_HERE, 'libqatiger_hip.so')
SOURCES = ['gemm.hip', 'gemm_split.hip', 'gemm_skinny.hip', 'gemm_plane.hip', 'attn.hip', 'xattn.hip', 'norm.hip', 'moe.hip', 'misc.hip', 'text.hip', 'softmax.hip']
GFX = 'gfx950'
ABI_VERSION = 212          # must equal QT_ABI_VERSION in csrc/qt_common.h (returned by qt_version()); bump on ANY signature change
HASH_PATH = LIB_PATH + '.srchash'
W4 = bool(os.environ.get('QT_PS_WITH_W4'))      # also build gemm_plane.hip's four-wave layout (an experiment: csrc/gemm_plane.hip "NWN")
MF0 = bool(os.environ.get('QT_PS_WITH_MF0'))    # also build its round-2 MFMA shape (QT_PS_MF=0: A/B runs)
FLAGS = ['-O3', f'--offload-arch={GFX}', '-fPIC', '-std=c++17', '-Wno-unused-result'] + (['-DQT_PS_WITH_W4'] if W4 else []) + (['-DQT_PS_WITH_MF0'] if MF0 else [])


class QTError(RuntimeError):
    pass


def build(force: bool = False, verbose: bool = False) -> str:
    """Compile the HIP kernels for gfx950 into one in-tree shared library."""
    srcs = [os.path.join(CSRC, s) for s in SOURCES]
    want = source_hash()

    def fresh():
        # content hash, not mtimes: the tree is copied to the GPU box and file times do not survive that reliably
        try:
            with open(HASH_PATH) as f:
                return os.path.exists(LIB_PATH) and f.read().strip() == want
        except OSError:
            return False

    if not force and fresh():
        return LIB_PATH
    # several ranks of one node may get here together (torch.distributed.run): one builds, the others wait and re-check
    import fcntl
    lock = open(os.path.join(_HERE, '.build.lock'), 'w')
    fcntl.flock(lock, fcntl.LOCK_EX)
    try:
        if not force and fresh():
            return LIB_PATH
        return _build_locked(srcs, verbose, force)
    finally:
        fcntl.flock(lock, fcntl.LOCK_UN)
        lock.close()


def source_hash() -> str:
    """sha256 over the kernel sources, the shared header and the compiler flags."""
    import hashlib
    h = hashlib.sha256(' '.join(FLAGS).encode())
    for name in SOURCES + ['qt_common.h']:
        h.update(name.encode())
        with open(os.path.join(CSRC, name), 'rb') as f:
            h.update(f.read())
    return h.hexdigest()


def _build_locked(srcs, verbose: bool, force: bool = False) -> str:
    hipcc = os.environ.get('HIPCC', '/opt/rocm/bin/hipcc')
    flags = FLAGS
    objdir = os.path.join(CSRC, '_obj')
    os.makedirs(objdir, exist_ok=True)
    # one object per translation unit, compiled in parallel; gemm_split.hip and gemm_plane.hip are built in parts (see their tails)
    jobs = []
    for src in srcs:
        base = os.path.splitext(os.path.basename(src))[0]
        parts, macro = (range(7), 'QT_X3_PART') if base == 'gemm_split' else (([p for p in range(12) if (p not in (1, 2) or MF0) and (p < 10 or W4)], 'QT_PS_PART') if base == 'gemm_plane' else ([None], ''))
        for part in parts:
            obj = os.path.join(objdir, base + ('' if part is None else f'_p{part}') + '.o')
            jobs.append(([hipcc, *flags, '-c', src, '-o', obj] + ([] if part is None else [f'-D{macro}={part}']), obj))

    import hashlib
    with open(os.path.join(CSRC, 'qt_common.h'), 'rb') as f:
        common = f.read()

    def obj_hash(cmd):
        h = hashlib.sha256(' '.join(c for c in cmd if not c.endswith('.o')).encode())
        h.update(common)
        with open(cmd[cmd.index('-c') + 1], 'rb') as f:
            h.update(f.read())
        return h.hexdigest()

    def run(job):
        # one object per job, skipped when its source, the shared header and the command line are unchanged (force=True compiles all)
        cmd, obj = job
        want = obj_hash(cmd)
        if not force and os.path.exists(obj):
            try:
                with open(obj + '.hash') as f:
                    if f.read().strip() == want:
                        return subprocess.CompletedProcess(cmd, 0, '', '')
            except OSError:
                pass
        if verbose:
            print(' '.join(cmd), file=sys.stderr)
        res = subprocess.run(cmd, capture_output=True, text=True)
        if res.returncode == 0:
            with open(obj + '.hash', 'w') as f:
                f.write(want + '\n')
        return res

    from concurrent.futures import ThreadPoolExecutor
    with ThreadPoolExecutor(max_workers=max(1, min(8, os.cpu_count() or 1))) as pool:
        results = list(pool.map(run, jobs))
    for (cmd, _), res in zip(jobs, results):
        if res.returncode != 0:
            raise QTError(f'hipcc failed: {" ".join(cmd)}\n{res.stdout}\n{res.stderr}')
    link = [hipcc, f'--offload-arch={GFX}', '-fPIC', '-shared', '-o', LIB_PATH] + [obj for _, obj in jobs]
    res = subprocess.run(link, capture_output=True, text=True)
    if res.returncode != 0:
        raise QTError(f'hipcc link failed:\n{res.stdout}\n{res.stderr}')
    with open(HASH_PATH, 'w') as f:
        f.write(source_hash() + '\n')
    return LIB_PATH


_p = C.c_void_p
_i, _l, _f, _u64 = C.c_int, C.c_long, C.c_float, C.c_ulonglong

# name -> argtypes; every function returns int (0 = ok) except the two noted
SIGNATURES = {
    'qt_gemm_f32': [_p, _i, _i, _i, _i, _i, _f, _p, _l, _p, _l, _p, _l, _p, _p, _l, _i, _p, _l, _f, _f, _u64, _i, _i, _i, _l, _l, _l],
    'qt_gemm_f32_ws': [_p, _i, _i, _i, _i, _i, _f, _p, _l, _p, _l, _p, _l, _p, _p, _l, _i, _p, _l, _f, _f, _u64, _i, _i, _i, _l, _l, _l, _p, _l],
    'qt_gemm_x3': [_p, _i, _i, _i, _i, _i, _i, _f, _p, _l, _p, _l, _p, _l, _p, _p, _l, _i, _p, _l, _f, _f, _u64, _i, _i, _p, _i, _l, _l, _l, _p, _p, _p, _p, _l],
    'qt_gemm_ps': [_p, _i, _i, _i, _i, _i, _f, _p, _l, _p, _p, _l, _p, _p, _l, _p, _p, _l, _i, _p, _l, _f, _f, _u64, _i, _i, _p, _p, _l],
    'qt_gemm_ps_batched': [_p, _i, _i, _i, _i, _i, _f, _p, _l, _p, _p, _l, _p, _p, _l, _i, _l, _l, _l],
    'qt_gemm_ps_img': [_p, _i, _i, _i, _i, _i, _f, _p, _l, _p, _p, _l, _p, _p, _l, _p, _p, _l, _i, _p, _l, _f, _f, _u64, _i, _i, _p, _p, _l],
    'qt_gemm_f8': [_p, _i, _i, _i, _f, _p, _l, _p, _p, _l, _p, _p, _l, _p, _p, _l, _i, _p, _l, _f, _f, _u64, _i, _i],
    'qt_quant_fp8': [_p, _p, _l, _l, _i, _p, _p, _l],
    'qt_split_f16': [_p, _p, _l, _l, _i, _p, _p, _l, _p, _i],
    'qt_split_f16_segments': [_p, _p, _p, _p, _i, _p, _p],
    'qt_zero_f32': [_p, _p, _l],
    'qt_set_dropout_salt': [_p],
    'qt_bump_dropout_salt': [_p, _p, _u64],
    'qt_adam_step_dev': [_p, _p, _p, _p, _p, _l, _p, _p],
    'qt_headbias_fwd': [_p, _p, _l, _p, _p, _l, _i, _i, _p, _l],
    'qt_headbias_bwd': [_p, _p, _l, _p, _p, _l, _i, _i, _p, _p, _i],
    'qt_adam_set_dev': [_p, _p, _p, C.c_longlong, _f, _f, _f, _f, _f, _f],
    'qt_amax_f32': [_p, _p, _l, _l, _i, _p, _i],
    'qt_amax_segments': [_p, _p, _p, _p, _i, _p],
    'qt_xattn_fwd': [_p, _p, _l, _l, _p, _i, _i, _i, _i, _i, _f, _f, _u64, _p, _p, _p],
    'qt_xattn_bwd': [_p, _p, _l, _l, _p, _p, _p, _p, _p, _p, _i, _i, _i, _i, _i, _f, _f, _u64, _p, _p, _p, _p],
    'qt_attn_fwd': [_p, _p, _l, _l, _l, _p, _l, _p, _l, _l, _p, _l, _l, _l, _i, _i, _i, _i, _i, _i, _i, _f, _f, _u64, _p],
    'qt_attn_fwd_img': [_p, _p, _l, _l, _l, _p, _l, _p, _l, _l, _p, _l, _l, _l, _i, _i, _i, _i, _i, _i, _i, _f, _f, _u64, _p, _p, _l],
    'qt_attn_causal_fwd': [_p, _p, _l, _l, _l, _p, _l, _p, _l, _l, _p, _l, _l, _l, _i, _i, _i, _i, _f],
    'qt_text_embed_fwd': [_p, _p, _i, _i, _i, _i, _p, _p, _p],
    'qt_text_eot_rows': [_p, _p, _i, _i, _i, _p, _p, _p],
    'qt_softmax_fwd': [_p, _p, _l, _l, _i, _f, _f, _u64, _p, _l, _p, _l],
    'qt_softmax_bwd': [_p, _p, _l, _p, _l, _l, _i, _f, _f, _u64, _p, _l],
    'qt_attn_avg_weights': [_p, _p, _l, _l, _l, _p, _l, _l, _i, _i, _i, _i, _i, _i, _i, _f, _f, _u64, _p],
    'qt_cast_f16_f32': [_p, _p, _p, _l],
    'qt_eval_counts': [_p, _p, _l, _i, _i, _p, _p, _i, _p, _p, _p, _p],
    'qt_attn_bwd': [_p, _p, _l, _l, _l, _p, _l, _p, _l, _l, _p, _l, _l, _l, _p, _l, _l, _l, _p, _l, _p, _l,
                    _i, _i, _i, _i, _i, _i, _i, _f, _f, _u64, _p, _p, _p, _p],
    'qt_add_ln_fwd': [_p, _i, _i, _i, _p, _p, _p, _p, _p, _f, _p, _p, _f, _p, _l, _i, _p, _p],
    'qt_add_ln_bwd': [_p, _i, _i, _i, _p, _l, _p, _p, _p, _p, _p, _p, _f, _p, _p, _p, _p, _p, _p, _i],
    'qt_moe_router_fwd': [_p, _p, _p, _i, _i, _i, _i, _f, _p, _p, _p, _p],
    'qt_moe_coef_fwd': [_p, _p, _p, _p, _i, _i, _i, _i, _p, _p],
    'qt_moe_treduce_fwd': [_p, _p, _p, _i, _i, _i, _i, _i, _p],
    'qt_moe_treduce_bwd': [_p, _p, _p, _p, _i, _i, _i, _i, _i, _p, _p, _p],
    'qt_moe_router_bwd': [_p, _p, _p, _p, _p, _p, _p, _p, _i, _i, _i, _i, _f, _p, _p],
    'qt_colsum': [_p, _p, _l, _i, _i, _p, _i],
    'qt_add_rows': [_p, _p, _l, _i, _f, _p, _l, _i, _f, _p, _l, _l, _i],
    'qt_mask_grad': [_p, _p, _l, _p, _l, _f, _u64, _p, _l, _l, _i, _p],
    'qt_xent_fwd_bwd': [_p, _p, _l, _p, _i, _i, _p, _p, _p, _l],
    'qt_adam_step': [_p, _p, _p, _p, _p, _l, _i, _f, _f, _f, _f, _f, _f],
}
EXPORTS = ['qt_version', 'qt_last_error'] + list(SIGNATURES)

_lib: Optional[C.CDLL] = None


def load() -> C.CDLL:
    """Load the library; (re)build first when it is absent or was built from other sources (content hash) and hipcc
    is available.  A library whose ABI version differs from this binding's is refused: the ctypes signatures below
    would otherwise turn an argument-list change into silent memory corruption."""
    global _lib
    if _lib is not None:
        return _lib
    try:
        build()                                        # hash-checked and lock-protected: a no-op for a current library
    except Exception as e:  # noqa: BLE001
        if not os.path.exists(LIB_PATH):
            raise QTError(f'{LIB_PATH} is missing and could not be built ({e}); '
                          'run `python -c "import __graft_entry__ as g; g.build()"`') from e
        if str(e).startswith('hipcc failed') or str(e).startswith('hipcc link failed'):
            raise                                      # the sources changed and do not compile: never run the old library in their place
        # no compiler here: a stale library is caught by the version check below
    lib = C.CDLL(LIB_PATH)
    lib.qt_version.restype = C.c_int
    lib.qt_last_error.restype = C.c_char_p
    if lib.qt_version() != ABI_VERSION:
        raise QTError(f'{LIB_PATH} reports ABI {lib.qt_version()}, this binding needs {ABI_VERSION}: rebuild it '
                      '(`python -c "import __graft_entry__ as g; g.build()"`)')
    for name, args in SIGNATURES.items():
        fn = getattr(lib, name)
        fn.argtypes = args
        fn.restype = C.c_int
    _lib = lib
    return lib


def check(rc: int, what: str) -> None:
    if rc != 0:
        msg = load().qt_last_error().decode(errors='replace')
        raise QTError(f'{what} failed (code {rc}): {msg}')
