"""qa-tiger_amd — MI355X (gfx950) native implementation of the QA-TIGER
question-aware Gaussian-expert temporal-fusion hot path.

The directory name is hyphenated (it is the name the build contract fixes), so
import it with `importlib.import_module('qa-tiger_amd')`; it also registers
itself as `qa_tiger_amd` for ordinary `import qa_tiger_amd` afterwards.

Layout: csrc/ (HIP kernels + C ABI, declared in include/qatiger.h), _lib.py
(ctypes binding), ops.py (autograd operators), src/models/ (the reference's
nn.Module surface), train.py (loss / fused Adam / data-parallel step), synth.py
(synthetic workload).
"""
import sys as _sys

from . import _lib, synth, ops                      # noqa: E402  (ops before the models)
from ._lib import QTError, build, load              # noqa: F401

_sys.modules.setdefault('qa_tiger_amd', _sys.modules[__name__])

from .src.models.net import QA_TIGER                # noqa: E402,F401
from .src.models.tspm import TSPM                    # noqa: E402,F401
from .src.models.modules import (                   # noqa: E402,F401
    Projection, AVQCrossAttn, QstGrounding, TempMoE, PatchSelecter)
from .src.models.encoders import CLIP_TEncoder      # noqa: E402,F401
from .train import CrossEntropyLoss, FusedAdam, FlatParams, DataParallelStep, GraphedForward, GraphedTrainStep   # noqa: E402,F401

__version__ = '0.1.0'
