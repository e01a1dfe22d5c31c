"""CPU oracle for the patch-by-patch texture-GAN hot path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``infinite_texture_gans_amd/`` may
import this package; only ``tests/``, ``__graft_entry__.smoke()`` and the
``cpu_baseline`` leg of ``bench.py`` use it, and there only as the checker /
timed CPU baseline, never as the thing shipped.

What it is: a restatement, on plain PyTorch-CPU fp32 ops, of the arithmetic
the reference performs on the path ``train.py:122-180`` (G+D train step) and
``utils.py:258-397`` (patch-grid inference).  All numerics of the reference
live in PyTorch itself (Conv2d, BatchNorm2d, F.pad, bmm/softmax, Adam ...,
SURVEY.md section 8c); the oracle therefore calls the same torch CPU
primitives but restates every piece of reference-owned logic (patch merge /
crop / local padding, block wiring, samplers, train-step control flow,
spectral-norm power iteration, Adam, EMA) independently, as functions over a
``state_dict``.

Parity pin: the reference ships no tests or golden vectors (SURVEY.md
section 4), so the oracle is pinned by fixtures generated in the build
container by importing the reference itself (``tests/golden/make_golden.py``,
outputs committed under ``tests/golden/``) and checked by
``tests/test_oracle_vs_golden.py``.  Hinge loss has no reference counterpart
(the ``--loss`` flag is never read, ``utils.py:85``) and is therefore
"parity unpinned".
"""
from . import patches, nets, step  # noqa: F401
