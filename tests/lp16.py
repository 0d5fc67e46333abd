"""The 16-bit storage type of the library build under test (torchreid._hip: AGRL_HIP_LP16 = fp16 (default) | bf16): the GPU
tests build their 16-bit operands and name the 16-bit precision mode through these two constants, so the same suite covers
either build (``AGRL_HIP_LP16=bf16 python -m pytest tests -m gpu`` runs it on libagrl_hip_bf16.so)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "agrl.pytorch_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)
from torchreid._hip import LP_DTYPE, LP_NAME  # noqa: E402

LP16 = LP_NAME

# Bars of the 16-bit mode against the fp32 CPU oracle (max |difference| / max |reference| of the embeddings): fp16 is held to the
# north star's 1e-3 like the fp32 modes; bf16 (8 significand bits) cannot meet it and keeps the looser bars it always had.
LP_EMBED_TOL = 1e-3 if LP_NAME == "fp16" else 5e-2
LP_STAGE_TOL = 1e-3 if LP_NAME == "fp16" else 1e-2
