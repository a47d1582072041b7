"""MI355X-native sparse-GP inference core for edr-gp.

Drop-in for the GP path of neuro-ml/edr-gp: ``edrgp_amd.gp_model`` mirrors
``edrgp.gp_model`` (sklearn-style regressors), ``edrgp_amd.edr`` / ``base`` /
``utils`` / ``datasets`` mirror the EDR layer that calls it.  All GP arithmetic
runs in hand-written HIP kernels for gfx950 behind a C ABI
(``include/edrgp_hip.h``); there is no CPU fallback.
"""
__version__ = "0.1.0"

from .edr import EffectiveDimensionalityReduction  # noqa: E402,F401
from .base import BlockEDR  # noqa: E402,F401

__all__ = ["EffectiveDimensionalityReduction", "BlockEDR"]
