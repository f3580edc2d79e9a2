"""Import-compatible alias of the reference module name: a notebook that does
``from optic.models.modelsGPU import manakovSSF`` switches to
``from opticommpy_amd.modelsGPU import manakovSSF`` and nothing else changes."""
from .models import (  # noqa: F401
    checkGPU, convergenceCondition, edfa, gaussianComplexNoise, manakovDBP, manakovSSF, nlinPhaseRot,
    setPowerforParSSFM, ssfm,
)
