"""opticommpy_amd -- MI355X-native split-step Fourier fiber propagation.

Drop-in for the fiber-channel functions of OptiCommPy's ``optic.models.modelsGPU``
(``ssfm``, ``manakovSSF``, ``manakovDBP``, ``edfa``, ``setPowerforParSSFM``) plus
``checkGPU``: numpy in, numpy out, same parameters-object API.  Host code is plain
Python + numpy calling hand-written HIP kernels (gfx950) through the C ABI declared
in ``include/ssf.h`` (``libssf_hip.so``).  There is no CPU fallback: without the
library or without a GPU every propagation call raises.
"""
from .utils import parameters  # noqa: F401
from .models import (  # noqa: F401
    blockwiseFFTConv, checkGPU, convergenceCondition, edc, edfa, linearFiberChannel, manakovDBP, manakovSSF, nlinPhaseRot,
    setPowerforParSSFM, ssfm,
    last_run, set_device, set_engine,
)

from .device import DeviceArray, to_device  # noqa: F401
from .rx import (  # noqa: F401
    balancedPD, coherentReceiver, decimate, delaySignal, firFilter, iqMixing, lowPassFIR, opticalHybrid2x4, pbs,
    pdmCoherentReceiver, pdmCoherentReceiverChain, photodiode,
)

from .wdm_tx import basicLaserModel, grayMapping, phaseNoise, pulseShape, simpleWDMTx  # noqa: F401

__all__ = ["simpleWDMTx", "basicLaserModel", "pulseShape", "phaseNoise", "grayMapping", "DeviceArray", "to_device", "firFilter", "lowPassFIR", "decimate", "delaySignal", "iqMixing", "pbs", "photodiode", "balancedPD",
           "opticalHybrid2x4", "coherentReceiver", "pdmCoherentReceiver", "pdmCoherentReceiverChain", "parameters", "ssfm", "manakovSSF", "manakovDBP", "nlinPhaseRot", "convergenceCondition", "edfa", "edc", "blockwiseFFTConv", "linearFiberChannel",
           "setPowerforParSSFM", "checkGPU", "last_run", "set_device", "set_engine"]
