"""said_amd — MI355X-native (gfx950) implementation of SAiD's inference hot path.

The package mirrors the reference's ``said.model`` class surface (``SAID``,
``SAID_UNet1D``, ``UNet1DConditionModel``, ``ModifiedWav2Vec2Model``) for the
denoising path only (reference: said/model/diffusion.py:308-472).  All device
math runs in hand-written HIP kernels behind the C ABI declared in
``include/said_hip.h``; there is no CPU fallback.
"""

__version__ = "0.1.0"
