"""CPU oracle — TEST INFRASTRUCTURE ONLY.

A plain PyTorch-CPU fp32 restatement of the reference's inference hot path
(`SAID.inference`, /root/reference/said/model/diffusion.py:308-472) used as the
checker for the HIP path.  Only ``tests/``, ``__graft_entry__.smoke()`` and
``bench.py``'s ``cpu_baseline`` leg may import it; nothing under ``said_amd/``
does, and the product path raises if the HIP library is missing.

Pinning status:
  * ``oracle.unet`` / ``oracle.wav2vec2`` / ``oracle.pipeline.process_audio`` /
    ``fit_audio_unet`` / CSV I/O are pinned against the reference's own modules,
    imported in the build container by ``tests/golden/make_golden.py``; the
    resulting vectors are committed under ``tests/golden/``.
  * ``oracle.scheduler`` — **parity unpinned**.  The algorithm lives in the
    third-party ``diffusers==0.19.*`` (``DDIMScheduler``, ``rescale_noise_cfg``;
    reference pyproject.toml:16), which is absent from /root/reference, not
    installed here and cannot be installed (no network); the reference holds no
    test or golden vector for it.  The restatement follows the published
    algorithm and the reference's call sites (diffusion.py:100-104, 361, 370,
    378, 441-443, 451-454) and is checked only by self-consistency properties.
"""
