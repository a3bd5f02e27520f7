"""MI355X-native drop-in for NR-NeRF's per-ray hot path (render_rays / batchify_rays).

The compute lives in ``lib/libnrnerf_hip.so`` (hand-written HIP for gfx950 behind
the C ABI of ``include/nrnerf.h``); this package is the thin Python boundary that
mirrors the reference's ``batchify_rays`` / ``render_rays`` call contract
(train.py:108-137, 792-980).
"""
__version__ = "0.1.0"
