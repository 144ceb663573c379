"""Importable name of the package that lives in ``real-time-video-quality-analysis_amd/``.

The product directory is named after the reference repository and contains
hyphens, which Python cannot import; this shim makes ``rtvqa_amd.<module>``
resolve to ``real-time-video-quality-analysis_amd/<module>.py``.
"""
import os as _os

_REAL = _os.path.join(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))),
                      "real-time-video-quality-analysis_amd")
__path__.insert(0, _REAL)

from . import _native  # noqa: E402,F401  (binding only; the .so loads on first Engine())
from .engine import DeviceFrames, Engine  # noqa: E402,F401
from . import complexity_metrics, frames, pooling, synth, video_processing  # noqa: E402,F401

__all__ = ["Engine", "DeviceFrames", "complexity_metrics", "video_processing", "pooling", "synth", "frames"]
