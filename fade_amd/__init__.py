"""fade_amd — MI355X-native `fade annotate` hot path (soft-clip re-alignment) behind a C ABI.

Layout: csrc/ (HIP kernels, C ABI, C++ host driver), _lib.py (ctypes binding), api.py (host-side
mirror of the reference interface), synth.py (deterministic synthetic BAM-shaped batches).
"""
from .api import Context, FadeHipError, Parasail, annotate_records, format_tags, stats_allreduce  # noqa: F401

__all__ = ["Context", "FadeHipError", "Parasail", "annotate_records", "format_tags", "stats_allreduce"]
