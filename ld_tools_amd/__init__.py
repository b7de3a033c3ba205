"""ld_tools_amd -- MI355X (gfx950) engine for the pairwise-LD hot path of PlatonB/ld-tools.

Scope: ``backend/calc_ld.py`` as driven by ``ld_triangle`` / ``ld_area``.

    from ld_tools_amd.backend.calc_ld import calc_ld          # drop-in, same signature / dict
    from ld_tools_amd import PackedPanel, ld_triangle, ld_area

All arithmetic runs in libldx.so (HIP, C ABI in include/ldx.h); importing the package without
that library fails -- there is no CPU fallback.
"""
from ._lib import LdxError, version  # noqa: F401  (loads libldx.so or raises)
from .ops import AreaHits, TriangleResult, ld_area, ld_from_counts, ld_triangle, pair_counts  # noqa: F401
from .panel import PackedPanel, encode_codes  # noqa: F401

__all__ = ["PackedPanel", "encode_codes", "ld_triangle", "ld_area", "pair_counts", "ld_from_counts",
           "TriangleResult", "AreaHits", "LdxError", "version"]
