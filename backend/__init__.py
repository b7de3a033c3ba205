"""The reference's package path for the LD hot path: ``from backend.calc_ld import calc_ld`` (ld_triangle.py:377,
ld_area.py:309, ld_lite.py:61) resolves here when the repository root is on ``sys.path``, so the three scripts need no
edit.  Everything lives in ``ld_tools_amd.backend``; this package only re-exports it."""
