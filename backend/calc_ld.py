"""``backend/calc_ld.py`` of the reference, same module path (backend/calc_ld.py:3): the GPU drop-in."""
from ld_tools_amd.backend.calc_ld import __version__, calc_ld, calc_ld_full  # noqa: F401
