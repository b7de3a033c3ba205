"""``backend/create_src_dict.py`` of the reference, same module path (ld_triangle.py:374, ld_area.py:306)."""
from ld_tools_amd.backend.create_src_dict import *  # noqa: F401,F403
from ld_tools_amd.backend.create_src_dict import create_src_dict  # noqa: F401
