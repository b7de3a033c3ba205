"""``backend/get_sample_names.py`` of the reference, same module path (ld_triangle.py:373, ld_area.py:305, ld_lite.py:59)."""
from ld_tools_amd.backend.get_sample_names import *  # noqa: F401,F403
from ld_tools_amd.backend.get_sample_names import get_sample_names  # noqa: F401
