"""``backend/prep_intgen_data.py`` of the reference, same module path (ld_triangle.py:372, ld_area.py:304, ld_lite.py:58), so
that the reference's scripts import unchanged.  The reference's function downloads the 1000 Genomes files, indexes them
and builds ``conversion.db`` (backend/prep_intgen_data.py:6-190: network, hours, one-time) -- out of scope here.  This one
only does the part every later run of the reference's function does: find the prepared folder's ``conversion.db`` and
return its path (prep_intgen_data.py:190), or say what is missing."""
import os

__version__ = "V3.0-ldx"


def prep_intgen_data(intgen_dir_path):
    intgen_convdb_path = os.path.join(intgen_dir_path, "conversion.db")      # prep_intgen_data.py:41-42
    if not os.path.exists(intgen_convdb_path):
        raise FileNotFoundError(
            f"{intgen_convdb_path} not found: the folder has to be prepared once with the reference's own "
            "backend/prep_intgen_data.py (1000 Genomes download, tabix indices, conversion.db); that step needs the "
            "network and is not part of the GPU path")
    return intgen_convdb_path
