"""Callers either side of the hot path: genotype ingest from pysam-style records, the
ld_triangle / ld_area / ld_lite drivers and their text writers.

The reference re-fetches both VCF records and rebuilds both 5008-entry genotype lists for EVERY pair
(ld_triangle.py:158-186, ld_area.py:215-235).  These drivers read each variant once, pack the panel once
(PackedPanel) and replace the pair loops by one batched kernel launch; what they write is byte-for-byte what the
reference writes (tests/test_drivers.py composes the expected text from the oracle's calc_ld in the reference's
loop order).  Any object with pysam's ``VariantFile.fetch(chrom, start, end)`` / ``VariantRecord`` attributes works
as the VCF source; pysam itself is only imported by the command-line shells.
"""
from .area import AreaQueryResult, area_scan, get_inld_vars, write_area_file  # noqa: F401
from .ingest import RaggedGenotypesError, codes_matrix, find_record, sample_genotypes  # noqa: F401
from .lite import DifChrsError, NotInIntgenConvDbError, NotRsIdError, check_rs_id, ld_lite_table  # noqa: F401
from .triangle import (TriangleMatrix, create_matrix, stream_triangle_table, triangle_matrix,  # noqa: F401
                       write_triangle_table)
