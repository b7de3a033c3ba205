"""Source-table reader (reference: backend/create_src_dict.py:5-47): the first ``rs<digits>`` of every line after
the meta lines, looked up in the ``variants`` table -> {chrom: [[pos, rsID], ...]}.  Same name and arguments."""
from __future__ import annotations

import os
import re
import sqlite3


def create_src_dict(src_dir_path, src_file_name, meta_lines_quan, intgen_convdb_path):
    with open(os.path.join(src_dir_path, src_file_name)) as src_file_opened:
        for _ in range(meta_lines_quan):
            src_file_opened.readline()
        rs_ids = set()
        for line in src_file_opened:
            found = re.search(r"rs\d+\b", line)
            if found is not None:
                rs_ids.add(found.group())
    if not rs_ids:
        return {}
    rs_ids = tuple(rs_ids)
    data_by_chrs = {}
    with sqlite3.connect(intgen_convdb_path) as conn:
        cursor = conn.cursor()
        for lo in range(0, len(rs_ids), 500):          # SQLite limits the number of bound parameters
            part = rs_ids[lo:lo + 500]
            query = f"SELECT * FROM variants WHERE ID IN ({','.join('?' * len(part))})"
            for chrom, pos, rs_id in cursor.execute(query, part):
                data_by_chrs.setdefault(chrom, []).append([pos, rs_id])
        cursor.close()
    return data_by_chrs
