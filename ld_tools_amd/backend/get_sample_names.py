"""Sample-panel selection (reference: backend/get_sample_names.py:5-21): the names of the 1000 Genomes samples of
the requested genders and populations, in the order the ``samples`` table returns them -- that order defines the
haplotype columns (index 2*s + phase) of every genotype list.  Same function name and arguments; the values are
bound as SQL parameters instead of being pasted into the statement."""
from __future__ import annotations

import sqlite3


def get_sample_names(gend_names, pop_names, intgen_convdb_path):
    gend_names, pop_names = tuple(gend_names), tuple(pop_names)
    query = f"SELECT sample FROM samples WHERE gender IN ({','.join('?' * len(gend_names))})"
    params = list(gend_names)
    if pop_names != ("ALL",):
        marks = ",".join("?" * len(pop_names))
        query += f" AND (super_pop IN ({marks}) OR pop IN ({marks}))"
        params += list(pop_names) * 2
    with sqlite3.connect(intgen_convdb_path) as conn:
        cursor = conn.cursor()
        sample_names = [row[0] for row in cursor.execute(query, params)]
        cursor.close()
    return sample_names
