"""Drop-in for the reference's ``backend`` package on the LD hot path (backend/calc_ld.py)."""
