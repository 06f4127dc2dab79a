"""CPU oracle of the reference's hot path -- test infrastructure only (see forward.py)."""
