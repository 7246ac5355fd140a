"""Mirror of the reference's `uglad.glad` package: glad.py (the unrolled cell) and glad_params.py (its 42 parameters)."""
