"""ORACLE package: CPU restatements of the reference's hot path used as the parity checker.

Test infrastructure only.  Importers allowed: tests/, __graft_entry__.smoke(), bench.py (cpu_baseline leg).
The product package `uglad_amd` must never import from here.
"""
