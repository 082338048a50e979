"""ORACLE -- test infrastructure only (CPU restatements of the reference's algorithms).

Importable from ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s cpu_baseline leg only.
"""
