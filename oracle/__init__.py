"""TEST INFRASTRUCTURE ONLY -- CPU oracle for the qed-splatter render hot path.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline``
leg may import this package.  The product path (``qed_splatter_amd``) never
imports it and fails loudly when the HIP library is missing.
"""
