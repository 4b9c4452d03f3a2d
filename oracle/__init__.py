"""CPU oracle of the query-localisation hot path -- TEST INFRASTRUCTURE ONLY.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import this
package.  PARITY UNPINNED: see the header of ``sfm_oracle.c``.
"""
