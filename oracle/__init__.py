"""CPU oracle for the GOF rasterization + cycle-aggregative projection hot path.

TEST INFRASTRUCTURE ONLY. Nothing under ``f3d-gaus_amd/`` may import this package; only ``tests/``,
``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg do, and only as the checker.
See the header of ``gof_oracle.c`` for what is and is not pinned against the reference.
"""
