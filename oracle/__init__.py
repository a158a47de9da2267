"""CPU oracle for the MDViT forward/backward path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``mdvit_amd/`` may import this package:
only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline``
leg use it, and only as the checker / the timed CPU baseline -- never as the
thing measured or shipped.

Parity status: the reference (siyi-wind/MDViT) ships no golden vectors or tests
(SURVEY.md section 4).  This oracle is pinned against outputs of the reference
itself, imported read-only in the build container by ``oracle/gen_golden.py``;
the resulting vectors live in ``tests/golden/*.npz``.
"""
