"""CPU oracle for the heatmap-regression hot path.  TEST INFRASTRUCTURE ONLY.

Everything under ``oracle/`` is a from-scratch CPU restatement (numpy / plain
PyTorch-CPU fp32) of what the reference computes on this path.  It exists so the
HIP path can be checked; it is never imported by ``lighthand_amd`` (the product).
Only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of
``bench.py`` may import it.

Pinning: the reference ships no tests or known-answer vectors for this path
(SURVEY.md section 4), so the oracle is pinned by outputs of the reference itself,
generated in the build container by ``tests/golden/make_golden.py`` (which imports
the reference from /root/reference) and committed as data under ``tests/golden/``
(G1 target render, G2 loss+grad, G3 arg-max decode, G5 whole-model forward,
G6 3-step Adam trajectory, G7 PCK/EPE/AUC).  ``tests/test_oracle_golden.py``
checks every oracle function against those fixtures on the CPU.
"""
