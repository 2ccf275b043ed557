"""CPU oracle for the LFCC -> ResNet/ECAPA -> OC-Softmax hot path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``oracle/`` is product code: only
``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg
may import it, and only as the checker.  The product path
(``asvspoof2021_air_amd``) never imports this package and fails loudly when
the HIP extension is missing.

Every function restates one piece of the reference algorithm
(``/root/reference``, yzyouzhang/ASVspoof2021_AIR) and cites the file:line it
follows.  Parity is PINNED: ``tests/golden/make_golden.py`` imports the real
reference (under four compatibility shims) in the build container, checks
each restatement against it and writes the fixtures in ``tests/golden/*.npz``
that ``tests/test_oracle_golden.py`` re-checks everywhere else.

Floating-point restatements use numpy / PyTorch-CPU fp32 (the reference is
itself PyTorch fp32); a float64 variant exists where an independent
higher-precision check is useful (LFCC).
"""
