"""CPU oracle for the few-shot detection hot path.  TEST INFRASTRUCTURE -- NOT PRODUCT CODE.

This package restates, on the CPU (numpy + PyTorch-CPU fp32/fp64), the algorithms of the
reference path named by BASELINE.json:north_star:

    cfg parsing            -> oracle.cfgparse   (reference cfg.py:198-228)
    box IoU                -> oracle.boxes      (reference utils.py:21-83)
    region losses          -> oracle.region     (reference region_loss.py:15-366)
    Darknet meta detector  -> oracle.net        (reference darknet_meta.py:16-479,
                                                 dynamic_conv.py:110-168, pooling.py:8-60,
                                                 darknet.py:61-341, cfg.py:411-481)
    box decode + NMS       -> oracle.decode     (reference utils.py:85-104,195-290)

Only `tests/`, `__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline` leg may import it,
and only as the checker.  The product (`fewshot_detection_amd`) never imports it and fails
loudly when its HIP library is missing.

Parity pinning: the reference ships no tests or golden vectors, but the
reference itself *does* execute in the build container under the py2->py3 compatibility
substitutions listed in tests/golden/ref_shim.py.  tests/golden/make_golden.py runs the
reference that way and commits its inputs/outputs as fixtures under tests/golden/; the oracle
is checked against those fixtures (tests/test_oracle_vs_reference.py, CPU-only), and live
against the reference when /root/reference is present.
"""
