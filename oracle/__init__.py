"""ORACLE — test infrastructure only.

CPU (PyTorch fp32, NCHW) restatement of the reference's U-Net hot path.  Only tests/, bench.py's
cpu_baseline leg and __graft_entry__.smoke() may import this package, and only as the checker; the
product (xview2_amd) never imports it and has no CPU fallback.
"""
