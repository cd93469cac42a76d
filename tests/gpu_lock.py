"""Shared / exclusive use of the ONE GPU among the xdist workers of `pytest -m gpu`.

Launches whose blocks wait for each other (include/xv2.h: xv2_conv2d_forward_bn_act and friends) need their whole grid
resident at once; two partly resident ones from different processes would wait for each other (the kernel traps after 4 s).
Rule that makes this impossible: the gated grids of all processes on the GPU together stay <= 240 blocks - then every block
finds a CU of its own in the worst case (256 CUs), whatever the kernels' footprints and however the dispatcher interleaves.
  * every GPU test holds this lock SHARED and runs with the workers' small cap (XV2_COOP_BLOCKS = 240 / workers, conftest.py);
  * a test marked `gpu_exclusive` holds it EXCLUSIVE - nobody else is on the GPU - and may lift its own cap to the chip's
    capacity (xv2_set_coop_blocks): the tests of large gated grids (tests/test_coop_gpu.py).
A turnstile file keeps a waiting exclusive holder from being starved by a stream of shared ones (flock has no fairness)."""
import contextlib
import fcntl
import os
import tempfile

_DIR = tempfile.gettempdir()


@contextlib.contextmanager
def gpu_lock(exclusive):
    turn = open(os.path.join(_DIR, "xv2_gpu_turnstile.lock"), "w")
    main = open(os.path.join(_DIR, "xv2_gpu_main.lock"), "w")
    try:
        fcntl.flock(turn, fcntl.LOCK_EX)
        try:
            fcntl.flock(main, fcntl.LOCK_EX if exclusive else fcntl.LOCK_SH)
        finally:
            fcntl.flock(turn, fcntl.LOCK_UN)
        try:
            yield
        finally:
            fcntl.flock(main, fcntl.LOCK_UN)
    finally:
        turn.close()
        main.close()
