"""CPU: the build-time check of csrc/ap_stream.hip (tools/check_stream_regs.py) -- every inline-asm plane request of every kernel
instance writes one of the ring's fixed register tuples.  A request site with registers of its own would be followed by a copy
into the slot's registers BEFORE the data has landed (found as NaNs in round 4); hipcc cross-compiles without a GPU."""
import os
import subprocess
import sys

from conftest import ROOT


def test_every_plane_request_writes_a_ring_slot():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "check_stream_regs.py")], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
    assert r.stdout.count(" ok") >= 12 and "MISMATCH" not in r.stdout
