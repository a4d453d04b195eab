"""CPU tier: the hand-counted `s_waitcnt vmcnt(n)` of the tiled kernel against the ISA hipcc really emitted (cross-compiled
for gfx950, no GPU needed).  See tools/isa/check_vmcnt.py for what is counted and why a too-large count is harmless and a
too-small one is a silent race."""
import os
import subprocess

from conftest import ROOT


def test_counted_vmcnt_wait_matches_the_emitted_stores():
    csrc = os.path.join(ROOT, "rust-seq2kminmers_amd", "csrc")
    r = subprocess.run(["make", "-C", csrc, "-j", "2", "isa-check"], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    assert "vector-memory ops per round" in r.stdout
