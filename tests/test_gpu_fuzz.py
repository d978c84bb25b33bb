"""Randomised cross-check of all four search engines, the k-NN lists and the normals against the
oracle's brute force (scripts/fuzz_engines.py): 150 seeded trials here; 4,000 were run once
(DESIGN.md section 2)."""
import os
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "scripts"))


def test_fuzz_engines_against_brute_force(oracle, monkeypatch):
    import fuzz_engines
    monkeypatch.setattr(sys, "argv", ["fuzz_engines.py", "150", "20260"])
    assert fuzz_engines.main() == 0


def test_fuzz_bounded_passes_against_unbounded(oracle, monkeypatch):
    """scripts/fuzz_bounded.py: the same registration with the ICP loop's bounded passes (nn_bounded.h) and without, in one
    process, bit for bit: 60 seeded trials here."""
    import fuzz_bounded
    monkeypatch.setattr(sys, "argv", ["fuzz_bounded.py", "60", "7300"])
    monkeypatch.setenv("ICPMI_NN_BOUNDED", "1")   # (the script flips it per call; restored afterwards)
    assert fuzz_bounded.main() == 0


def test_fuzz_stopping_tests_against_the_oracle_loop(oracle, monkeypatch):
    """scripts/fuzz_stopping.py: registrations with the caller's settings (50, 1e-6, 1e-9) on LiDAR-like pairs and room
    corners through AUTO, the all-pairs and the culled engine: num_iterations, converged and history length equal to the
    oracle's on every trial whose stopping-test margin exceeds 1e-12 (trials under it are counted and printed): 40
    seeded trials here."""
    import fuzz_stopping
    assert fuzz_stopping.main(["fuzz_stopping.py", "40", "9000"]) == 0
