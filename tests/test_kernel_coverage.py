"""Every kernel instantiation of libdifformer_hip.so is launched by a GPU parity test.
`profiles/r06_kernel_coverage.txt` is what `scripts/kernel_coverage.py` wrote from rocprofv3 kernel traces of `pytest tests -m gpu`
on the MI355X; this CPU test holds it to the library the tree builds NOW: the same set of kernel symbols (a new instantiation
without a launch in the tracked file fails here until the coverage run is repeated), and nothing under UNLAUNCHED."""
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "scripts"))
import kernel_symbols as ks  # noqa: E402

TRACKED = os.path.join(ROOT, "profiles", "r06_kernel_coverage.txt")


def _tracked():
    launched, missing = {}, []
    section = launched
    for line in open(TRACKED):
        line = line.rstrip("\n")
        if not line or line.startswith("#"):
            continue
        if line.strip() == "UNLAUNCHED":
            section = None
            continue
        count, name = line.split(None, 1)
        if section is None:
            missing.append(name.strip())
        else:
            launched[name.strip()] = int(count)
    return launched, missing


def test_tracked_coverage_has_no_unlaunched_kernel():
    launched, missing = _tracked()
    assert not missing, f"{len(missing)} kernel instantiations were never launched by the GPU tests: {missing[:8]}"
    assert len(launched) > 300 and all(v > 0 for v in launched.values())


def test_tracked_coverage_matches_the_built_library():
    if not os.path.exists(ks.DEFAULT_LIB):
        pytest.skip("libdifformer_hip.so not built")
    launched, missing = _tracked()
    have = set(ks.kernels())
    tracked = set(launched) | set(missing)
    new = sorted(have - tracked)
    gone = sorted(tracked - have)
    assert not new, (f"{len(new)} kernel instantiations of the library are not in profiles/r06_kernel_coverage.txt -- add a parity "
                     f"test that launches them and re-run scripts/kernel_coverage.py on the GPU: {new[:8]}")
    assert not gone, f"profiles/r06_kernel_coverage.txt lists kernels the library no longer has: {gone[:8]}"
