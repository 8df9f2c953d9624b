"""oracle/ref_loader.py -- import the *reference itself* in the build container.

TEST INFRASTRUCTURE ONLY.  Used by tests/golden/make_golden.py and by the `-m "not gpu"` tests
that cross-check oracle/ against the reference when /root/reference is mounted.  Nothing here is
reachable on the GPU box (no /root/reference there) and nothing here is used by the product path.

The reference package cannot be imported as shipped in this image (rocco/__init__.py:3 pulls in
rocco/scores.py:17 -> `import pysam`, which is not installed), so a stub parent package named
`rocco` is registered whose search path is the reference's own `rocco/` directory followed by
oracle/_ref/ (where `make -C oracle ref` puts the `_chain_dp` extension compiled from the
reference's own source).  Only `rocco.dp` is imported; `rocco.rocco`'s BED helpers are loaded with
a dummy `pysam` module in place.
"""
from __future__ import annotations

import importlib
import os
import sys
import types

REFERENCE_ROOT = os.environ.get("ROCCO_REFERENCE_ROOT", "/root/reference")
_REF_BUILD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "_ref")


def reference_available() -> bool:
    return os.path.isfile(os.path.join(REFERENCE_ROOT, "rocco", "dp.py")) and any(
        name.startswith("_chain_dp") for name in (os.listdir(_REF_BUILD) if os.path.isdir(_REF_BUILD) else [])
    )


def _ensure_stub_package() -> None:
    if "rocco" in sys.modules and getattr(sys.modules["rocco"], "__rocco_reference_stub__", False):
        return
    stub = types.ModuleType("rocco")
    stub.__path__ = [os.path.join(REFERENCE_ROOT, "rocco"), _REF_BUILD]
    stub.__rocco_reference_stub__ = True
    sys.modules["rocco"] = stub


def load_reference_dp():
    """Return the reference's `rocco.dp` module (with its compiled `_chain_dp`)."""
    if not reference_available():
        raise RuntimeError("reference not available (need /root/reference and `make -C oracle ref`)")
    _ensure_stub_package()
    dp = importlib.import_module("rocco.dp")
    if dp._chain_dp is None:  # pragma: no cover
        raise RuntimeError("reference _chain_dp extension did not load from oracle/_ref")
    return dp


def load_reference_rocco():
    """Return the reference's `rocco.rocco` module (BED helpers, median scoring)."""
    if not reference_available():
        raise RuntimeError("reference not available")
    _ensure_stub_package()
    if "pysam" not in sys.modules:
        dummy = types.ModuleType("pysam")
        dummy.AlignedSegment = type("AlignedSegment", (), {})
        sys.modules["pysam"] = dummy
    return importlib.import_module("rocco.rocco")
