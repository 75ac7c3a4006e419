"""CPU-side checks of the drop-in boundary: the C-ABI library loads and exports every symbol
include/mcp_ba.h declares, and refuses to run (loudly) without a GPU -- no compute here."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared(header):
    txt = open(os.path.join(ROOT, "include", header)).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    names = set(re.findall(r"\b(mcp_[a-z0-9_]+)\s*\(", txt))
    return sorted(n for n in names if not n.endswith("_fn"))


def test_library_exports_every_declared_symbol():
    import __graft_entry__ as g
    g.build()
    from mcptam_amd import chain_bundle
    L = ctypes.CDLL(chain_bundle.LIB_PATH)
    names = _declared("mcp_ba.h")
    assert len(names) >= 25
    for n in names:
        assert hasattr(L, n), "libmcptam_hip.so does not export " + n
    assert set(chain_bundle.BA_SYMBOLS) == set(names)
    from mcptam_amd import keyframe
    img_names = [n for n in _declared("mcp_img.h")]
    assert len(img_names) >= 16
    for n in img_names:
        assert hasattr(L, n), "libmcptam_hip.so does not export " + n
    assert set(keyframe.IMG_SYMBOLS) == set(img_names)


def test_create_fails_loudly_without_gpu():
    from mcptam_amd import chain_bundle, synth
    if chain_bundle.device_count() > 0:
        pytest.skip("a GPU is present")
    p = synth.make_config("tiny")
    with pytest.raises(RuntimeError):
        chain_bundle.ChainBundle(p.cams)


def test_product_package_does_not_import_oracle():
    """The product path must never route through oracle/."""
    for dirpath, _, files in os.walk(os.path.join(ROOT, "mcptam_amd")):
        for f in files:
            if f.endswith((".py", ".h", ".hip", ".cpp", ".hpp")) or f == "Makefile":
                txt = open(os.path.join(dirpath, f)).read()
                assert not re.search(r"import\s+oracle|from\s+oracle|liborc|oracle/|ba_oracle|img_oracle", txt), os.path.join(dirpath, f)
