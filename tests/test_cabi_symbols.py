"""CPU: the C-ABI library loads and exports every symbol include/tdc_hip.h declares (no compute without a GPU)."""
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    import __graft_entry__ as ge
    ge.build()
    import tdc_video_amd  # noqa: F401
    from tdc_video_amd import lib as L
    lib = L.load()
    hdr = open(os.path.join(ROOT, "include", "tdc_hip.h")).read()
    declared = set(re.findall(r"\b(tdc_[a-z0-9_]+)\s*\(", hdr))
    assert declared, "no declarations parsed"
    for name in sorted(declared):
        assert hasattr(lib, name), "symbol %s declared in include/tdc_hip.h but not exported" % name
    assert declared == set(L.SIGNATURES), (declared ^ set(L.SIGNATURES))
    assert lib.tdc_version().startswith(b"tdc_hip")


def test_product_path_never_imports_oracle():
    pkg = os.path.join(ROOT, "tdc-video_amd")
    for dp, _, files in os.walk(pkg):
        for f in files:
            if f.endswith(".py"):
                src = open(os.path.join(dp, f)).read()
                assert "oracle" not in src.replace("# oracle", ""), "%s references the oracle" % f
