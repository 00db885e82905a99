"""Import the product's ctypes binding (its package directory has a '-' in the name)."""
import importlib.util
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
PKG = ROOT / "opv-cxx-demod_amd"


def load():
    if "opv_amd" in sys.modules:
        return sys.modules["opv_amd"]
    spec = importlib.util.spec_from_file_location("opv_amd", PKG / "opv_amd.py")
    mod = importlib.util.module_from_spec(spec)
    sys.modules["opv_amd"] = mod
    spec.loader.exec_module(mod)
    return mod
