"""ctypes binding of libhvlm_hip.so (the C ABI declared in include/hvlm.h).

The product path has NO fallback: if the shared library is missing or a symbol is absent, importing
the ops raises.  Signatures are parsed from include/hvlm.h so the header stays the single source of
truth (tests check that every declared symbol is exported).
"""
from __future__ import annotations

import ctypes
import os
import re
from pathlib import Path

_PKG_ROOT = Path(__file__).resolve().parent.parent          # handsonvlm-release_amd/
REPO_ROOT = _PKG_ROOT.parent
LIB_PATH = _PKG_ROOT / "lib" / "libhvlm_hip.so"
HEADER_PATH = REPO_ROOT / "include" / "hvlm.h"

_CTYPES = {
    "int": ctypes.c_int,
    "long": ctypes.c_long,
    "float": ctypes.c_float,
    "size_t": ctypes.c_size_t,
    "void": None,
}


def _ctype_of(decl: str):
    decl = decl.strip()
    if "*" in decl:
        if decl.startswith("const char") or decl.startswith("char"):
            return ctypes.c_char_p
        return ctypes.c_void_p
    decl = decl.replace("const ", "").strip()
    base = decl.split()[0]
    return _CTYPES[base]


def parse_header(path: Path = HEADER_PATH, dev: bool = False):
    """Returns {name: (restype, [argtypes])} for every `hvlm_*` prototype in the header; the `#ifdef HVLM_DEV` blocks
    (entry points that exist in `make DEV=1` builds only) are included with dev=True."""
    text = path.read_text()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    text = re.sub(r"//[^\n]*", "", text)
    if not dev:
        text = re.sub(r"^\s*#ifdef HVLM_DEV\b.*?^\s*#endif[^\n]*$", "", text, flags=re.S | re.M)
    text = re.sub(r"^\s*#[^\n]*$", "", text, flags=re.M)          # preprocessor lines
    text = text.replace('extern "C" {', "")
    protos = {}
    for m in re.finditer(r"([A-Za-z_][\w\s\*]*?)\b(hvlm_\w+)\s*\(([^;{]*?)\)\s*;", text):
        ret, name, args = m.group(1).strip(), m.group(2), m.group(3).strip()
        if ret.startswith("const char"):
            restype = ctypes.c_char_p
        else:
            restype = _ctype_of(ret)
        argtypes = []
        if args and args != "void":
            for a in args.split(","):
                a = a.strip()
                # strip the parameter name
                if "*" in a:
                    argtypes.append(_ctype_of(a))
                else:
                    argtypes.append(_ctype_of(" ".join(a.split()[:-1]) or a))
        protos[name] = (restype, argtypes)
    return protos


class HvlmError(RuntimeError):
    pass


_lib = None
_protos = None


def load(path: os.PathLike | None = None) -> ctypes.CDLL:
    """Loads the library (once) and attaches prototypes.  Raises if it is missing."""
    global _lib, _protos
    if _lib is not None:
        return _lib
    # HVLM_LIB: development A/B of two builds on one box (e.g. lib/libhvlm_base.so); the default is the in-tree library
    p = Path(path) if path else (Path(os.environ["HVLM_LIB"]) if os.environ.get("HVLM_LIB") else LIB_PATH)
    if not p.exists():
        raise HvlmError(
            f"{p} not found: build it with `make -C {_PKG_ROOT}` (or __graft_entry__.build()). "
            "There is no CPU/PyTorch fallback for the HIP path.")
    # The library must share the HIP runtime PyTorch brings (torch/lib/libamdhip64.so): whichever copy is mapped first
    # serves both by soname, and with the system copy mapped first a later torch.cuda initialisation left this library
    # without a device ("no ROCm-capable device is detected" at the first launch).  Import torch before dlopen.
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    lib = ctypes.CDLL(str(p))
    lib.hvlm_is_dev_build.restype = ctypes.c_int
    protos = parse_header(dev=bool(lib.hvlm_is_dev_build()))
    for name, (restype, argtypes) in protos.items():
        try:
            fn = getattr(lib, name)
        except AttributeError as e:
            raise HvlmError(f"{p} does not export {name} declared in {HEADER_PATH}") from e
        fn.restype = restype
        fn.argtypes = argtypes
    if lib.hvlm_abi_version() != 2:
        raise HvlmError("ABI version mismatch between include/hvlm.h and libhvlm_hip.so")
    _lib, _protos = lib, protos
    return lib


def last_error() -> str:
    return load().hvlm_last_error().decode()


def check(rc: int, what: str = "") -> None:
    if rc != 0:
        raise HvlmError(f"{what}: rc={rc}: {last_error()}")


def configure(key: str, value: int) -> None:
    """hvlm_configure: sets one named library switch (include/hvlm.h).  Raises HvlmError for an unknown switch, or for a
    development-only variant in a production build."""
    check(load().hvlm_configure(key.encode(), int(value)), f"configure({key}={value})")


def config_get(key: str) -> int:
    v = ctypes.c_int(0)
    check(load().hvlm_config_get(key.encode(), ctypes.byref(v)), f"config_get({key})")
    return v.value


class configured:
    """Context manager: `with configured(decode_unfused=1): ...` sets switches and restores the previous values."""

    def __init__(self, **kv):
        self.kv, self.old = kv, {}

    def __enter__(self):
        for k, v in self.kv.items():
            self.old[k] = config_get(k)
            configure(k, v)
        return self

    def __exit__(self, *exc):
        for k, v in self.old.items():
            configure(k, v)
        return False
