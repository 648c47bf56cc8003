"""ctypes binding of libchadavit_hip.so (C ABI: include/chadavit_hip.h).

There is NO fallback: if the HIP library is missing the import of any product module fails loudly.
"""
from __future__ import annotations

import ctypes
import os
import re

import torch  # noqa: F401  -- MUST precede the CDLL below: both link libamdhip64; torch's bundled runtime has to be the
#                              one already mapped, otherwise the process ends up with two HIP runtimes (hipErrorNoDevice)

HERE = os.path.dirname(os.path.abspath(__file__))
PRODUCT_LIB = os.path.join(HERE, "libchadavit_hip.so")


def _resolve_lib_path() -> str:
    """CHADAVIT_HIP_LIB loads another build of the same ABI instead of the product library (same-box A/B of kernel variants: the side
    builds under scratch/sidebuild/).  Such a library is foreign code with the product's name on its results, so the variable alone is
    not enough: it is honoured only together with CHADAVIT_ALLOW_FOREIGN_LIB=1, and refused loudly otherwise."""
    other = os.environ.get("CHADAVIT_HIP_LIB")
    if not other or os.path.realpath(other) == os.path.realpath(PRODUCT_LIB):
        return PRODUCT_LIB
    if os.environ.get("CHADAVIT_ALLOW_FOREIGN_LIB") != "1":
        raise HipExtensionMissing(
            f"CHADAVIT_HIP_LIB={other!r} names a library other than the product's ({PRODUCT_LIB}); set CHADAVIT_ALLOW_FOREIGN_LIB=1 to load it "
            "(A/B measurement runs only) or unset the variable")
    return other


LIB_PATH = None  # resolved at the first lib() call
HEADER = os.path.join(os.path.dirname(HERE), "include", "chadavit_hip.h")
ABI_VERSION = 9


class HipExtensionMissing(RuntimeError):
    pass


def declared_prototypes(header: str = HEADER):
    """{entry point: declared return type} for every `int chadavit_*(` / `long long chadavit_*(` in the public header."""
    with open(header) as f:
        src = f.read()
    return {name: ret for ret, name in re.findall(r"\b(int|long long)\s+(chadavit_\w+)\s*\(", src)}


def declared_symbols(header: str = HEADER):
    return sorted(declared_prototypes(header))


_ARG_TYPES = {"ptr": ctypes.c_void_p, "int": ctypes.c_int, "float": ctypes.c_float, "double": ctypes.c_double, "long long": ctypes.c_longlong}


def declared_signatures(header: str = HEADER):
    """{entry point: [argument kind, ...]} with kind in ptr / int / float / double / long long, parsed from the public header (comments
    stripped): what `lib()` installs as ctypes `argtypes` -- a call then takes plain Python ints / floats / addresses (no ctypes object per
    argument: ~1 ms of a 700-launch step) and a mistyped argument is an ArgumentError instead of a silently truncated register."""
    with open(header) as f:
        src = f.read()
    src = re.sub(r"/\*.*?\*/", " ", src, flags=re.S)
    src = re.sub(r"//[^\n]*", " ", src)
    out = {}
    for _, name, args in re.findall(r"\b(int|long long)\s+(chadavit_\w+)\s*\(([^;{]*?)\)\s*;", src, flags=re.S):
        kinds = []
        for a in args.split(","):
            a = " ".join(a.split())
            if a in ("", "void"):
                continue
            if "*" in a:
                kinds.append("ptr")
            else:
                t = a.rsplit(" ", 1)[0].replace("const ", "").strip()
                if t not in _ARG_TYPES:
                    raise HipExtensionMissing(f"{header}: {name}: argument type {t!r} is not one the binding knows")
                kinds.append(t)
        out[name] = kinds
    return out


_lib = None


def lib() -> ctypes.CDLL:
    global _lib, LIB_PATH
    if _lib is None:
        LIB_PATH = _resolve_lib_path()
        if not os.path.exists(LIB_PATH):
            raise HipExtensionMissing(
                f"{LIB_PATH} not found: build it with `python -m chadavit_amd.build` (hipcc, gfx950). "
                "chadavit_amd has no CPU fallback.")
        _lib = ctypes.CDLL(LIB_PATH)
        sigs = declared_signatures()
        for name, ret in declared_prototypes().items():
            fn = getattr(_lib, name)  # AttributeError if the library does not export a declared symbol
            fn.restype = ctypes.c_longlong if ret == "long long" else ctypes.c_int  # as the header declares it
            fn.argtypes = [_ARG_TYPES[k] for k in sigs[name]]
        if _lib.chadavit_abi_version() != ABI_VERSION:
            raise HipExtensionMissing(f"ABI mismatch: library {_lib.chadavit_abi_version()} != binding {ABI_VERSION}; rebuild")
    return _lib
