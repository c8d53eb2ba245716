"""ctypes binding of libtris_hip.so (the C ABI declared in include/tris_hip.h).

The header is the single source of truth: argument types are parsed from it, so a signature change
cannot silently desynchronise the Python side.  There is NO fallback: if the library is missing or a
kernel returns an error the call raises.
"""
import ctypes
import os
import re

# PyTorch bundles its own libamdhip64.so.7.  It must be in the process BEFORE libtris_hip.so is dlopen'ed so that both
# resolve to the SAME HIP runtime instance (same device context, streams and allocations); loading ours first would pull
# /opt/rocm's copy and torch would then talk to a second, uninitialised runtime (hipErrorNoDevice on the first launch).
import torch  # noqa: F401

_HERE = os.path.dirname(os.path.abspath(__file__))
_ROOT = os.path.dirname(_HERE)
HEADER = os.path.join(_ROOT, "include", "tris_hip.h")
LIBPATH = os.environ.get("TRIS_HIP_LIB") or os.path.join(_HERE, "libtris_hip.so")  # env: developer knob (kernel experiments)

_CT = {
    "int": ctypes.c_int,
    "long": ctypes.c_long,
    "float": ctypes.c_float,
}


def parse_header(path=HEADER):
    """-> {name: (restype, [argtypes]), ...} plus the TRIS_EW_* constants."""
    src = open(path).read()
    consts = {m.group(1): int(m.group(2)) for m in re.finditer(r"#define\s+(TRIS_\w+)\s+(\d+)", src)}
    src = re.sub(r"/\*.*?\*/", " ", src, flags=re.S)
    decls = {}
    for m in re.finditer(r"\b(int|long)\s+(tris_\w+)\s*\(([^)]*)\)\s*;", src):
        ret, name, args = m.group(1), m.group(2), m.group(3)
        types = []
        for a in args.split(","):
            a = a.strip()
            if a in ("", "void"):
                continue
            if "*" in a:
                types.append(ctypes.c_void_p)
            else:
                base = a.split()[0] if a.split()[0] != "const" else a.split()[1]
                types.append(_CT[base])
        decls[name] = (_CT[ret], types)
    return decls, consts


DECLS, CONSTS = parse_header()
_lib = None


class TrisHipError(RuntimeError):
    pass


def load():
    global _lib
    if _lib is None:
        if not os.path.exists(LIBPATH):
            raise TrisHipError(
                f"{LIBPATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "(tris_amd has no CPU or eager fallback by design)")
        lib = ctypes.CDLL(LIBPATH)
        for name, (ret, args) in DECLS.items():
            fn = getattr(lib, name)  # AttributeError if the .so lacks a declared symbol
            fn.restype = ret
            fn.argtypes = args
        _lib = lib
    return _lib


_FN = {}


def call(name, *args):
    """Call an `int`-returning entry point; raise on a non-zero hipError_t."""
    fn = _FN.get(name)
    if fn is None:
        fn = _FN[name] = getattr(load(), name)
    err = fn(*args)
    if err != 0:
        raise TrisHipError(f"{name} failed with hipError_t {err}")


def query(name, *args):
    """Call a `long`-returning helper (workspace sizes)."""
    return getattr(load(), name)(*args)
