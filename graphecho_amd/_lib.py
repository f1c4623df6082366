"""ctypes binding of libgraphecho_hip.so, generated from include/graphecho_hip.h.

The product path has no CPU or ATen fallback for its hot ops: if the HIP library is missing this module
raises at import of the first op (``load()``), and every entry point's non-zero return code becomes a
``RuntimeError`` carrying ``ge_last_error()``.
"""
import ctypes
import os
import re

_HERE = os.path.dirname(os.path.abspath(__file__))
_ROOT = os.path.dirname(_HERE)
HEADER = os.path.join(_ROOT, "include", "graphecho_hip.h")
LIB_PATH = os.environ.get("GE_LIB_PATH") or os.path.join(_HERE, "csrc", "libgraphecho_hip.so")   # override: tuning builds

_SCALARS = {
    "int": ctypes.c_int,
    "long long": ctypes.c_longlong,
    "float": ctypes.c_float,
}
_DECL = re.compile(r"^\s*(const char\*|long long|int|void)\s+(ge_\w+)\s*\(([^;]*)\)\s*;", re.M)


def parse_header(path=HEADER):
    """Return {name: (restype, [argtypes])} for every ge_* declaration in the public header."""
    text = open(path).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    out = {}
    for ret, name, args in _DECL.findall(text):
        restype = {"const char*": ctypes.c_char_p, "long long": ctypes.c_longlong, "int": ctypes.c_int, "void": None}[ret]
        argtypes = []
        args = args.strip()
        if args and args != "void":
            for a in args.split(","):
                a = a.strip()
                if "*" in a:
                    argtypes.append(ctypes.c_void_p)
                else:
                    ty = a.rsplit(" ", 1)[0].replace("const ", "").strip()
                    argtypes.append(_SCALARS[ty])
        out[name] = (restype, argtypes)
    return out


class _Lib:
    def __init__(self):
        self._cdll = None
        self.signatures = None

    def load(self):
        if self._cdll is not None:
            return self
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f"graphecho_amd: HIP extension not built ({LIB_PATH} missing). "
                "Run `python -c 'import __graft_entry__ as g; g.build()'` or `make -C graphecho_amd/csrc`. "
                "There is no CPU fallback for the hot path."
            )
        # torch bundles its own libamdhip64; import it FIRST so this library binds to the same HIP runtime instance
        # (streams and device pointers are shared with torch).  Loading in the other order gives two runtimes.
        import torch  # noqa: F401

        cdll = ctypes.CDLL(LIB_PATH)
        self.signatures = parse_header()
        for name, (restype, argtypes) in self.signatures.items():
            fn = getattr(cdll, name)  # AttributeError if the header declares a symbol the library lacks
            fn.restype = restype
            fn.argtypes = argtypes
        self._cdll = cdll
        return self

    def __getattr__(self, name):
        if name.startswith("ge_"):
            self.load()
            fn = getattr(self._cdll, name)
            self.__dict__[name] = fn     # later lookups find the entry point on the instance (a step makes ~1500 of them)
            return fn
        raise AttributeError(name)

    def last_error(self):
        self.load()
        msg = self._cdll.ge_last_error()
        return msg.decode() if msg else ""


lib = _Lib()


def check(rc, what=""):
    if rc != 0:
        raise RuntimeError(f"libgraphecho_hip: {what} failed with code {rc}: {lib.last_error()}")


def build(verbose=False):
    """Compile libgraphecho_hip.so for gfx950 in-tree (hipcc cross-compiles without a GPU)."""
    import subprocess

    cmd = ["make", "-C", os.path.join(_HERE, "csrc"), "-j8"]
    res = subprocess.run(cmd, capture_output=not verbose, text=True)
    if res.returncode != 0:
        raise RuntimeError("hipcc build failed:\n" + (res.stdout or "") + (res.stderr or ""))
    return LIB_PATH
