"""ctypes binding of libr4r_hip.so (the C ABI declared in include/r4r.h).

The signatures are parsed from the header itself, so the header is the single
source of truth for the boundary.  There is NO fallback: if the shared library
is missing or a call fails, a RuntimeError is raised -- the product path never
routes through PyTorch ops or the CPU oracle.
"""
import ctypes
import os
import re
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
CSRC_DIR = os.path.join(_HERE, 'csrc')
LIB_PATH = os.environ.get('R4R_LIBRARY') or os.path.join(CSRC_DIR, 'libr4r_hip.so')   # override: instrumented builds
HEADER_PATH = os.path.join(os.path.dirname(_HERE), 'include', 'r4r.h')

_SCALARS = {
    'int': ctypes.c_int, 'float': ctypes.c_float, 'double': ctypes.c_double, 'int64_t': ctypes.c_int64,
    'uint64_t': ctypes.c_uint64, 'size_t': ctypes.c_size_t, 'int32_t': ctypes.c_int32, 'uint32_t': ctypes.c_uint32,
}

_lib = None
_decls = None


def parse_header(path=HEADER_PATH):
    """-> {name: (restype, [argtypes], [argnames])} for every function declared in r4r.h."""
    src = open(path).read()
    src = re.sub(r'/\*.*?\*/', '', src, flags=re.S)
    src = re.sub(r'//[^\n]*', '', src)
    src = '\n'.join(l for l in src.splitlines() if not l.strip().startswith('#'))
    decls = {}
    for m in re.finditer(r'([A-Za-z_][\w\s\*]*?)\b(r4r_\w+)\s*\(([^)]*)\)\s*;', src):
        ret, name, args = m.group(1).strip(), m.group(2), m.group(3).strip()
        if '*' in ret:
            restype = ctypes.c_char_p
        else:
            restype = _SCALARS[ret.replace('const', '').strip()]
        argtypes, argnames = [], []
        if args and args != 'void':
            for a in args.split(','):
                a = a.strip()
                argnames.append(re.findall(r'(\w+)\s*$', a)[0])
                if '*' in a:
                    argtypes.append(ctypes.c_void_p)
                else:
                    argtypes.append(_SCALARS[a.replace('const', '').split()[0]])
        decls[name] = (restype, argtypes, argnames)
    return decls


def build(verbose=False):
    """Compile libr4r_hip.so for gfx950 in-tree (hipcc cross-compiles without a GPU)."""
    out = subprocess.run(['make', '-C', CSRC_DIR, '-j4'], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if verbose or out.returncode != 0:
        print(out.stdout)
    if out.returncode != 0:
        raise RuntimeError('building libr4r_hip.so failed:\n' + out.stdout)
    return LIB_PATH


def lib():
    global _lib, _decls
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            'reviews4rec_amd: %s is missing. The HIP hot path has no fallback; build it with '
            '`python -c "import __graft_entry__ as g; g.build()"` (or `make -C reviews4rec_amd/csrc`).' % LIB_PATH)
    # PyTorch-ROCm ships its own libamdhip64.  Load it first: if this library pulled the system
    # runtime in before torch brought its own, the process would hold two HIP runtimes and the
    # second one finds no device ("no ROCm-capable device is detected").
    import torch  # noqa: F401
    l = ctypes.CDLL(LIB_PATH)
    _decls = parse_header()
    for name, (restype, argtypes, _) in _decls.items():
        fn = getattr(l, name)          # AttributeError here = header/library mismatch: fail loudly
        fn.restype = restype
        fn.argtypes = argtypes
    _lib = l
    return l


def declared_symbols():
    return sorted(parse_header())


def check(rc, what):
    if rc != 0:
        msg = lib().r4r_last_error()
        raise RuntimeError('%s failed (rc=%d): %s' % (what, rc, msg.decode() if msg else '?'))


def ptr(t):
    """Device pointer of a tensor (None -> NULL)."""
    return None if t is None else t.data_ptr()


def current_stream():
    import torch
    return torch.cuda.current_stream().cuda_stream


def call(name, *args):
    """Invoke an int-returning entry point on torch's current stream and raise on error."""
    rc = getattr(lib(), name)(*args, current_stream())
    check(rc, name)
