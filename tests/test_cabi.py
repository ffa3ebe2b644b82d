"""CPU-side checks of the drop-in boundary: the shared library loads, exports every
symbol include/r4r.h declares, argument validation works without a GPU, and the
product package never touches the oracle."""
import ctypes
import os
import re

import pytest
import torch

from helpers import GOLDEN_CASES, Golden

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope='module')
def lib():
    from reviews4rec_amd import _lib
    _lib.build()
    return _lib.lib()


def test_header_declares_the_expected_surface():
    from reviews4rec_amd import _lib
    names = _lib.declared_symbols()
    for must in ('r4r_textcnn_fwd', 'r4r_textcnn_wgrad', 'r4r_fm_fwd', 'r4r_fm_bwd', 'r4r_adam_multi',
                 'r4r_narre_attn_fwd', 'r4r_narre_attn_bwd', 'r4r_embed_gather', 'r4r_embed_scatter_add',
                 'r4r_linear_fwd', 'r4r_linear_bwd', 'r4r_dropout_fwd', 'r4r_last_error', 'r4r_version'):
        assert must in names
    assert len(names) >= 26


def test_library_exports_every_declared_symbol(lib):
    from reviews4rec_amd import _lib
    for name in _lib.declared_symbols():
        assert hasattr(lib, name), name
    assert lib.r4r_version() >= 1
    assert lib.r4r_adam_chunk_elems() == 8192


def test_every_entry_point_cites_the_reference():
    """include/r4r.h must say which reference call site each group replaces (file:line)."""
    src = open(os.path.join(ROOT, 'include', 'r4r.h')).read()
    assert len(re.findall(r'\b\w+\.py:\d+', src)) >= 25


def test_argument_validation_needs_no_gpu(lib):
    # E not a multiple of 4 -> R4R_ERR_ARG before any launch
    rc = lib.r4r_textcnn_fwd(1, 10, 1, 1, 1, 1, 1, 1, 1 << 30, 2, 8, 6, 100, None)
    assert rc == -1 and b'multiple of 4' in lib.r4r_last_error()
    rc = lib.r4r_textcnn_fwd(1, 10, 1, 1, 1, 1, 1, 1, 16, 2, 8, 8, 100, None)
    assert rc == -3 and b'workspace' in lib.r4r_last_error()
    assert lib.r4r_textcnn_ws_bytes(128, 1000, 300, 100, 50002) > 0
    rc = lib.r4r_fm_fwd(1, 1, 1, 1, 1, 4, 513, 8, None)             # (n <= 512 since round 3: several inputs per lane)
    assert rc == -1
    rc = lib.r4r_dropout_fwd(1, 1, 1, 4, ctypes.c_float(1.5), 0, 0, None, None)
    assert rc == -1


@pytest.mark.parametrize('case', GOLDEN_CASES)
def test_state_dict_keys_and_shapes_match_reference(case):
    """Checkpoint compatibility (main.py:125,133): same key set, same shapes."""
    import reviews4rec_amd
    g = Golden(case)
    ref = g.params()
    hp = dict(g.hp)
    key = 'target.word2vec.weight' if hp['model_type'].startswith('transnet') else 'word2vec.weight'
    if key in ref:
        hp['word_vectors'] = ref[key].numpy()
    model = reviews4rec_amd.get_model_class(hp['model_type'])(hp)
    sd = model.state_dict()
    assert set(sd) == set(ref)
    for k in ref:
        assert tuple(sd[k].shape) == tuple(ref[k].shape), k
    model.load_state_dict(ref, strict=True)
    frozen = [k for k, p in model.named_parameters() if not p.requires_grad]
    assert frozen == ([key] if key in ref else [])           # only the word table is frozen


def test_models_refuse_cpu_tensors():
    import reviews4rec_amd
    g = Golden('mf_dot')
    model = reviews4rec_amd.get_model_class('MF_dot')(dict(g.hp))
    with pytest.raises(RuntimeError, match='ROCm device'):
        model(g.batch(0)[0])


def test_product_package_never_imports_the_oracle():
    pkg = os.path.join(ROOT, 'reviews4rec_amd')
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith(('.py', '.hip', '.cpp', '.h')):
                src = open(os.path.join(dirpath, f)).read()
                assert not re.search(r'^\s*(from|import)\s+oracle\b', src, re.M), os.path.join(dirpath, f)
                assert 'oracle/' not in src and 'oracle.' not in src.replace('the CPU oracle.', ''), f


def test_header_is_plain_c(tmp_path):
    """include/r4r.h is the boundary a C / cgo / JNI host would include: it must compile as C99, not
    only as C++ (the library itself includes it from .hip / .cpp files)."""
    import shutil
    import subprocess
    if shutil.which('gcc') is None:
        pytest.skip('no gcc')
    src = tmp_path / 'use_r4r.c'
    src.write_text('#include "r4r.h"\n'
                   'int main(void) {\n'
                   '    size_t (*ws)(int64_t, int, int, int, int64_t) = r4r_textcnn_ws_bytes;\n'
                   '    const char *(*err)(void) = r4r_last_error;\n'
                   '    return (ws != 0 && err != 0 && R4R_OK == 0 && R4R_TIMING_SLOTS > 0) ? r4r_version() : 1;\n'
                   '}\n')
    inc = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'include')
    out = subprocess.run(['gcc', '-std=c99', '-Wall', '-Wextra', '-pedantic', '-Werror', '-fsyntax-only',
                          '-I', inc, str(src)], capture_output=True, text=True)
    assert out.returncode == 0, out.stderr


def test_integration_stub_matches_the_header():
    """The reference-side ctypes stub printed in INTEGRATION.md, executed verbatim against the built
    library: every argtypes / restype it declares must be what include/r4r.h declares."""
    import ctypes
    import re
    from reviews4rec_amd import _lib
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    text = open(os.path.join(root, 'INTEGRATION.md')).read()
    blocks = [b for b in re.findall(r'```python\n(.*?)```', text, flags=re.S) if 'r4r_binding.py' in b]
    assert len(blocks) == 1
    code = blocks[0].replace("ctypes.CDLL('libr4r_hip.so')", 'ctypes.CDLL(%r)' % _lib.LIB_PATH)
    ns = {}
    exec(compile(code, 'INTEGRATION.md:r4r_binding.py', 'exec'), ns)
    decls = _lib.parse_header()
    stub = ns['_lib']
    touched = set(re.findall(r'_lib\.(r4r_\w+)\.(?:argtypes|restype)', code))
    assert {'r4r_textcnn_ws_bytes', 'r4r_textcnn_fwd', 'r4r_last_error'} <= touched
    for name in touched:
        restype, argtypes, _ = decls[name]
        fn = getattr(stub, name)
        if fn.argtypes is not None:
            assert list(fn.argtypes) == argtypes, name
        assert fn.restype in (restype, ctypes.c_int if restype is ctypes.c_int else restype), name
    # and the call sites inside the stub pass as many arguments as the header declares
    for name in ('r4r_textcnn_ws_bytes', 'r4r_textcnn_fwd'):
        call = re.search(r'_lib\.%s\((.*?)\)\n' % name, code, flags=re.S).group(1)
        depth, n = 0, 1
        for ch in call:
            depth += ch in '([' 
            depth -= ch in ')]'
            n += (ch == ',' and depth == 0)
        assert n == len(decls[name][1]), (name, n)


def test_conv_pick_reproduces_the_measured_decisions():
    """r4r_conv_pick's cost model against the (configuration, data) points measured on MI355X
    (profiles/r02_conv_rule.txt): distinct rows -> the algorithm that was faster."""
    from reviews4rec_amd import _lib
    lib = _lib.lib()
    PROJECT, DIRECT = 2, 1
    points = [   # E, T, documents (all towers), vocabulary, [(distinct rows, faster algorithm), ...]
        (300, 1000, 256, 50002, [(29547, PROJECT), (45464, PROJECT), (71421, PROJECT), (92264, PROJECT)]),      # cfg3
        (64, 100, 2560, 50002, [(19741, PROJECT), (31970, PROJECT), (49182, PROJECT), (75188, PROJECT)]),       # cfg4
        (64, 1000, 384, 1000000, [(76899, PROJECT), (137449, DIRECT), (177702, DIRECT), (360391, DIRECT)]),     # cfg5
        (300, 1000, 16, 50002, [(3757, PROJECT)]), (300, 1000, 32, 50002, [(6402, PROJECT)]),                    # cfg3 at B = 8, 16:
        (300, 1000, 64, 50002, [(11055, PROJECT)]),                                                              # ... 32 (column parts)
    ]
    for E, T, docs, V, cases in points:
        for rows, want in cases:
            assert lib.r4r_conv_pick(E, T, docs, rows, V) == want, (E, T, docs, V, rows)
    # the request -> algorithm map the engines consult (static rule; F != 100 has no projection kernels)
    assert lib.r4r_conv_algo(0, 128, 1000, 300, 100) == PROJECT and lib.r4r_conv_algo(0, 8, 100, 64, 100) == DIRECT
    assert lib.r4r_conv_algo(1, 128, 1000, 300, 100) == DIRECT and lib.r4r_conv_algo(2, 8, 100, 64, 100) == PROJECT
    assert lib.r4r_conv_algo(2, 128, 1000, 300, 64) == DIRECT
