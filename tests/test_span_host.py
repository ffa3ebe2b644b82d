"""Host side of the span entry points (include/r4r.h "Spans", csrc/span.hip, engine._Spans) where no GPU is needed:
the descriptor's batch arithmetic (r4r_span_batch is pure host code), its argument checks, the resident-table form,
and the conv rule's span planning against its per-step twin.  The loop these replace: /root/reference/main.py:23-60
over data_fast.py:99-109 / data.py:250-372 (contiguous slices of batch_size ratings, ragged tail)."""
import ctypes

import pytest

from reviews4rec_amd import _lib

WORDS = 28


def _desc(**kw):
    w = [0] * WORDS
    for k, v in kw.items():
        w[int(k[1:])] = v
    return (ctypes.c_uint64 * WORDS)(*w)


def _slots(words, b):
    out = (ctypes.c_uint64 * 8)()
    rc = _lib.lib().r4r_span_batch(words, b, out)
    return rc, list(out)


def test_batch_pointers_follow_the_group_layout():
    """Batch b lives in group b // G, ring slot (b // G) & 1, as rows [j B, (j + 1) B) of the group's five fields
    [n, doc] [n, 10] [n, 10] [n, doc] [n, doc] (r4r_batch_build's block for the group's n ratings)."""
    B, G, T, N = 8, 4, 16, 8 * 11 + 5                       # 11 full batches: groups of 4, 4, 3; a ragged tail of 5
    ring, stride = 0x100000, 4 * 8 * (3 * 16 + 20) + 64
    u, i, y = 0x2000000, 0x3000000, 0x4000000
    words = _desc(w10=u, w11=i, w12=1, w13=1, w14=1, w15=y, w17=T, w22=ring, w23=stride, w24=N, w25=G, w26=B)
    for b in range(11):
        g, j = divmod(b, G)
        n = min(G * B, 11 * B - g * G * B)
        base = ring + (g & 1) * stride * 8
        rc, s = _slots(words, b)
        assert rc == 0
        assert s[0] == base + 8 * (j * B * T)
        assert s[1] == base + 8 * (n * T + j * B * 10)
        assert s[2] == base + 8 * (n * T + n * 10 + j * B * 10)
        assert s[3] == base + 8 * (n * T + 2 * n * 10 + j * B * T)
        assert s[4] == base + 8 * (2 * n * T + 2 * n * 10 + j * B * T)
        assert s[5:] == [u + 8 * b * B, i + 8 * b * B, y + 4 * b * B]
    assert _slots(words, 11)[0] != 0                        # the ragged tail is not a span batch
    assert _slots(words, -1)[0] != 0
    # NARRE: documents are R x W
    words = _desc(w10=u, w11=i, w12=1, w13=1, w14=1, w15=y, w17=100, w18=10, w19=12, w22=ring, w23=10 ** 6, w24=64, w25=2, w26=8)
    rc, s = _slots(words, 3)
    assert rc == 0 and s[3] - s[0] == 8 * (16 * 120 + 2 * 16 * 10)


def test_descriptor_checks():
    lib = _lib.lib()
    assert lib.r4r_span_batch(None, 0, (ctypes.c_uint64 * 8)()) != 0 and b'span' in lib.r4r_last_error()
    small = _desc(w10=1, w11=1, w12=1, w13=1, w14=1, w15=1, w17=16, w22=0x1000, w23=10, w24=64, w25=4, w26=8)
    assert _slots(small, 0)[0] != 0 and b'stride' in lib.r4r_last_error()       # the ring cannot hold a group
    no_ids = _desc(w17=16, w24=64, w26=8)
    assert _slots(no_ids, 0)[0] != 0
    built = ctypes.c_int64(-1)
    ids_only = _desc(w10=1, w11=1, w15=1, w24=64, w26=8)
    assert lib.r4r_span_build(ids_only, 0, ctypes.byref(built), None) != 0       # nothing to build for an ids-only loader
    rc, s = _slots(ids_only, 7)
    assert rc == 0 and s[:5] == [0] * 5 and s[5] == 1 + 8 * 56


def test_resident_table_cycles():
    """Word 27: a host table of G built batches; batch b is entry b % G, nothing is constructed."""
    G = 3
    table = (ctypes.c_uint64 * (8 * G))(*range(100, 100 + 8 * G))
    words = _desc(w24=1 << 40, w25=G, w26=128, w27=ctypes.addressof(table))
    for b in (0, 1, 2, 3, 7, 3000001):
        rc, s = _slots(words, b)
        assert rc == 0 and s == list(range(100 + 8 * (b % G), 108 + 8 * (b % G)))


def test_every_declared_span_symbol_is_exported():
    lib, decl = _lib.lib(), _lib.parse_header()
    names = [n for n in decl if n.endswith('_span') or n.startswith('r4r_span_')]
    assert sorted(names) == sorted(['r4r_span_build', 'r4r_span_batch', 'r4r_deepconn_span', 'r4r_deepconnpp_span',
                                    'r4r_narre_span', 'r4r_transnet_span', 'r4r_mf_span', 'r4r_idnet_span'])
    for n in names:
        assert getattr(lib, n) is not None
        # a span's arguments are its family's step arguments: the per-step pointers give way to the descriptor
        assert decl[n][2][0] == 'loader'


@pytest.mark.parametrize('E,V,T,docs', [(304, 50002, 1000, 128), (64, 50002, 100, 1280), (64, 1000000, 1000, 128)])
def test_span_planning_agrees_with_the_per_step_rule(E, V, T, docs):
    """engine._ConvRule: _rule_peek (what a span may cover) against _rule_request (what every single step asks for):
    a span never contains a probing step, carries the request its steps would have made, and advances the rule's count
    like they would."""
    from reviews4rec_amd.engine import _ConvRule

    class Rule(_ConvRule):
        def __init__(self):
            self.E, self.V, self.conv_algo, self.gemm_math = E, V, 0, 'f32'
            self._rule_reset()

    a, b = Rule(), Rule()
    step = 0
    while step < 2200:
        limit, req, algo, counted = b._rule_peek(docs, T)
        if limit <= 0:                                      # the next step is its own: it must be a probing one
            r, al, probe = a._rule_request(docs, T, True)
            assert probe
            b._rule_request(docs, T, True)
            a._rule_choice = b._rule_choice = 2             # (what a probe would have decided)
            step += 1
            continue
        k = min(limit, 64)
        for _ in range(k):
            r, al, probe = a._rule_request(docs, T, True)
            assert not probe and (r, al) == (req, algo)
        b._rule_advance(k, counted)
        assert a._rule_n == b._rule_n
        # `ahead`: the planner's look past the span it is about to issue
        assert (b._rule_peek(docs, T)[0] > 0) == (Rule._rule_peek(a, docs, T)[0] > 0)
        step += k
