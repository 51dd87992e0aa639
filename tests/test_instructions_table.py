"""Native instruction table: append / append-unique / get round trip and dependencies.
Mirrors /root/reference/examples/testing/testInstructionsTable.m:20-63 and the self-test in
lib/csparse/instructionsTableUTHash.c:1980-2061."""
import numpy as np

from tenscalc_amd import instructions as I
from tenscalc_amd.instructions import InstructionsTable


def test_append_get_roundtrip(native_built):
    rs = np.random.RandomState(0)
    t = InstructionsTable()
    base = [t.append(I.I_set, unique=False) for _ in range(10)]
    assert base == list(range(1, 11))
    recs = []
    for k in range(2000):
        typ = int(rs.randint(3, 21))
        nop = int(rs.randint(1, 8))
        ops = rs.randint(1, len(t) + 1, size=nop)
        par = rs.rand(int(rs.randint(0, 4)))
        idx = t.append(typ, par, ops, unique=False)
        assert idx == len(t)
        recs.append((idx, typ, par, ops))
    for idx, typ, par, ops in recs:
        t2, p2, o2 = t.get(idx)
        assert t2 == typ and np.array_equal(p2, par) and np.array_equal(o2, ops)


def test_append_unique_dedups(native_built):
    t = InstructionsTable()
    a = t.append(I.I_set, unique=False)
    b = t.append(I.I_set, unique=False)
    assert a != b                      # I_set is never merged (csparse.m:1102-1104)
    s1 = t.append(I.I_sum, [1, -1], [a, b])
    s2 = t.append(I.I_sum, [1, -1], [a, b])
    s3 = t.append(I.I_sum, [1, 1], [a, b])
    s4 = t.append(I.I_sum, [1, -1], [b, a])
    assert s1 == s2 and s3 != s1 and s4 != s1
    assert len(t) == 5
    l1 = t.append(I.I_load, [0.5], [])
    l2 = t.append(I.I_load, [0.5], [])
    l3 = t.append(I.I_load, [-0.5], [])
    assert l1 == l2 != l3


def test_unique_many_random(native_built):
    rs = np.random.RandomState(1)
    t = InstructionsTable()
    t.append_many(I.I_set, None, np.zeros((50, 0), dtype=np.int64), unique=False)
    ops = rs.randint(1, 51, size=(20000, 2))
    idx = t.append_many(I.I_div, None, ops)
    uniq = {tuple(o) for o in ops}
    assert len(t) == 50 + len(uniq)
    seen = {}
    for o, i in zip(map(tuple, ops), idx):
        assert seen.setdefault(o, i) == i


def test_dependencies_and_find(native_built):
    t = InstructionsTable()
    a = t.append(I.I_set, unique=False)
    b = t.append(I.I_set, unique=False)
    c = t.append(I.I_sumprod, [2, 1], [a, b])
    d = t.append(I.I_div, [], [c, a])
    ch, pa = t.dependencies()
    assert sorted(zip(ch.tolist(), pa.tolist())) == sorted([(c, a), (c, b), (d, c), (d, a)])
    assert t.find_by_type(I.I_set).tolist() == [True, True, False, False]
    types, opStart, ops, parStart, pars = t.export()
    assert types.tolist() == [I.I_set, I.I_set, I.I_sumprod, I.I_div]
    assert ops[opStart[2]:opStart[3]].tolist() == [a, b]
    assert pars[parStart[2]:parStart[3]].tolist() == [2.0, 1.0]


def test_rejects_forward_reference(native_built):
    import pytest
    t = InstructionsTable()
    a = t.append(I.I_set, unique=False)
    with pytest.raises(RuntimeError):
        t.append(I.I_sum, [1, 1], [a, 7])
