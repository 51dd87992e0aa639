"""CPU: the tape container round-trips (save -> load -> Tape) and a class built from the file alone
emits the same per-instruction code as one built from the live front-end objects."""
import numpy as np


def test_roundtrip_and_build_from_file(sls_spec, sls_tape_path, tmp_path):
    from tenscalc_amd.hip_emitter import emit_class
    from tenscalc_amd.tapeio import load_records
    from tenscalc_amd.tools.build_class import spec_from_tape
    r = load_records(sls_tape_path)
    T = sls_spec.tape
    assert np.array_equal(r["types"], T.types) and np.array_equal(r["ops"], T.ops)
    assert np.array_equal(r["pars"], T.pars) and int(r["nGroups"][0]) == T.nGroups
    spec2 = spec_from_tape(sls_tape_path, folder=str(tmp_path))
    assert spec2.classname == sls_spec.classname
    assert spec2.defines["nU"] == 8 and spec2.defines["desiredDualityGap"] == 1e-5
    assert [n for n, _ in spec2.outputs] == ["J", "x"]
    a, _ = emit_class(sls_spec, fp_contract="off", roll=False)
    code = sls_spec.code
    try:
        sls_spec.code = None
        b0, _ = emit_class(sls_spec, fp_contract="off", roll=False)
    finally:
        sls_spec.code = code
    b, _ = emit_class(spec2, fp_contract="off", roll=False)
    dev = lambda s: s[:s.index("static const int TC_NIN")]  # noqa: E731  (device code: groups, entry points, kernels)
    assert dev(b) == dev(b0)                # same text from the file as from the live tape
    assert a.count("tc_grp") == b.count("tc_grp")


def test_cli_builds_library(sls_tape_path, tmp_path):
    import ctypes
    from tenscalc_amd.tools.build_class import main
    so = main([sls_tape_path, "--strict", "--out", str(tmp_path)])
    L = ctypes.CDLL(so, mode=ctypes.RTLD_LOCAL)
    assert hasattr(L, "tc_solve") and hasattr(L, "ipmPD_CSsolver") and hasattr(L, "tmp_minslsc2_set_A")


def test_cg_files_roundtrip(sls_spec, tmp_path, tape_oracle):
    """export to the documented .cg/.cgc/.cgio/.cgs container and back: same instructions, same
    dependency groups, same solve."""
    from tenscalc_amd.cgformat import load_cg, save_cg
    from tenscalc_amd.problems.sls import sls_data
    T = sls_spec.tape
    base = str(tmp_path / "sls")
    save_cg(T, base)
    T2 = load_cg(base)
    assert T2.n == T.n and np.array_equal(T2.types, T.types)
    assert np.array_equal(T2.ops, T.ops) and np.array_equal(T2.pars, T.pars)
    assert T2.nGroups == T.nGroups and np.array_equal(T2.group, T.group)
    assert [e["name"] for e in T2.gets] == [e["name"] for e in T.gets]
    assert [(e["name"], e["children"]) for e in T2.sets] == [(e["name"], e["children"]) for e in T.sets]
    assert T2.defines["nF"] == 16 and T2.defines["gradTolerance"] == 1e-4
    o1, o2 = tape_oracle(T, "a"), tape_oracle(T2, "b")
    c = sls_spec.classname
    A, b, x0 = sls_data(3)
    r = []
    for o in (o1, o2):
        o.set(c + "_set_A", A); o.set(c + "_set_b", b); o.set(c + "_set_x", x0)
        r.append((o.solve(1.0, 30), o.get(c + "_getOutputs")))
    assert r[0][0] == r[1][0] and np.array_equal(r[0][1], r[1][1])


def test_reference_dialect_cg_files(sls_spec, tmp_path, tape_oracle):
    """a file as the reference's saveScalarized writes it (lib/@csparse/saveScalarized.m): int32 index words, the
    registry's function type ids (CGregistration.m), scalars2ftensor / ftensor2scalars nodes around gets and sets,
    1-based ids in the set/get io records, no descriptions, no solver defines -- loads (the defines given
    separately), yields the same instructions and groups, and solves to the same bits."""
    import struct
    from tenscalc_amd.cgformat import TYPE_OF_OPCODE, load_cg, save_cg
    from tenscalc_amd import instructions as I
    from tenscalc_amd.problems.sls import sls_data
    T = sls_spec.tape
    base = str(tmp_path / "ref")
    save_cg(T, base, word=32, extensions="opcodes")       # sls has I_heaviside (getHessInertia__): extension id
    raw = open(base + ".cg", "rb").read()
    assert raw[0] == 32
    assert TYPE_OF_OPCODE[I.I_sum] == 20001 and TYPE_OF_OPCODE[I.I_sumprod] == 20002 and TYPE_OF_OPCODE[I.I_div] == 20005
    assert TYPE_OF_OPCODE[I.I_DDatan] == 20032
    words = struct.unpack("<%di" % ((len(raw) - 1) // 4), raw[1:1 + 4 * ((len(raw) - 1) // 4)])
    assert 30001 in words and 30002 in words                # s2ft / ft2s nodes
    sraw = open(base + ".cgs", "rb").read()
    assert b"|" not in sraw                                  # no descriptions, no defines
    T2 = load_cg(base, defines=sls_spec.defines)
    assert T2.n == T.n and np.array_equal(T2.types, T.types)
    assert np.array_equal(T2.ops, T.ops) and np.array_equal(T2.pars, T.pars)
    assert T2.nGroups == T.nGroups and np.array_equal(T2.group, T.group)
    assert [(e["name"], e["external"], len(e["ins"])) for e in T2.sets] == [(e["name"], e["external"], len(e["ins"])) for e in T.sets]
    go = T2.entry(sls_spec.classname + "_getOutputs")[2]["outputs"]
    assert [o["size"] for o in go] == [(), (8,)]             # shapes recovered from the subscripts
    o1, o2 = tape_oracle(T, "a"), tape_oracle(T2, "b")
    c = sls_spec.classname
    A, b, x0 = sls_data(5)
    r = []
    for o in (o1, o2):
        o.set(c + "_set_A", A); o.set(c + "_set_b", b); o.set(c + "_set_x", x0)
        r.append((o.solve(1.0, 30), o.get(c + "_getOutputs")))
    assert r[0][0] == r[1][0] and np.array_equal(r[0][1], r[1][1])


def test_zero_and_componentwise_opcodes(tmp_path, tape_oracle):
    """I_zero (49) and I_componentwise (50, C text with an operand hole: instructionsTableUTHash.c:1333-1345) in a
    hand-made tape: the oracle evaluates them, the HIP printer prints them, the schedulers of the table-driven
    back-ends map the plain calls to opcodes, and they survive the .cg container."""
    from tenscalc_amd import instructions as I
    from tenscalc_amd.cgformat import load_cg, save_cg
    from tenscalc_amd.csparse import Tape
    from tenscalc_amd.hip_emitter import Printer
    cw = [float(ord(ch)) for ch in "fabs("] + [0.0] + [float(ord(")"))]
    #        1,2,3: set x[3]     4: I_zero     5: fabs(x1)          6: x2 + zero            7: -x3 + fabs(x1)
    types = [I.I_set] * 3 + [I.I_zero, I.I_componentwise, I.I_sum, I.I_sum]
    ops = [[], [], [], [], [1], [2, 4], [3, 5]]
    pars = [[], [], [], [], cw, [1, 1], [-1, 1]]
    opStart = np.cumsum([0] + [len(o) for o in ops]); parStart = np.cumsum([0] + [len(p_) for p_ in pars])
    mk = lambda: Tape.from_instructions(                         # noqa: E731
        types, opStart, [j for o in ops for j in o], parStart, [v for p_ in pars for v in p_],
        [dict(name="t_set_x", varname="x", size=(3,), ins=[1, 2, 3], external=True)],
        [dict(name="t_get_y", outputs=[dict(name="y", size=(4,), ins=[4, 5, 6, 7])])], [], {})
    T = mk()
    o = tape_oracle(T, "t")
    o.set("t_set_x", [-1.5, 2.0, 3.0])
    assert np.array_equal(o.get("t_get_y"), [0.0, 1.5, 2.0, -1.5])
    assert I.componentwise_text(cw) == "fabs(%s)" and I.componentwise_opcode(cw) == I.I_abs
    assert Printer.op_text(I.I_componentwise, tuple(cw), ["MM(3)"]) == "(fabs((MM(3))))"
    assert Printer.op_text(I.I_zero, (), []) == "0.0"
    base = str(tmp_path / "z")
    save_cg(T, base, word=32, extensions="opcodes")
    T2 = load_cg(base, defines={})
    assert np.array_equal(T2.types, T.types) and np.array_equal(T2.pars, T.pars) and np.array_equal(T2.ops, T.ops)
    o2 = tape_oracle(T2, "t2")
    o2.set("t_set_x", [-1.5, 2.0, 3.0])
    assert np.array_equal(o2.get("t_get_y"), [0.0, 1.5, 2.0, -1.5])
