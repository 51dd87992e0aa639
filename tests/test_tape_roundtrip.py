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
