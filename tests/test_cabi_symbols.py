"""CPU: every generated class library loads and exports every symbol include/tenscalc_hip.h
declares, plus the reference's own per-class C symbols (no compute calls: no GPU here)."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def libs():
    import __graft_entry__ as g
    return g.build()


def _declared(header):
    txt = open(header).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(tc_[a-z_0-9]+)\s*\(", txt)))


def test_batched_abi_exported(libs):
    names = _declared(os.path.join(ROOT, "include", "tenscalc_hip.h"))
    assert len(names) >= 25
    for tag, so in libs.items():
        L = ctypes.CDLL(so, mode=ctypes.RTLD_LOCAL)
        for n in names:
            assert hasattr(L, n), f"{tag}: {n} missing"


def test_reference_symbols_exported(libs):
    for tag, so in libs.items():
        L = ctypes.CDLL(so, mode=ctypes.RTLD_LOCAL)
        L.tc_classname.restype = ctypes.c_char_p
        cls = L.tc_classname().decode()
        L.tc_input_name.restype = ctypes.c_char_p
        inputs = [L.tc_input_name(k).decode() for k in range(L.tc_num_inputs())]
        assert len(inputs) >= 2          # qsaddle has no parameters: two players' variables only
        # writeCfunctionpergroup.m:524,607-635; ipmPD_CSsolver.c:132 (unprefixed name)
        solver = "ipmPDeq_CSsolver" if tag.startswith("games_") else "ipmPD_CSsolver"     # ipmPDeq_CSsolver.c:122 for equilibrium classes
        for n in [f"{cls}_set_{i}" for i in inputs] + [f"{cls}_getOutputs", solver]:
            assert hasattr(L, n), f"{tag}: {n} missing"


def test_introspection_matches_spec(libs, sls_spec):
    from tenscalc_amd.runtime import _bind
    L = _bind(libs["sls_fast"])
    assert L.tc_classname().decode() == sls_spec.classname
    ins = [(L.tc_input_name(k).decode(), L.tc_input_numel(k), L.tc_input_is_variable(k))
           for k in range(L.tc_num_inputs())]
    assert ins == [("A", 800, 0), ("b", 100, 0), ("x", 8, 1)]
    outs = [(L.tc_output_name(k).decode(), L.tc_output_numel(k), L.tc_output_offset(k))
            for k in range(L.tc_num_outputs())]
    assert outs == [("J", 1, 0), ("x", 8, 1)]
    assert L.tc_define_int(b"nU") == 8 and L.tc_define_int(b"nF") == 16 and L.tc_define_int(b"nG") == 0
    assert L.tc_define_double(b"desiredDualityGap") == 1e-5
    assert L.tc_flops_static(b"groups") == sls_spec.tape.nGroups
    assert L.tc_probe_numel(b"getGrad__") == 8 and L.tc_probe_numel(b"nope") == -1


def test_instruction_table_abi():
    txt = open(os.path.join(ROOT, "include", "tc_instructions_table.h")).read()
    names = sorted(set(re.findall(r"\b(tcit_[a-z_]+)\s*\(", txt)))
    L = ctypes.CDLL(os.path.join(ROOT, "tenscalc_amd", "libtcitable.so"))
    for n in names:
        assert hasattr(L, n), n


def test_missing_library_fails_loudly(tmp_path):
    from tenscalc_amd.runtime import HipSolver, TensCalcHipError
    with pytest.raises(TensCalcHipError):
        HipSolver(str(tmp_path / "libnope.so"))
