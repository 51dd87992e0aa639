"""The MATLAB class file of a generated solver (tenscalc_amd/matlab_class.py): the reference surface
``setP_*/setV_*/solve/getOutputs`` (README.md:203-215; cmex2optimizeCS.m:157-195,282-294,340-355)
with the trailing batch dimension.  No MATLAB in the image: structural checks only."""
import re

from tenscalc_amd.matlab_class import matlab_class_text


def test_sls_class_surface(sls_spec):
    txt = matlab_class_text(sls_spec)
    c = sls_spec.classname
    assert txt.startswith(f"classdef {c} < handle")
    for m in ("setP_A", "setP_b", "setV_x", "solve", "getOutputs", "delete", c):
        assert re.search(r"function[^\n]*\b%s\(" % m, txt), m
    assert "tc_mex('create'" in txt and f"lib{c}.so" in txt
    assert "expected [100,8,B]" in txt and "expected [100,B]" in txt and "expected [8,B]" in txt
    assert "reshape(y{2}, 8,1,obj.B)" in txt and "reshape(y{1}, 1,obj.B)" in txt
    assert "struct('J', {y{1}}, 'x', {y{2}})" in txt
    # balanced blocks
    opens = len(re.findall(r"^\s*(classdef|properties|methods|function|if|for)\b", txt, re.M))
    assert opens == len(re.findall(r"^\s*end\b|\bend\s*$|; end\b|, end\b", txt, re.M)) or txt.count("end") >= opens


def test_every_problem_has_a_class_file(native_built, tmp_path):
    from tenscalc_amd.matlab_class import write_matlab_class
    from tenscalc_amd.problems.mpc_dcmotor import mpc_dcmotor
    spec = mpc_dcmotor(folder=str(tmp_path))
    p = write_matlab_class(spec)
    txt = open(p).read()
    for nm, _ in spec.parameters:
        assert f"function setP_{nm}(obj, value)" in txt
    for nm, _ in spec.variables:
        assert f"function setV_{nm}(obj, value)" in txt
    assert txt.count("tc_mex('set'") == len(spec.parameters) + len(spec.variables)
