"""build_class(try_flags=...): compiler flags that are worth having but not worth failing for (LLVM's experimental
scheduling strategies crash clang on some classes) are tried first and dropped when hipcc fails with them; what was used
is recorded next to the library."""
import os


def test_try_flags_fall_back_when_the_compiler_rejects_them(native_built, tmp_path):
    import __graft_entry__ as g
    from tenscalc_amd.problems.small import small_ls
    spec = small_ls("free", folder=str(tmp_path))
    kw = dict(g.VARIANT_KW["strict"], tag="tf")
    so = spec.build(try_flags=("-mllvm", "-no-such-llvm-option-xyz"), **kw)
    assert os.path.exists(so)
    assert open(so[:-3] + ".flags").read().strip() == ""            # dropped: hipcc refused the flag
    so2 = spec.build(try_flags=("-DTC_SOME_HARMLESS_DEFINE=1",), **dict(kw, tag="tf2"))
    assert open(so2[:-3] + ".flags").read().strip() == "-DTC_SOME_HARMLESS_DEFINE=1"


def test_try_flags_alternatives_are_tried_in_order(native_built, tmp_path):
    """a tuple of tuples: the first alternative the compiler accepts is used (the w8x classes: LLVM's experimental
    iterative-ilp strategy, else the generic -misched=ilpmax, else the default scheduling)"""
    import __graft_entry__ as g
    from tenscalc_amd.problems.small import small_ls
    spec = small_ls("free", folder=str(tmp_path))
    kw = dict(g.VARIANT_KW["strict"], tag="tf3")
    so = spec.build(try_flags=(("-mllvm", "-no-such-llvm-option-xyz"), ("-DTC_SECOND_CHOICE=1",)), **kw)
    assert open(so[:-3] + ".flags").read().strip() == "-DTC_SECOND_CHOICE=1"
    so = spec.build(try_flags=(("-mllvm", "-no-such-llvm-option-xyz"), ("-mllvm", "-nor-this-one")), **dict(kw, tag="tf4"))
    assert os.path.exists(so) and open(so[:-3] + ".flags").read().strip() == ""
    alts = g.VARIANT_KW["w8x"]["try_flags"]
    assert alts[0] == ("-mllvm", "-amdgpu-sched-strategy=iterative-ilp") and alts[1] == ("-mllvm", "-misched=ilpmax")
