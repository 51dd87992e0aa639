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
