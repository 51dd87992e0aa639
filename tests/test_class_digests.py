"""The class libraries are prebuilt, git-ignored binaries that travel with the push (tenscalc_amd/generated/@*/lib*.so).
What ties a shipped library to the TRACKED sources is the pair of digests written next to it when it was compiled:
<cls><tag>.key (tape + every option + the generator's sources) and <cls><tag>.stamp (the emitted HIP source + compiler
flags + every file the source includes).  Recomputing both here, from the tracked tree alone, and comparing them with
the recorded ones shows that the binary that will be loaded on the GPU box was built from exactly these sources."""
import os

import pytest

import __graft_entry__ as g

# every library whose tape is cheap to rebuild (the big tapes -- l1l2, mpc_quadcopter -- take minutes to generate; their
# libraries are covered by build(): a stale key there triggers a rebuild, and the GPU parity tests load what was built)
CHEAP = ("sls_", "sls37_", "small_", "slsadd_", "dcmotor_strict", "dcmotor_interp", "flops30_", "zoo_strict", "zoo_interp", "sens_",
         "games_")


def _shipped():
    out = []
    for tag, make, kw in g.solver_variants():
        if tag.startswith(CHEAP):
            out.append((tag, make, kw))
    return out


def test_recorded_keys_match_the_tracked_sources(native_built):
    from tenscalc_amd import tcalculus as tc
    checked = 0
    for tag, make, kw in _shipped():
        tc.clear()
        spec = make()
        sfx = "_" + kw["tag"] if kw.get("tag") else ""
        d = os.path.join(spec.folder, "@" + spec.classname)
        so, key = os.path.join(d, "lib%s%s.so" % (spec.classname, sfx)), os.path.join(d, "%s%s.key" % (spec.classname, sfx))
        assert os.path.exists(so) and os.path.exists(key), "library of %s not built" % tag
        assert open(key).read().strip() == spec.build(key_only="quick", **kw), "%s: lib is not what the tracked sources build" % tag
        checked += 1
    assert checked >= 40


@pytest.mark.parametrize("tag", ["sls_strict", "sls_w8x", "dcmotor_interp", "small_eq_strict", "flops30_strict_stats"])
def test_recorded_source_digest_matches_a_fresh_emission(native_built, tag):
    """the slower half: emit the class source again and hash it with the flags and its include closure -- equal to the
    .stamp written when the shipped library was compiled"""
    from tenscalc_amd import tcalculus as tc
    tc.clear()
    _, make, kw = [v for v in g.solver_variants() if v[0] == tag][0]
    spec = make()
    sfx = "_" + kw["tag"] if kw.get("tag") else ""
    d = os.path.join(spec.folder, "@" + spec.classname)
    quick, digest = spec.build(key_only=True, **kw)
    assert open(os.path.join(d, "%s%s.key" % (spec.classname, sfx))).read().strip() == quick
    assert open(os.path.join(d, "%s%s.stamp" % (spec.classname, sfx))).read().strip() == digest
    flags = open(os.path.join(d, "lib%s%s.flags" % (spec.classname, sfx))).read().split()
    tf = kw.get("try_flags", ())
    alts = [tuple(a) for a in tf] if (tf and isinstance(tf[0], (tuple, list))) else [tuple(tf)]
    assert tuple(flags) in alts + [()]        # one of the alternatives, or none where the compiler failed with all of them
