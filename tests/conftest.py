import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def native_built():
    import __graft_entry__ as g
    g.build_native_table()
    from oracle import build as ob
    ob.build_oracle()
    return True


@pytest.fixture(scope="session")
def sls_spec(native_built):
    import __graft_entry__ as g
    from tenscalc_amd.problems.sls import sls
    return sls(folder=g.GENERATED)


@pytest.fixture(scope="session")
def sls_tape_path(sls_spec, tmp_path_factory):
    from tenscalc_amd.tapeio import save_tape
    p = str(tmp_path_factory.mktemp("tapes") / "sls.tape")
    save_tape(sls_spec.tape, p)
    return p


@pytest.fixture()
def tape_oracle(native_built, tmp_path):
    """factory: Tape -> Oracle"""
    from tenscalc_amd.tapeio import save_tape
    from oracle.tc_oracle import Oracle
    made = []

    def make(tape, name="t"):
        p = str(tmp_path / f"{name}{len(made)}.tape")
        save_tape(tape, p)
        o = Oracle(p)
        made.append(o)
        return o
    return make
