import os
os.environ.setdefault("SLS_LAB", "1")      # the tests steer kernel routing through the diagnostic knobs: lab mode
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle")):
    if p not in sys.path:
        sys.path.insert(0, p)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    _ensure_built()


def _ensure_built():
    """The built artefacts are git-ignored and travel to the GPU box as they are.  `make check` compares a content hash of
    the sources with the one recorded at link time (file times do not survive the copy), so an edited kernel can never be
    tested against a stale binary: on a mismatch the library is rebuilt here (hipcc cross-compiles gfx950 without a GPU).
    The oracle's C port is plain gcc (1 s): always handed to make."""
    import subprocess
    csrc = os.path.join(ROOT, "systemlevelcontrol.jl_amd", "csrc")
    if subprocess.call(["make", "-s", "-C", csrc, "check"]) != 0:
        # -B: after the copy to the GPU box the objects may LOOK newer than the sources; a plain make would then do nothing
        subprocess.check_call(["make", "-B", "-C", csrc])
        if subprocess.call(["make", "-s", "-C", csrc, "check"]) != 0:
            raise pytest.UsageError("libsls_mi355x.so still does not match its sources after a forced rebuild")
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle")])


@pytest.fixture(scope="session")
def slc():
    import slc_amd
    return slc_amd


@pytest.fixture(scope="session")
def oracle():
    import sls_oracle
    return sls_oracle


@pytest.fixture(scope="session")
def readme(slc):
    P, S, meta = slc.workloads.make_workload("readme_chain")
    return P, S, meta


@pytest.fixture(scope="session")
def golden_readme():
    return np.load(os.path.join(GOLDEN, "readme_chain_phi.npz"))


@pytest.fixture(scope="session")
def gpu_ctx(slc):
    """A context on device 0.  Fails (does not skip) when the HIP path is unavailable:
    GPU tests must never pass on a fallback."""
    ctx = slc.Context([0])
    yield ctx
    ctx.close()


def split_vals(flat, nnz_list):
    out, o = [], 0
    for n in nnz_list:
        out.append(np.asarray(flat[o:o + n])); o += n
    assert o == len(flat)
    return out


def flat_phi(Phi, masks):
    """Φ[t] (scipy CSC) → concatenated values in the mask's CSC order."""
    import scipy.sparse as sp
    out = []
    for F, Sm in zip(Phi, masks):
        Sm = sp.csc_matrix(Sm); Sm.sort_indices()
        rows = Sm.indices
        cols = np.repeat(np.arange(Sm.shape[1]), np.diff(Sm.indptr))
        out.append(np.asarray(sp.csc_matrix(F)[rows, cols]).ravel())
    return np.concatenate(out)
