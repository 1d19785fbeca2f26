import os
import sys

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
GOLDEN = os.path.join(HERE, "golden")
for p in (ROOT, HERE):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


@pytest.fixture(scope="session")
def nttdata():
    """The reference's shipped cases: {id: (n, p, a, b, ab)} (cases 1-3 packed u32, case 0 from text)."""
    cases = {}
    tin = open(os.path.join(GOLDEN, "nttdata_0.in")).read().split()
    n, p = int(tin[0]), int(tin[1])
    a = np.array(tin[2:2 + n], dtype=np.uint32)
    b = np.array(tin[2 + n:2 + 2 * n], dtype=np.uint32)
    ab = np.array(open(os.path.join(GOLDEN, "nttdata_0.out")).read().split(), dtype=np.uint32)
    cases[0] = (n, p, a, b, ab)
    for cid in (1, 2, 3):
        z = np.load(os.path.join(GOLDEN, f"nttdata_{cid}.npz"))
        cases[cid] = (int(z["n"]), int(z["p"]), z["a"], z["b"], z["ab"])
    return cases


@pytest.fixture(scope="session")
def ref_ntt_vectors():
    return np.load(os.path.join(GOLDEN, "ref_ntt_vectors.npz"))


@pytest.fixture(scope="session")
def ref_polymul_vectors():
    return np.load(os.path.join(GOLDEN, "ref_polymul_vectors.npz"))


@pytest.fixture(scope="session")
def crt_vectors():
    return np.load(os.path.join(GOLDEN, "crt_vectors.npz"))


@pytest.fixture(params=["f64", "f64-general", "int"])
def lanes64(request, monkeypatch):
    """The arithmetic policies of the 64-bit path (kernels_u64.hip): primes below 2^51 run as exact integers in
    FP64 lanes by default (below 2^50 in the variant without the closing fold); NTT_AMD_PLAN64_INT (read when a
    plan is created) = 2 keeps every such prime on the general FP64 policy, = 1 on the Montgomery lanes.
    Primes of 51 bits and more take the Montgomery lanes either way."""
    if request.param == "int":
        monkeypatch.setenv("NTT_AMD_PLAN64_INT", "1")
    elif request.param == "f64-general":
        monkeypatch.setenv("NTT_AMD_PLAN64_INT", "2")
    else:
        monkeypatch.delenv("NTT_AMD_PLAN64_INT", raising=False)
    return request.param


@pytest.fixture(autouse=True)
def crt_four_prime_route(monkeypatch):
    """ntt_crt_multiply[_batched] sends a modulus that is itself an NTT-friendly prime for the length (the
    reference's 1337006139375617 is) through the direct 64-bit plan by default (crt.cpp).  The CRT tests exist to
    check the FOUR-PRIME path -- residue products, Garner merge (main-openmp.cc:424-485) -- so every test runs with
    NTT_AMD_CRT_DIRECT=0 (read per call) unless it asks for the default route itself
    (tests/test_gpu_round3.py::test_crt_multiply_routes_to_direct_plan)."""
    monkeypatch.setenv("NTT_AMD_CRT_DIRECT", "0")
