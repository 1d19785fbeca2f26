"""CPU-only tests of the product's host side: the C-ABI library loads and exports every symbol
include/ntt_amd.h declares, the harness file helpers keep the reference's text formats, the host
number theory used for plan creation is right, and compute entry points fail loudly (no CPU
fallback) when there is no GPU.  No kernel is launched here.
"""
import ctypes as C
import os
import re
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def nt():
    import ntt_amd
    return ntt_amd


def declared_functions():
    text = open(os.path.join(ROOT, "include", "ntt_amd.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    names = re.findall(r"\b(poly_multiply|ntt_[a-z0-9_]+)\s*\(", text)
    return sorted(set(names))


def test_library_exports_every_declared_symbol(nt):
    lib = C.CDLL(os.path.join(ROOT, "ntt_amd", "libntt_amd.so"))
    decl = declared_functions()
    assert "poly_multiply" in decl and "ntt_crt_multiply" in decl and len(decl) >= 19
    for name in decl:
        assert getattr(lib, name) is not None, name
    assert sorted(nt.EXPORTED_SYMBOLS) == decl


def test_product_does_not_link_or_import_the_oracle():
    # the oracle is test infrastructure: nothing under ntt_amd/ may reference it
    out = subprocess.run(["readelf", "-d", os.path.join(ROOT, "ntt_amd", "libntt_amd.so")],
                         capture_output=True, text=True).stdout
    assert "oracle" not in out
    for dirpath, _, files in os.walk(os.path.join(ROOT, "ntt_amd")):
        if "build" in dirpath:
            continue
        for f in files:
            if f.endswith((".py", ".cpp", ".hpp", ".hip", ".h")):
                src = open(os.path.join(dirpath, f)).read()
                assert "ntt_oracle" not in src and "import orc" not in src, f


def test_read_check_write_roundtrip(nt, golden_dir, tmp_path, capfd):
    # the reference's tiny fixture, verbatim text formats
    data = tmp_path / "nttdata"
    data.mkdir()
    (data / "0.in").write_bytes(open(os.path.join(golden_dir, "nttdata_0.in"), "rb").read())
    (data / "0.out").write_bytes(open(os.path.join(golden_dir, "nttdata_0.out"), "rb").read())
    n, p, a, b = nt.read_case(data, 0)
    assert (n, p) == (4, 7340033)
    assert a.tolist() == [4, 1, 5, 2] and b.tolist() == [1, 5, 5, 4]
    good = np.array([4, 21, 30, 48, 39, 30, 8], dtype=np.uint64)
    assert nt.check_case(data, 0, good) is True
    assert "多项式乘法结果正确" in capfd.readouterr().out
    bad = good.copy()
    bad[3] += 1
    assert nt.check_case(data, 0, bad) is False
    assert "多项式乘法结果错误" in capfd.readouterr().out
    out = tmp_path / "files"
    out.mkdir()
    nt.write_case(out, 0, good)
    assert (out / "0.out").read_text() == "4\n21\n30\n48\n39\n30\n8\n"       # fWrite: one value per line


def test_read_case_large_and_errors(nt, nttdata, tmp_path):
    n, p, a, b, ab = nttdata[1]
    d = tmp_path
    with open(d / "1.in", "w") as f:       # the reference's layout: each value followed by a space
        f.write(f"{n} {p}\n" + "".join(f"{v} " for v in a) + "\n" + "".join(f"{v} " for v in b))
    with open(d / "1.out", "w") as f:
        f.write("".join(f"{v} " for v in ab))
    n2, p2, a2, b2 = nt.read_case(d, 1)
    assert (n2, p2) == (n, p) and np.array_equal(a2, a) and np.array_equal(b2, b)
    assert nt.check_case(d, 1, ab.astype(np.uint64)) is True
    # 51-bit modulus values survive the loader (main-openmp.cc u64 overload)
    with open(d / "4.in", "w") as f:
        f.write("2 1337006139375617\n1337006139375616 5 \n7 1337006139375000 ")
    n4, p4, a4, b4 = nt.read_case(d, 4)
    assert p4 == 1337006139375617 and a4.tolist() == [1337006139375616, 5] and b4[1] == 1337006139375000
    with pytest.raises(nt.NttError) as e:
        nt.read_case(d, 9)                 # missing file: an error, not a spurious verdict
    assert e.value.code == nt.ERR_IO
    with open(d / "5.in", "w") as f:
        f.write("3 7340033\n1 2 3 \n4 5 ")  # truncated b[]
    with pytest.raises(nt.NttError):
        nt.read_case(d, 5)
    with pytest.raises(nt.NttError):
        nt.check_case(d, 9, np.zeros(3, dtype=np.uint64))


def test_packed_case_roundtrip(nt, golden_dir, nttdata, tmp_path, capfd):
    """Packed binary cases (ntt_pack_case / ntt_read_case_bin / ntt_check_case_bin / ntt_write_case_bin):
    same content as the reference's text files, no decimal parsing."""
    import struct
    d = tmp_path
    (d / "0.in").write_bytes(open(os.path.join(golden_dir, "nttdata_0.in"), "rb").read())
    (d / "0.out").write_bytes(open(os.path.join(golden_dir, "nttdata_0.out"), "rb").read())
    nt.pack_case(d, 0)
    raw = (d / "0.inb").read_bytes()
    assert raw[:4] == b"NTTB" and struct.unpack("<IIIQQ", raw[4:32]) == (1, 1, 4, 4, 7340033)
    assert len(raw) == 32 + 2 * 4 * 4
    n, p, a, b = nt.read_case_bin(d, 0)
    assert (n, p) == (4, 7340033) and a.dtype == np.uint32
    assert a.tolist() == [4, 1, 5, 2] and b.tolist() == [1, 5, 5, 4]
    good = np.array([4, 21, 30, 48, 39, 30, 8], dtype=np.uint32)
    assert nt.check_case_bin(d, 0, good) is True
    assert "多项式乘法结果正确" in capfd.readouterr().out
    bad = good.copy()
    bad[6] ^= 1
    assert nt.check_case_bin(d, 0, bad) is False
    assert "多项式乘法结果错误" in capfd.readouterr().out
    # a full-size case: packed files equal the golden arrays and are about half the size of the text
    n1, p1, a1, b1, ab1 = nttdata[1]
    with open(d / "1.in", "w") as f:
        f.write(f"{n1} {p1}\n" + "".join(f"{v} " for v in a1) + "\n" + "".join(f"{v} " for v in b1))
    with open(d / "1.out", "w") as f:
        f.write("".join(f"{v} " for v in ab1))
    nt.pack_case(d, 1)
    n2, p2, a2, b2 = nt.read_case_bin(d, 1)
    assert (n2, p2) == (n1, p1) and np.array_equal(a2, a1) and np.array_equal(b2, b1)
    assert nt.check_case_bin(d, 1, ab1.astype(np.uint32)) is True
    assert os.path.getsize(d / "1.inb") == 32 + 8 * n1 < os.path.getsize(d / "1.in") * 0.6
    # 64-bit coefficients when p needs them; inputs without an expected output pack too
    with open(d / "4.in", "w") as f:
        f.write("2 1337006139375617\n1337006139375616 5 \n7 1337006139375000 ")
    nt.pack_case(d, 4)
    assert not (d / "4.outb").exists()
    n4, p4, a4, b4 = nt.read_case_bin(d, 4)
    assert p4 == 1337006139375617 and a4.dtype == np.uint64 and a4.tolist() == [1337006139375616, 5]
    out = np.array([1, 2, 1337006139375616], dtype=np.uint64)
    nt.write_case_bin(d, 4, out, p4)
    assert nt.check_case_bin(d, 4, out) is True
    with pytest.raises(nt.NttError):
        nt.check_case_bin(d, 4, out.astype(np.uint32))           # width differs from the file
    with pytest.raises(nt.NttError):
        nt.read_case_bin(d, 9)
    (d / "7.inb").write_bytes(b"NTTX" + raw[4:])
    with pytest.raises(nt.NttError):
        nt.read_case_bin(d, 7)


def test_no_cpu_fallback(nt):
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    with pytest.raises(nt.NttError) as e:
        nt.Context(0)
    assert e.value.code == nt.ERR_HIP
    # the reference entry point cannot report errors: it must abort, not compute on the CPU
    code = ("import numpy as np, ntt_amd; "
            "ntt_amd.poly_multiply(np.array([1,2],dtype=np.int32), np.array([3,4],dtype=np.int32), 7340033)")
    r = subprocess.run([sys.executable, "-c", code], cwd=ROOT, capture_output=True, text=True)
    assert r.returncode != 0 and "poly_multiply failed" in r.stderr


def test_host_number_theory_probe(tmp_path):
    """modarith.hpp on the host: Montgomery/Barrett constants, primality, generators -- compiled into
    a tiny g++ program and compared with Python integers."""
    src = tmp_path / "probe.cpp"
    src.write_text(r'''
#include <cstdio>
#include "modarith.hpp"
using namespace ntt;
int main() {
    const uint32_t ps[] = {7340033u, 104857601u, 469762049u, 998244353u, 1004535809u, 1224736769u, 2013265921u, 17u};
    for (uint32_t p : ps) {
        ModParams m = make_mod_params(p);
        std::printf("%u %u %u %u %u %u %u %u %d\n", p, h_primitive_root(p), m.nprime, m.r1, m.r2, m.kbits, m.mu,
                    Mont32::make_tw(12345u % p, m), (int)h_is_prime(p));
        // device-side formulas evaluated on the host
        uint32_t x = 0xfffffff0u, w = p - 2;
        std::printf("%u %u %u\n", Mont32::mul_tw(x, Mont32::make_tw(w, m), m), Barrett32::mul_tw(x, Barrett32::make_tw(w, m), m),
                    Barrett32::mul_raw(p - 1, p - 2, m));
    }
    std::printf("%d %d %d\n", (int)h_is_prime(998244351u), (int)h_is_prime(1u), (int)h_is_prime(2147483647u));
    return 0;
}''')
    exe = tmp_path / "probe"
    subprocess.run(["g++", "-O1", "-std=c++17", "-I", os.path.join(ROOT, "ntt_amd", "csrc"), str(src), "-o", str(exe)],
                   check=True)
    lines = subprocess.run([str(exe)], capture_output=True, text=True, check=True).stdout.split("\n")
    R = 1 << 32
    ps = [7340033, 104857601, 469762049, 998244353, 1004535809, 1224736769, 2013265921, 17]
    for i, p in enumerate(ps):
        f = list(map(int, lines[2 * i].split()))
        assert f[0] == p
        g = f[1]
        phi = p - 1
        fac = {q for q in range(2, 1 << 16) if phi % q == 0 and all(q % r for r in range(2, int(q ** .5) + 1))}
        assert all(pow(g, phi // q, p) != 1 for q in fac), p
        if p in (7340033, 104857601, 469762049, 998244353, 1004535809, 1224736769):
            assert g == 3                     # the reference hard-codes g = 3 (main-montgomery.cc:257)
        assert (f[2] * p + 1) % R == 0        # nprime = -p^-1 mod 2^32
        assert f[3] == R % p and f[4] == R * R % p
        assert f[5] == p.bit_length() and f[6] == (1 << (2 * f[5])) // p
        assert f[7] == (12345 % p) * R % p and f[8] == 1
        v = list(map(int, lines[2 * i + 1].split()))
        x, w = 0xfffffff0, p - 2
        assert v[0] % p == x * w % p and v[0] < 2 * p
        assert v[1] % p == x * w % p and v[1] < 2 * p
        assert v[2] == (p - 1) * (p - 2) % p
    assert lines[2 * len(ps)].split() == ["0", "0", "1"]


def test_header_is_plain_c(tmp_path):
    """include/ntt_amd.h is the boundary a cgo / JNI / ctypes binding consumes: it must compile as C99 (and as
    C++) on its own, and a C program must link against the library with nothing but the header."""
    hdr = os.path.join(ROOT, "include", "ntt_amd.h")
    subprocess.run(["gcc", "-std=c99", "-Wall", "-Wextra", "-pedantic", "-Werror", "-fsyntax-only", "-x", "c", hdr], check=True)
    subprocess.run(["g++", "-std=c++11", "-Wall", "-Werror", "-fsyntax-only", "-x", "c++", hdr], check=True)
    src = tmp_path / "probe.c"
    src.write_text(
        '#include <stdio.h>\n#include "ntt_amd.h"\n'
        'int main(void) {\n'
        '    uint32_t p = 0;\n'
        '    if (ntt_crt_prime(3, &p) != NTT_OK || p != 998244353u) return 1;\n'
        '    if (ntt_crt_prime(4, &p) != NTT_ERR_INVALID) return 2;          /* status codes, not exceptions */\n'
        '    if (ntt_last_error()[0] == 0) return 3;\n'
        '    printf("ok\\n");\n'
        '    return 0;\n}\n')
    exe = tmp_path / "probe"
    libdir = os.path.join(ROOT, "ntt_amd")
    subprocess.run(["gcc", "-std=c99", "-I", os.path.join(ROOT, "include"), str(src), "-L", libdir, "-lntt_amd",
                    "-Wl,-rpath," + libdir, "-o", str(exe)], check=True)
    r = subprocess.run([str(exe)], capture_output=True, text=True)
    assert r.returncode == 0 and r.stdout.strip() == "ok", (r.returncode, r.stderr)


def test_no_exception_crosses_the_c_abi(tmp_path):
    """A C++ exception inside an entry point comes back as a status code (function-try-blocks, NTT_CATCH_STATUS):
    under an address-space limit the 8 GiB operand vectors of a case header claiming n = 2^30 cannot be
    allocated; std::bad_alloc must surface as NTT_ERR_NOMEM, not terminate the process."""
    import subprocess
    import sys
    (tmp_path / "7.in").write_text("1073741824 7340033\n1 2 3\n")
    code = r'''
import resource, sys
sys.path.insert(0, %r)
import ntt_amd
vm = [int(l.split()[1]) for l in open("/proc/self/status") if l.startswith("VmSize")][0] * 1024
resource.setrlimit(resource.RLIMIT_AS, (vm + (1 << 30), vm + (1 << 30)))
try:
    ntt_amd.pack_case(%r, 7)
except ntt_amd.NttError as e:
    print("status", e.code)
else:
    print("no error")
''' % (os.path.dirname(os.path.dirname(os.path.abspath(__file__))), str(tmp_path))
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr[-2000:]
    assert r.stdout.strip() == "status 4"          # NTT_ERR_NOMEM


def test_bench_prime_search_matches_trial_division():
    """bench.py's RNS prime list (Miller-Rabin) is the one the GPU tests build by trial division (SURVEY 8d, configs[3]:
    the 32 smallest primes above 2^30 with p = 1 mod 2^17)."""
    sys.path.insert(0, ROOT)
    import bench
    want, step, p = [], 1 << 17, (1 << 30) + 1
    while len(want) < 32:
        if all(p % q for q in range(3, int(p ** 0.5) + 1, 2)):
            want.append(p)
        p += step
    assert bench.rns_primes(32, 17, 1 << 30) == want
