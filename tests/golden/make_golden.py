#!/usr/bin/env python3
"""Generate the committed golden fixtures under tests/golden/.  Run in the build container only
(needs /root/reference and oracle/_ref/libref_baseline.so = the reference's main-baseline.cc
compiled in place by oracle/Makefile).

Outputs (data only -- inputs and expected outputs, no reference source text):
  nttdata_0.in / nttdata_0.out  the reference's tiny text fixture, verbatim (loader format tests)
  nttdata_{1,2,3}.npz           the reference's n=131072 fixtures packed as u32 (n, p, a, b, ab)
  ref_ntt_vectors.npz           forward / inverse NTT vectors produced by the compiled reference
                                NTT_iterative (main-baseline.cc:148) on seeded inputs
  ref_polymul_vectors.npz       poly-multiply vectors produced by the compiled reference multiply
                                body (main-baseline.cc:201-232), incl. non-power-of-two n and
                                edge values {0, 1, p-1}
  crt_vectors.npz               large-modulus (p = 1337006139375617) products for n in {256, 1000},
                                expected values from Python big-int schoolbook (independent of
                                both the oracle and the reference: "parity unpinned" by the
                                reference's own data, nttdata/4.* is not shipped)
"""
import hashlib
import os
import shutil
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import orc  # noqa: E402

REF = "/root/reference"
SHA = {
    "0.in": "810473fab82fd44dc6cf5107bdb46ef5f246bce1fe0cbc90687d312ad1a548e9",
    "0.out": "4fffe5a88275d27f42c271479f6f810c830f4c39721b5f8166da213b1e7cee2b",
    "1.in": "7170c05be88e0d7e38465503512be70dfe6d04859f7415b9c615f58a4f5741fb",
    "1.out": "f9036a050f9301b9aaad717a722daed1b3e5c14a3551f018d72e17836adc9d3a",
    "2.in": "4c5088c5d15fa2476fb1ad703b99e019f47f56ce6a71e76f3bc5a61fe6b02864",
    "2.out": "c6d99f43a1931d149b05db5f1750ada0c102b93d8fd902950d893731990ca6cd",
    "3.in": "41c920902f2e941290dbffc019794c78fbf9da8b364f692c0f7feb536a1fa389",
    "3.out": "fec409c97c716c219c46d8057e2214136a4d6a8bfa43d5b79638b7c0b8e5629b",
}
PRIMES = (7340033, 104857601, 469762049, 998244353)
P64 = 1337006139375617  # main-openmp.cc:636


def sha(path):
    return hashlib.sha256(open(path, "rb").read()).hexdigest()


def pack_nttdata():
    for name, want in SHA.items():
        got = sha(f"{REF}/nttdata/{name}")
        assert got == want, (name, got)
    shutil.copyfile(f"{REF}/nttdata/0.in", f"{HERE}/nttdata_0.in")
    shutil.copyfile(f"{REF}/nttdata/0.out", f"{HERE}/nttdata_0.out")
    for cid in (1, 2, 3):
        tin = np.array(open(f"{REF}/nttdata/{cid}.in").read().split(), dtype=np.uint64)
        n, p = int(tin[0]), int(tin[1])
        a, b = tin[2:2 + n].astype(np.uint32), tin[2 + n:2 + 2 * n].astype(np.uint32)
        ab = np.array(open(f"{REF}/nttdata/{cid}.out").read().split(), dtype=np.uint64).astype(np.uint32)
        assert a.size == n and b.size == n and ab.size == 2 * n - 1
        np.savez_compressed(f"{HERE}/nttdata_{cid}.npz", n=np.int64(n), p=np.int64(p), a=a, b=b, ab=ab)
        print("packed case", cid, n, p)


def ref_vectors():
    r = orc.ref()
    rng = np.random.default_rng(20250711)
    out = {}
    for p in PRIMES:
        for L in (2, 4, 8, 16, 64, 1024, 4096):
            x = rng.integers(0, p, size=L, dtype=np.int64).astype(np.int32)
            x[0] = p - 1
            if L > 2:
                x[1] = 0
                x[2] = 1
            f = x.copy()
            r.ref_ntt_iterative(f, L, p, 1)
            i = x.copy()
            r.ref_ntt_iterative(i, L, p, -1)
            out[f"in_{p}_{L}"] = x.astype(np.uint32)
            out[f"fwd_{p}_{L}"] = f.astype(np.uint32)
            out[f"inv_{p}_{L}"] = i.astype(np.uint32)
    np.savez_compressed(f"{HERE}/ref_ntt_vectors.npz", **out)
    print("ref ntt vectors:", len(out))

    out = {}
    for p in PRIMES:
        for n in (1, 2, 3, 4, 5, 255, 256, 1000, 4096):
            a = rng.integers(0, p, size=n, dtype=np.int64).astype(np.int32)
            b = rng.integers(0, p, size=n, dtype=np.int64).astype(np.int32)
            a[0] = p - 1
            b[-1] = p - 1
            if n > 3:
                a[1], a[2], b[0], b[1] = 0, 1, 1, 0
            ab = np.zeros(2 * n - 1, dtype=np.int32)
            r.ref_poly_multiply_ntt(a, b, ab, n, p)
            if n <= 1000:   # second opinion from the reference's own schoolbook poly_multiply
                ab2 = np.zeros(2 * n - 1, dtype=np.int32)
                r.ref_poly_multiply_naive(a, b, ab2, n, p)
                assert np.array_equal(ab, ab2), (p, n)
            out[f"a_{p}_{n}"] = a.astype(np.uint32)
            out[f"b_{p}_{n}"] = b.astype(np.uint32)
            out[f"ab_{p}_{n}"] = ab.astype(np.uint32)
    np.savez_compressed(f"{HERE}/ref_polymul_vectors.npz", **out)
    print("ref polymul vectors:", len(out))


def crt_vectors():
    import random
    rnd = random.Random(20250711)
    out = {}
    for n in (256, 1000):
        a = [rnd.randrange(P64) for _ in range(n)]
        b = [rnd.randrange(P64) for _ in range(n)]
        a[0], b[-1] = P64 - 1, P64 - 1
        ab = [0] * (2 * n - 1)
        for i, x in enumerate(a):
            for j, y in enumerate(b):
                ab[i + j] += x * y
        full = ab
        out[f"a_{n}"] = np.array(a, dtype=np.uint64)
        out[f"b_{n}"] = np.array(b, dtype=np.uint64)
        out[f"ab_{n}"] = np.array([v % P64 for v in full], dtype=np.uint64)
        for k, m in enumerate(orc.CRT_MODS):
            out[f"res{k}_{n}"] = np.array([v % m for v in full], dtype=np.uint32)
    np.savez_compressed(f"{HERE}/crt_vectors.npz", p64=np.uint64(P64), **out)
    print("crt vectors:", len(out))


if __name__ == "__main__":
    pack_nttdata()
    ref_vectors()
    crt_vectors()
