// ntt_cli.cpp -- the reference harness loop on top of libntt_amd.so's C ABI.
//
// Mirrors main() of main-montgomery.cc:298-362 / main-openmp.cc:632-727: for each case id in
// [test_begin, test_end] read "<data>/<id>.in", multiply (50 epochs, averaged), check against
// "<data>/<id>.out", print the verdict and the "average latency for n = .. p = .. : .. (us)" line
// (value in milliseconds, as the reference prints it), write "<out>/<id>.out".
// Differences on purpose: the data/out directories and the case range come from the command
// line (the reference hard-codes "/nttdata/" and 0..4), a missing case is reported and skipped
// instead of producing a spurious verdict, and the exit status is non-zero on any mismatch.
//
//   ntt_cli [--data DIR] [--out DIR] [--begin N] [--end N] [--epochs N] [--barrett]
//           [--pack]      convert "<id>.in" / "<id>.out" to the packed "<id>.inb" / "<id>.outb" and exit
//           [--bin]       read, check and write the packed files instead of the text ones
//           [--direct64]  moduli above 2^31: one 64-bit transform (ntt_plan64) instead of the 4-prime CRT
//           [--devices 0,1,2,3]  the rank layout of main-mpi.cc:642-819 inside one process: a multi-device context
//                         with one slot per listed GPU; cases with a 32-bit modulus run on slot 0 while the others idle
//                         (main-mpi.cc:710-738), a larger modulus puts CRT prime k on slot k / ceil(4 / slots) and
//                         exchanges coefficient slices before the merge (:739-797; multidev.cpp)
#include <hip/hip_runtime.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/ntt_amd.h"

int main(int argc, char **argv)
{
    std::string data = "./nttdata", out = "./files";
    int begin = 0, end = 4, epochs = 50, kind = NTT_MODMUL_MONTGOMERY;
    bool pack = false, bin = false, direct64 = false;
    std::vector<int> devices{0};
    for (int i = 1; i < argc; ++i) {
        auto need = [&](const char *flag) -> const char * {
            if (i + 1 >= argc) { std::fprintf(stderr, "%s needs a value\n", flag); std::exit(2); }
            return argv[++i];
        };
        if (!std::strcmp(argv[i], "--data")) data = need("--data");
        else if (!std::strcmp(argv[i], "--out")) out = need("--out");
        else if (!std::strcmp(argv[i], "--begin")) begin = std::atoi(need("--begin"));
        else if (!std::strcmp(argv[i], "--end")) end = std::atoi(need("--end"));
        else if (!std::strcmp(argv[i], "--epochs")) epochs = std::atoi(need("--epochs"));
        else if (!std::strcmp(argv[i], "--barrett")) kind = NTT_MODMUL_BARRETT;
        else if (!std::strcmp(argv[i], "--pack")) pack = true;
        else if (!std::strcmp(argv[i], "--bin")) bin = true;
        else if (!std::strcmp(argv[i], "--direct64")) direct64 = true;
        else if (!std::strcmp(argv[i], "--devices")) {
            devices.clear();
            for (const char *c = need("--devices"); *c; ) {
                char *e = nullptr;
                devices.push_back((int)std::strtol(c, &e, 10));
                if (e == c) { std::fprintf(stderr, "--devices takes a comma-separated list of device ids\n"); return 2; }
                c = *e == ',' ? e + 1 : e;
            }
            if (devices.empty()) { std::fprintf(stderr, "--devices: empty list\n"); return 2; }
        }
        else { std::fprintf(stderr, "unknown argument %s\n", argv[i]); return 2; }
    }
    if (epochs < 1) epochs = 1;
    if (pack) {                                          // host-only: no GPU needed
        int failed = 0;
        for (int id = begin; id <= end; ++id) {
            if (ntt_pack_case(data.c_str(), id) != NTT_OK) {
                std::fprintf(stderr, "case %d not packed: %s\n", id, ntt_last_error());
                ++failed;
            }
        }
        return failed ? 1 : 0;
    }
    ntt_ctx *top = nullptr, *ctx = nullptr;               // top: every slot (large-modulus products); ctx: slot 0 (the rest)
    if (ntt_ctx_create(devices.data(), (int)devices.size(), &top) != NTT_OK ||
        ntt_ctx_device(top, 0, &ctx, nullptr) != NTT_OK || hipSetDevice(devices[0]) != hipSuccess) {
        std::fprintf(stderr, "ntt_ctx_create: %s\n", ntt_last_error());
        return 1;
    }
    const size_t cap = 1u << 22;
    std::vector<uint64_t> a(cap), b(cap), ab;
    int bad = 0;
    for (int id = begin; id <= end; ++id) {
        int n = 0;
        uint64_t p = 0;
        int rc, width = 8;
        if (bin) {
            rc = ntt_read_case_bin(data.c_str(), id, a.data(), b.data(), cap * 8, &n, &p, &width);
            if (rc == NTT_OK && width == 4) {            // widen in place, back to front
                const uint32_t *a4 = reinterpret_cast<const uint32_t *>(a.data());
                const uint32_t *b4 = reinterpret_cast<const uint32_t *>(b.data());
                for (int i = n - 1; i >= 0; --i) { a[i] = a4[i]; b[i] = b4[i]; }
            }
        } else {
            rc = ntt_read_case(data.c_str(), id, a.data(), b.data(), cap, &n, &p);
        }
        if (rc != NTT_OK) {
            std::fprintf(stderr, "case %d skipped: %s\n", id, ntt_last_error());
            continue;
        }
        ab.assign((size_t)2 * n - 1, 0);
        double total_ms = 0;
        if (p < (1ULL << 31)) {
            ntt_plan *plan = nullptr;
            if (ntt_plan_create(ctx, (uint32_t)p, n, 1, kind, &plan) != NTT_OK) {
                std::fprintf(stderr, "case %d: %s\n", id, ntt_last_error());
                ++bad;
                continue;
            }
            std::vector<uint32_t> a32(n), b32(n), ab32((size_t)2 * n - 1);
            for (int i = 0; i < n; ++i) { a32[i] = (uint32_t)a[i]; b32[i] = (uint32_t)b[i]; }
            uint32_t *da = nullptr, *db = nullptr, *dab = nullptr;
            if (hipMalloc(&da, (size_t)n * 4) != hipSuccess || hipMalloc(&db, (size_t)n * 4) != hipSuccess ||
                hipMalloc(&dab, ab32.size() * 4) != hipSuccess ||
                hipMemcpy(da, a32.data(), (size_t)n * 4, hipMemcpyHostToDevice) != hipSuccess ||
                hipMemcpy(db, b32.data(), (size_t)n * 4, hipMemcpyHostToDevice) != hipSuccess) {
                std::fprintf(stderr, "case %d: device allocation or copy failed\n", id);
                (void)hipFree(da); (void)hipFree(db); (void)hipFree(dab);
                ntt_plan_destroy(plan);
                ++bad;
                continue;
            }
            for (int e = 0; e < epochs + 1; ++e) {      // epoch 0 warms up (module load), untimed
                auto t0 = std::chrono::high_resolution_clock::now();
                rc = ntt_poly_multiply_batched(plan, da, db, dab, ab32.size(), 1);
                if (rc == NTT_OK) rc = ntt_ctx_synchronize(ctx);
                auto t1 = std::chrono::high_resolution_clock::now();
                if (rc != NTT_OK) break;
                if (e > 0) total_ms += std::chrono::duration<double, std::milli>(t1 - t0).count();
            }
            if (rc == NTT_OK && hipMemcpy(ab32.data(), dab, ab32.size() * 4, hipMemcpyDeviceToHost) != hipSuccess) {
                std::fprintf(stderr, "case %d: copy back failed\n", id);
                rc = NTT_ERR_HIP;
            }
            (void)hipFree(da); (void)hipFree(db); (void)hipFree(dab);
            ntt_plan_destroy(plan);
            for (size_t i = 0; i < ab32.size(); ++i) ab[i] = ab32[i];
        } else if (direct64) {
            ntt_plan64 *plan = nullptr;
            if (ntt_plan64_create(ctx, p, n, 1, &plan) != NTT_OK) {
                std::fprintf(stderr, "case %d: %s\n", id, ntt_last_error());
                ++bad;
                continue;
            }
            for (int e = 0; e < epochs + 1; ++e) {
                auto t0 = std::chrono::high_resolution_clock::now();
                rc = ntt_poly_multiply64_batched(plan, a.data(), b.data(), ab.data(), ab.size(), 1);
                auto t1 = std::chrono::high_resolution_clock::now();
                if (rc != NTT_OK) break;
                if (e > 0) total_ms += std::chrono::duration<double, std::milli>(t1 - t0).count();
            }
            ntt_plan64_destroy(plan);
        } else {
            for (int e = 0; e < epochs + 1; ++e) {
                auto t0 = std::chrono::high_resolution_clock::now();
                rc = ntt_crt_multiply(top, a.data(), b.data(), ab.data(), n, p);
                auto t1 = std::chrono::high_resolution_clock::now();
                if (rc != NTT_OK) break;
                if (e > 0) total_ms += std::chrono::duration<double, std::milli>(t1 - t0).count();
            }
        }
        if (rc != NTT_OK) {
            std::fprintf(stderr, "case %d: %s\n", id, ntt_last_error());
            ++bad;
            continue;
        }
        std::vector<uint32_t> ab4;
        if (bin && width == 4) {
            ab4.resize(ab.size());
            for (size_t i = 0; i < ab.size(); ++i) ab4[i] = (uint32_t)ab[i];
        }
        const void *abw = (bin && width == 4) ? (const void *)ab4.data() : (const void *)ab.data();
        rc = bin ? ntt_check_case_bin(data.c_str(), id, abw, n, width) : ntt_check_case(data.c_str(), id, ab.data(), n);
        if (rc == NTT_ERR_MISMATCH) ++bad;
        else if (rc != NTT_OK) std::fprintf(stderr, "case %d not checked: %s\n", id, ntt_last_error());
        std::printf("average latency for n = %d p = %llu : %g (us) \n", n, (unsigned long long)p,
                    total_ms / epochs);
        if ((bin ? ntt_write_case_bin(out.c_str(), id, abw, n, p, width) : ntt_write_case(out.c_str(), id, ab.data(), n)) != NTT_OK)
            std::fprintf(stderr, "case %d not written: %s\n", id, ntt_last_error());
    }
    ntt_ctx_destroy(top);
    return bad ? 1 : 0;
}
