// harness.cpp -- the reference harness's file helpers with its text formats, behind the C ABI.
//
//   ntt_read_case  <- fRead  (main-montgomery.cc:15-32; u64 overload main-openmp.cc:21-40)
//   ntt_check_case <- fCheck (main-montgomery.cc:34-54)
//   ntt_write_case <- fWrite (main-montgomery.cc:56-69)
//
// Formats: "<id>.in" = "n p\n" + n decimals + n decimals, whitespace
// separated; "<id>.out" = 2n-1 decimals; written files hold one decimal per line.  The reference
// parses with iostream `>>`; 2 MB of decimals per case make that the slowest step once the
// multiply runs on the GPU, so the loader here reads the file whole and scans digits by hand.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/ntt_amd.h"
#include "engine.hpp"

namespace {

bool slurp(const std::string &path, std::vector<char> &buf)
{
    FILE *f = std::fopen(path.c_str(), "rb");
    if (!f) return false;
    std::fseek(f, 0, SEEK_END);
    long sz = std::ftell(f);
    std::fseek(f, 0, SEEK_SET);
    buf.resize(sz > 0 ? (size_t)sz : 0);
    size_t got = buf.empty() ? 0 : std::fread(buf.data(), 1, buf.size(), f);
    std::fclose(f);
    return got == buf.size();
}

// Next unsigned decimal at or after pos; false at end of data or on a non-numeric token.
bool next_u64(const std::vector<char> &b, size_t &pos, uint64_t &v)
{
    while (pos < b.size() && (b[pos] == ' ' || b[pos] == '\n' || b[pos] == '\r' || b[pos] == '\t')) ++pos;
    if (pos >= b.size() || b[pos] < '0' || b[pos] > '9') return false;
    uint64_t x = 0;
    while (pos < b.size() && b[pos] >= '0' && b[pos] <= '9') x = x * 10 + (uint64_t)(b[pos++] - '0');
    v = x;
    return true;
}

std::string case_path(const char *dir, int id, const char *ext)
{
    std::string d = dir ? dir : ".";
    if (!d.empty() && d.back() != '/') d += '/';
    return d + std::to_string(id) + ext;
}

}  // namespace

extern "C" int ntt_read_case(const char *dir, int id, uint64_t *a, uint64_t *b, size_t cap, int *n, uint64_t *p)
{
    if (!a || !b || !n || !p) return ntt::fail(NTT_ERR_INVALID, "ntt_read_case: null argument");
    const std::string path = case_path(dir, id, ".in");
    std::vector<char> buf;
    if (!slurp(path, buf)) return ntt::fail(NTT_ERR_IO, "ntt_read_case: cannot read " + path);
    size_t pos = 0;
    uint64_t nn = 0, pp = 0;
    if (!next_u64(buf, pos, nn) || !next_u64(buf, pos, pp) || nn == 0 || nn > 0x7fffffffULL)
        return ntt::fail(NTT_ERR_IO, "ntt_read_case: bad header in " + path);
    if (nn > cap) return ntt::fail(NTT_ERR_INVALID, "ntt_read_case: n exceeds the caller's capacity");
    for (uint64_t i = 0; i < nn; ++i)
        if (!next_u64(buf, pos, a[i])) return ntt::fail(NTT_ERR_IO, "ntt_read_case: truncated a[] in " + path);
    for (uint64_t i = 0; i < nn; ++i)
        if (!next_u64(buf, pos, b[i])) return ntt::fail(NTT_ERR_IO, "ntt_read_case: truncated b[] in " + path);
    *n = (int)nn;
    *p = pp;
    return NTT_OK;
}

extern "C" int ntt_check_case(const char *dir, int id, const uint64_t *ab, int n)
{
    if (!ab || n <= 0) return ntt::fail(NTT_ERR_INVALID, "ntt_check_case: bad argument");
    const std::string path = case_path(dir, id, ".out");
    std::vector<char> buf;
    if (!slurp(path, buf)) return ntt::fail(NTT_ERR_IO, "ntt_check_case: cannot read " + path);
    size_t pos = 0;
    for (int i = 0; i < 2 * n - 1; ++i) {
        uint64_t x = 0;
        if (!next_u64(buf, pos, x)) return ntt::fail(NTT_ERR_IO, "ntt_check_case: truncated " + path);
        if (x != ab[i]) {
            std::puts("多项式乘法结果错误");          // the reference's verdict line, main-montgomery.cc:48
            std::fflush(stdout);
            return ntt::fail(NTT_ERR_MISMATCH, "ntt_check_case: mismatch at coefficient " + std::to_string(i));
        }
    }
    std::puts("多项式乘法结果正确");                  // main-montgomery.cc:52
    std::fflush(stdout);
    return NTT_OK;
}

extern "C" int ntt_write_case(const char *dir, int id, const uint64_t *ab, int n)
{
    if (!ab || n <= 0) return ntt::fail(NTT_ERR_INVALID, "ntt_write_case: bad argument");
    const std::string path = case_path(dir, id, ".out");
    FILE *f = std::fopen(path.c_str(), "wb");
    if (!f) return ntt::fail(NTT_ERR_IO, "ntt_write_case: cannot open " + path);
    std::string out;
    out.reserve((size_t)(2 * n - 1) * 12);
    char tmp[24];
    for (int i = 0; i < 2 * n - 1; ++i) {
        int len = std::snprintf(tmp, sizeof(tmp), "%llu\n", (unsigned long long)ab[i]);
        out.append(tmp, (size_t)len);
    }
    const bool ok = std::fwrite(out.data(), 1, out.size(), f) == out.size();
    std::fclose(f);
    return ok ? NTT_OK : ntt::fail(NTT_ERR_IO, "ntt_write_case: short write to " + path);
}
