// comm_ranks.cpp -- `comm_ranks N`: the shape of the reference's MPI program (main-mpi.cc:645-797: one process per rank,
// every rank holds the whole case, the residue products are exchanged and merged) on ntt_comm in HOST mode, in plain C++:
// N processes forked before anything touches the GPU, each with its own single-device context and communicator; the
// all-to-all the library calls back for is a copy through a shared-memory segment with a process-shared barrier -- where
// an MPI program passes MPI_Alltoall.  Every rank checks its results against the same library on
// its own single context: the large-modulus product, one split transform forward and back, one split product.
// All ranks use device 0 unless NTT_COMM_RANKS_SPREAD=1 (rank % device count).  Exit status 0 = every rank matched.
#include <hip/hip_runtime.h>
#include <pthread.h>
#include <sys/mman.h>
#include <sys/wait.h>
#include <unistd.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "../../include/ntt_amd.h"

struct Shared {
    pthread_barrier_t bar;
    size_t block;                                          // room per (sender, receiver) pair
};
struct Link {
    Shared *sh;
    unsigned char *area;                                   // [world][world][block]
    int world, rank;
};

static int alltoall(void *user, const void *send, void *recv, size_t bytes)
{
    Link *l = static_cast<Link *>(user);
    if (bytes > l->sh->block) return 1;
    const unsigned char *s = static_cast<const unsigned char *>(send);
    unsigned char *r = static_cast<unsigned char *>(recv);
    for (int h = 0; h < l->world; ++h)
        std::memcpy(l->area + ((size_t)l->rank * l->world + h) * l->sh->block, s + (size_t)h * bytes, bytes);
    pthread_barrier_wait(&l->sh->bar);
    for (int o = 0; o < l->world; ++o)
        std::memcpy(r + (size_t)o * bytes, l->area + ((size_t)o * l->world + l->rank) * l->sh->block, bytes);
    pthread_barrier_wait(&l->sh->bar);                     // nobody overwrites the segment before everyone has read it
    return 0;
}

static uint64_t splitmix(uint64_t &x)
{
    uint64_t z = (x += 0x9E3779B97F4A7C15ULL);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
    return z ^ (z >> 31);
}

#define NTT_TRY(x) do { int rc_ = (x); if (rc_) { std::fprintf(stderr, "rank %d: %s -> %d: %s\n", rank, #x, rc_, ntt_last_error()); return 1; } } while (0)
#define HIP_OK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::fprintf(stderr, "rank %d: %s: %s\n", rank, #x, hipGetErrorString(e_)); return 1; } } while (0)

static int run_rank(int rank, int world, Link *link)
{
    int ndev = 1;
    HIP_OK(hipGetDeviceCount(&ndev));
    const char *spread = std::getenv("NTT_COMM_RANKS_SPREAD");
    int dev = (spread && spread[0] == '1') ? rank % ndev : 0;
    ntt_ctx *ctx = nullptr;
    NTT_TRY(ntt_ctx_create(&dev, 1, &ctx));
    ntt_comm *comm = nullptr;
    NTT_TRY(ntt_comm_create_host(ctx, alltoall, link, world, rank, &comm));
    int bad = 0;

    // (1) the reference's fifth case shape: n = 2^17, 51-bit modulus, the same operands on every rank
    {
        const int n = 131072;
        const uint64_t p64 = 1337006139375617ULL;
        std::vector<uint64_t> a(n), b(n), ab(2 * n - 1), want(2 * n - 1);
        uint64_t s = 0xC47A;
        for (int i = 0; i < n; ++i) { a[i] = splitmix(s) % p64; b[i] = splitmix(s) % p64; }
        NTT_TRY(ntt_comm_crt_multiply(comm, a.data(), b.data(), ab.data(), n, p64));
        NTT_TRY(ntt_crt_multiply(ctx, a.data(), b.data(), want.data(), n, p64));
        if (ab != want) { std::fprintf(stderr, "rank %d: large-modulus product differs\n", rank); ++bad; }
    }
    // (2) one transform of 2^20 points split over the ranks, and the product of two polynomials of 2^19 coefficients
    if (world >= 2 && (world & (world - 1)) == 0) {
        const int lgL = 20;
        const uint32_t p = 469762049u, L = 1u << lgL, S = L / (uint32_t)world;
        std::vector<uint32_t> x(L), y(L / 2), X(L), xy(L);
        uint64_t s = 26;
        for (auto &v : x) v = (uint32_t)(splitmix(s) % p);
        for (auto &v : y) v = (uint32_t)(splitmix(s) % p);
        ntt_plan *whole = nullptr;
        NTT_TRY(ntt_plan_create(ctx, p, (int)(L / 2), 1, NTT_MODMUL_MONTGOMERY, &whole));
        X = x;
        NTT_TRY(ntt_forward(whole, X.data(), 1));            // natural order in and out: X[k]
        NTT_TRY(ntt_poly_multiply_batched(whole, x.data(), y.data(), xy.data(), L - 1, 1));      // x[0 .. L/2) times y
        xy[L - 1] = 0;
        ntt_dist_plan *dp = nullptr;
        NTT_TRY(ntt_dist_plan_create(ctx, p, lgL, world, rank, NTT_MODMUL_MONTGOMERY, &dp));
        uint32_t *din, *dout, *dback, *db;
        HIP_OK(hipMalloc(&din, S * 4)); HIP_OK(hipMalloc(&dout, S * 4)); HIP_OK(hipMalloc(&dback, S * 4)); HIP_OK(hipMalloc(&db, S * 4));
        HIP_OK(hipMemcpy(din, x.data() + (size_t)rank * S, S * 4, hipMemcpyHostToDevice));
        NTT_TRY(ntt_comm_dist_forward(comm, dp, din, dout));
        NTT_TRY(ntt_comm_dist_inverse(comm, dp, dout, dback));
        NTT_TRY(ntt_ctx_synchronize(ctx));
        std::vector<uint32_t> got(S), back(S);
        HIP_OK(hipMemcpy(got.data(), dout, S * 4, hipMemcpyDeviceToHost));
        HIP_OK(hipMemcpy(back.data(), dback, S * 4, hipMemcpyDeviceToHost));
        for (uint32_t k2 = 0; k2 < S; ++k2)
            if (got[k2] != X[(size_t)rank + (size_t)world * k2]) { std::fprintf(stderr, "rank %d: split transform differs at %u\n", rank, k2); ++bad; break; }
        if (std::memcmp(back.data(), x.data() + (size_t)rank * S, S * 4) != 0) { std::fprintf(stderr, "rank %d: round trip differs\n", rank); ++bad; }
        const bool holds = rank < world / 2;                 // blocks of the operands live on the lower half of the ranks
        if (holds) HIP_OK(hipMemcpy(db, y.data() + (size_t)rank * S, S * 4, hipMemcpyHostToDevice));
        NTT_TRY(ntt_comm_dist_multiply(comm, dp, holds ? din : nullptr, holds ? db : nullptr, dout));
        NTT_TRY(ntt_ctx_synchronize(ctx));
        HIP_OK(hipMemcpy(got.data(), dout, S * 4, hipMemcpyDeviceToHost));
        if (std::memcmp(got.data(), xy.data() + (size_t)rank * S, S * 4) != 0) { std::fprintf(stderr, "rank %d: split product differs\n", rank); ++bad; }
        (void)hipFree(din); (void)hipFree(dout); (void)hipFree(dback); (void)hipFree(db);
        NTT_TRY(ntt_dist_plan_destroy(dp));
        NTT_TRY(ntt_plan_destroy(whole));
    }
    NTT_TRY(ntt_comm_destroy(comm));
    NTT_TRY(ntt_ctx_destroy(ctx));
    return bad ? 1 : 0;
}

int main(int argc, char **argv)
{
    const int world = argc > 1 ? std::atoi(argv[1]) : 2;
    if (world < 1 || world > 16) { std::fprintf(stderr, "usage: comm_ranks N   (1 <= N <= 16)\n"); return 2; }
    const size_t block = world == 1 ? (8u << 20) : (2u << 20);    // the largest block of the calls below: 4 MiB at one rank, 1 MiB at two
    const size_t bytes = sizeof(Shared) + (size_t)world * world * block;
    void *mem = mmap(nullptr, bytes, PROT_READ | PROT_WRITE, MAP_SHARED | MAP_ANONYMOUS, -1, 0);
    if (mem == MAP_FAILED) { std::perror("mmap"); return 2; }
    Shared *sh = static_cast<Shared *>(mem);
    sh->block = block;
    pthread_barrierattr_t at;
    pthread_barrierattr_init(&at);
    pthread_barrierattr_setpshared(&at, PTHREAD_PROCESS_SHARED);
    pthread_barrier_init(&sh->bar, &at, (unsigned)world);
    std::vector<pid_t> kids;
    for (int r = 0; r < world; ++r) {                      // fork first: no process that has initialised the GPU forks
        const pid_t pid = fork();
        if (pid < 0) { std::perror("fork"); return 2; }
        if (pid == 0) {
            Link link{sh, reinterpret_cast<unsigned char *>(sh + 1), world, r};
            const int rc = run_rank(r, world, &link);
            std::fflush(nullptr);
            _exit(rc);
        }
        kids.push_back(pid);
    }
    int failed = 0;
    for (pid_t pid : kids) {
        int st = 0;
        waitpid(pid, &st, 0);
        if (!WIFEXITED(st) || WEXITSTATUS(st) != 0) ++failed;
    }
    std::printf("comm_ranks: %d ranks, %s\n", world, failed ? "MISMATCH" : "ok");
    return failed ? 1 : 0;
}
