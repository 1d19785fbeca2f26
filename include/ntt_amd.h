/*
 * ntt_amd.h -- C ABI of the MI355X-native NTT polynomial-multiplication engine (libntt_amd.so).
 *
 * This is the drop-in boundary for the reference's one hot path: the
 * teacher-harness contract poly_multiply(a, b, ab, n, p) with fRead / fCheck / fWrite around it.
 * The reference has no FFI layer; each declaration cites the reference item it replaces
 * (file:line relative to the reference tree).  Plain pointers and sizes only; no exceptions
 * cross this boundary; every function except poly_multiply returns an int status.
 *
 * Pointer arguments documented "host or device" are classified with hipPointerGetAttributes:
 * device pointers are used in place (no copies, asynchronous on the context's stream), host
 * pointers are staged through device memory and the call returns after the D2H copy.
 *
 * There is NO CPU fallback: without a usable gfx950 device every compute entry point fails
 * (NTT_ERR_HIP), and poly_multiply aborts with a message.
 */
#ifndef NTT_AMD_H
#define NTT_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum {
    NTT_OK = 0,
    NTT_ERR_INVALID = 1,      /* bad argument (null pointer, n <= 0, batch <= 0, ...) */
    NTT_ERR_UNSUPPORTED = 2,  /* p not an NTT-friendly prime for this length, p >= 2^31, L too large */
    NTT_ERR_HIP = 3,          /* HIP runtime failure; ntt_last_error() holds hipGetErrorString */
    NTT_ERR_NOMEM = 4,
    NTT_ERR_IO = 5,           /* fixture file missing / malformed */
    NTT_ERR_MISMATCH = 6,     /* ntt_check_case: result differs from the .out file */
    NTT_ERR_INTERNAL = 7      /* a C++ exception other than bad_alloc stopped at the boundary (none crosses it) */
} ntt_status;

/* Modular-multiplication flavour of a plan.  Both give canonical residues in [0,p) and hence
 * bit-identical results.  MONTGOMERY replaces class montgomery (main-montgomery.cc:133-234,
 * REDC :170-181) with 32-bit lanes, R = 2^32; BARRETT replaces class Barrett
 * (main-barrett.cc:130-172, reduce :145-150) with 32-bit lanes and per-twiddle precomputed
 * quotients. */
typedef enum { NTT_MODMUL_MONTGOMERY = 0, NTT_MODMUL_BARRETT = 1 } ntt_modmul_kind;

typedef struct ntt_ctx ntt_ctx;     /* one GPU + one stream (or one slot per GPU, ntt_ctx_create); thread-safe per handle */
typedef struct ntt_plan ntt_plan;   /* (p, n, L, batch capacity, twiddle tables, work buffers) */

/* ---- 1. the reference's own entry point --------------------------------------------------- */

/* Replaces: void poly_multiply(int *a, int *b, int *ab, int n, int p)
 *           main-montgomery.cc:71-77 (identical in every .cc), called at :337-338.
 * Same signature and ownership: a, b hold n values in [0,p) (host memory), ab receives the
 * 2n-1 product coefficients in [0,p); entries >= 2n-1 are not touched.  Overwrite semantics
 * (the reference accumulates into a pre-zeroed ab; with ab zeroed by the caller, as the harness
 * does at :314, the results are identical).  p must be an NTT-friendly prime < 2^31.
 * One-shot wrapper over a lazily created process-global context on device 0 with cached plans;
 * it cannot report errors through its signature, so it prints a message and aborts on failure. */
void poly_multiply(int *a, int *b, int *ab, int n, int p);

/* ---- 2. handle API ------------------------------------------------------------------------ */

/* Create a context on the given HIP device(s).
 * ndev == 1: one GPU + one stream; every entry point below takes such a context (the reference's CUDA files use the
 * implicit device, e.g. main-montgomery-GPU.cu:383).
 * ndev > 1: a MULTI-DEVICE context, the in-library form of the reference's distributed program main-mpi.cc:642-819
 * for one process that drives the GPUs of a node over HIP peer-to-peer (xGMI): one "slot" per entry of device_ids
 * (an id may repeat: {0, 0, 0, 0} rehearses four slots on a one-GPU box); peer access is enabled between every pair
 * of distinct devices.  On such a context
 *   - ntt_plan_create / ntt_negacyclic_plan_create make one plan per slot; ntt_poly_multiply_batched,
 *     ntt_negacyclic_multiply_batched, ntt_forward and ntt_inverse shard their batch rows over the slots:
 *     slot s takes rows [s * ceil(batch / ndev), ...), the reference's rule for primes (main-mpi.cc:657-660)
 *     applied to independent products -- no exchange.  Arrays may be host arrays (each slot stages its rows) or
 *     device arrays that every slot can reach (its own device, or any device of the context with peer access);
 *   - ntt_crt_multiply puts prime k on slot k / ceil(4 / ndev) (main-mpi.cc:657-660, :770), then ONE exchange:
 *     every slot receives coefficient slice h of all four residue polynomials (peer copies), merges its slice on
 *     its own GPU and writes it to ab -- instead of the reference's gather to rank 0 and merge there
 *     (main-mpi.cc:772-796); ntt_crt_multiply_batched with batch > 1 shards the products by rows instead;
 *   - ntt_dist_plan_create (world = ndev) + ntt_dist_forward / ntt_dist_inverse run ONE transform split over the
 *     slots with both exchanges inside the library (below);
 *   - calls return when the result is complete, except ntt_dist_forward / ntt_dist_inverse and calls whose arrays
 *     are all device arrays (queued; ntt_ctx_synchronize waits for every slot);
 *   - everything else (streams, kernel timing, 64-bit plans, the CRT halves, streamed and no-alloc products) is per
 *     slot: ntt_ctx_device(ctx, slot, &slot_ctx, &device) hands out the slot's single-device context, owned by ctx. */
int ntt_ctx_create(const int *device_ids, int ndev, ntt_ctx **out_ctx);
int ntt_ctx_device_count(const ntt_ctx *ctx);
/* Products under SEVERAL moduli in one call -- RNS limbs (BASELINE.json configs[3]; the reference's four CRT primes are
 * the same shape, main-openmp.cc:697: one thread per prime): limb i multiplies the `batch` rows of a[i], b[i] into ab[i]
 * (row stride ab_stride) with plans[i] -- plans of ntt_plan_create (rows of 2n - 1 coefficients) or of
 * ntt_negacyclic_plan_create (rows of n: the limbs of an x^n + 1 ring).  Distinct plans of ONE single-device context, device arrays only.  Each limb's
 * chain of kernels runs whole on stream lane i % lanes through its plan's own work buffers, with one fork in front of
 * the first limb and one join behind the last; a loop of ntt_poly_multiply_batched calls forks and joins the lanes per
 * call instead, which drains the chip 32 times for 32 limbs (943 k vs 1 037 k products/s for 32 x 256 products at
 * n = 2^16, profiles/rns_probe_r03.txt).  Queued; outputs of different limbs must not overlap (they are written
 * concurrently). */
int ntt_poly_multiply_limbs(ntt_plan *const *plans, int nlimbs, const uint32_t *const *a, const uint32_t *const *b,
                            uint32_t *const *ab, size_t ab_stride, int batch);
/* Products whose operands already live where they are multiplied: a plan of a multi-device context, one array triple per
 * slot -- a[s], b[s] are [count[s]][n], ab[s] is [count[s]][ab_stride], on (or reachable from) the device of slot s;
 * count[s] <= ceil(batch / slots), 0 for a slot without work.  The batch-per-GPU layout of BASELINE.json configs[1] /
 * north_star item (4) for ONE process (ntt_poly_multiply_batched on such a plan takes one array and slices it by
 * ntt_partition instead); no exchange, one host thread per slot.  Queued for device arrays. */
int ntt_poly_multiply_batched_slots(ntt_plan *plan, const uint32_t *const *a, const uint32_t *const *b, uint32_t *const *ab,
                                    size_t ab_stride, const int *count);
/* The sharding rule every multi-device call uses, as a function: units [*start, *start + *count) of `total_units` belong
 * to slot (or rank) `rank` of `world` -- ceil(total / world) units each, the last ones fewer or none.  It is the
 * reference's rule for its four primes, mods_per_process = ceil(4 / size), start = rank * mods_per_process,
 * end = min(start + mods_per_process, 4) (main-mpi.cc:657-660), for any number of units.  Pure host arithmetic. */
int ntt_partition(long total_units, int world, int rank, long *start, long *count);
int ntt_ctx_device(ntt_ctx *ctx, int slot, ntt_ctx **out_slot_ctx, int *out_device);
/* Destroy every plan made from a context BEFORE the context: a plan handle points into its context. */
int ntt_ctx_destroy(ntt_ctx *ctx);
/* Use an existing hipStream_t (passed as void*) for all work of this context; NULL restores the
 * context's own stream (to run on the null stream itself pass the handle (void*)1, hipStreamLegacy; the library then uses stream 0).
 * ntt_ctx_stream returns the stream in use (as void*). */
int ntt_ctx_set_stream(ntt_ctx *ctx, void *hip_stream);
void *ntt_ctx_stream(ntt_ctx *ctx);
int ntt_ctx_synchronize(ntt_ctx *ctx);

/* Build a plan for products of two length-n polynomials modulo the prime p (< 2^31):
 * chooses L = the smallest power of two >= 2n (main-barrett.cc:249-253; the power-of-four rule
 * of main-montgomery.cc:317-321 gives the same first 2n-1 coefficients), finds a primitive root
 * (the reference hard-codes g = 3, main-montgomery.cc:257; any generator gives the same
 * canonical result), builds the twiddle tables in HBM and sizes the work buffers for `batch`
 * simultaneous products.  Replaces the per-call setup of main-montgomery.cc:315-321 and the
 * per-stage power() calls of NTT_iterative :262-268. */
int ntt_plan_create(ntt_ctx *ctx, uint32_t p, int n, int batch, int modmul_kind, ntt_plan **out_plan);
int ntt_plan_destroy(ntt_plan *plan);
/* Transform length L of the plan. */
int ntt_plan_length(const ntt_plan *plan);

/* batch products: a, b are [batch][n] u32 (row stride n), ab is [batch][2n-1] stored with row
 * stride `ab_stride` elements (>= 2n-1; SURVEY 8d uses 2n); batch <= the plan's batch capacity.
 * Host or device pointers.
 * Replaces the harness multiply body main-montgomery.cc:326-351 / main-barrett.cc:258-282
 * (pad, NTT, NTT, pointwise, inverse NTT, copy 2n-1). */
int ntt_poly_multiply_batched(ntt_plan *plan, const uint32_t *a, const uint32_t *b, uint32_t *ab,
                              size_t ab_stride, int batch);

/* The same products for callers that cannot allocate (code under HIP graph capture, tight serving loops,
 * harnesses that forbid hipMalloc): NO counterpart in the reference.  Same arguments and results as
 * ntt_poly_multiply_batched (bit for bit), with these differences:
 *   - a, b, ab must be DEVICE pointers (a host array would need a staging allocation: NTT_ERR_INVALID);
 *   - the intermediates of the transform passes live in a static device array of 64 MiB inside the library
 *     (one copy per GPU, held until the process ends) instead of the plan's work buffers; a product needs
 *     2 * L * 4 bytes of it (L = ntt_plan_length), so plans up to L = 2^23 (n <= 2^22) are served and longer
 *     ones are rejected with NTT_ERR_UNSUPPORTED; batch <= 65535 (any batch the plan was or was not sized for:
 *     the products go through the scratch array in groups of floor(64 MiB / (8 L)));
 *   - one stream (the context's), no kernel-timing events; the call makes no hipMalloc / hipFree / event or
 *     stream creation.  Asynchronous: results are complete after ntt_ctx_synchronize;
 *   - every call on a GPU shares that GPU's scratch array.  The library orders the calls of one PROCESS on a GPU
 *     itself: a per-device event marks the end of the last call there, each call makes its stream wait on it first and
 *     records it behind its last kernel (no allocation: the event exists since ntt_ctx_create), so two contexts, or
 *     one context whose stream changed between calls, serialise on the array in call order instead of racing.
 *     What the library cannot see still races: a second process on the same GPU has its own copy (no conflict), but
 *     work under HIP graph replay that was captured from this call runs without the wait and must not overlap
 *     another call on that GPU.
 * Not for negacyclic plans. */
int ntt_poly_multiply_batched_noalloc(ntt_plan *plan, const uint32_t *a, const uint32_t *b, uint32_t *ab,
                                      size_t ab_stride, int batch);

/* Negacyclic products: c = a * b mod (x^n + 1, p), n a power of two with
 * 2n | p - 1: a, b are [batch][n] u32, c is [batch][n] canonical residues with row stride c_stride
 * (>= n).  One transform of length n per operand (no zero padding); the twist by psi^i
 * (psi = g^((p-1)/2n)) is fused into the first forward pass and psi^-i into the last inverse pass.
 * The reference has no negacyclic routine: it is the ring FHE-style RNS workloads (BASELINE.json
 * configs[3]'s namesake) multiply in, built from the same NTT_iterative replacement
 * (main-montgomery.cc:255-295); its definition here is c[i] = ab[i] - ab[i+n] of the linear product.
 * ntt_forward / ntt_inverse on such a plan are the plain cyclic transforms of length n. */
int ntt_negacyclic_plan_create(ntt_ctx *ctx, uint32_t p, int n, int batch, int modmul_kind, ntt_plan **out);
int ntt_negacyclic_multiply_batched(ntt_plan *plan, const uint32_t *a, const uint32_t *b, uint32_t *c,
                                    size_t c_stride, int batch);

/* Products whose operands live in HOST memory, streamed: a, b are [count][n] u32, ab is
 * [count][ab_stride]; `count` is unbounded (slices of the plan's batch capacity travel host -> device
 * -> multiply -> host on three streams, transfers of one slice overlapping the kernels of another).
 * Pinned host memory (ntt_host_alloc) is read and written by DMA directly; pageable memory is staged
 * through pinned bounce buffers.  Returns when ab is complete.  Entries >= 2n-1 of a row stay untouched.
 * Same calling convention as the harness multiply body (host arrays in, host array out,
 * main-montgomery.cc:297, :326-351) for any number of products; SURVEY 8f rank 2. */
int ntt_poly_multiply_stream(ntt_plan *plan, const uint32_t *a, const uint32_t *b, uint32_t *ab,
                             size_t ab_stride, size_t count);
/* Page-locked host memory for the call above (hipHostMalloc / hipHostFree).  Arrays from here are also used IN PLACE by
 * a small ntt_poly_multiply_batched / poly_multiply call (at most 1 MiB per operand: one n = 2^17 product): 67 us per
 * call; pageable arrays of such a call go through page-locked buffers of the context and a CPU memcpy each way: 108 us
 * (128 through hipMemcpy; profiles/latency_r03c.txt).  Locking the caller's existing arrays in place (hipHostRegister)
 * is not offered: it fails with `invalid argument` on the machines this was developed on. */
int ntt_host_alloc(size_t bytes, void **ptr);
int ntt_host_free(void *ptr);

/* Batched in-place transforms of length L = ntt_plan_length(plan): data is [batch][L] u32,
 * natural order in, natural order out, canonical residues, forward root g^((p-1)/L) with the
 * plan's generator g (g = 3 for the reference's primes), inverse scaled by L^-1.
 * Replaces NTT_iterative(a, len, p, +1 / -1, m): main-montgomery.cc:255-295,
 * main-barrett.cc:193-228 (including bit_reverse :238-253). */
int ntt_forward(ntt_plan *plan, uint32_t *data, int batch);
int ntt_inverse(ntt_plan *plan, uint32_t *data, int batch);

/* Large-modulus product through the four fixed 30/31-bit primes
 * {1004535809, 1224736769, 469762049, 998244353} and a GPU CRT merge: a, b hold n values
 * < p64 (p64 < 2^62 and n * p64^2 < the product of the four primes), ab receives 2n-1 values
 * in [0,p64).  Host or device pointers.
 * Replaces the p_ >= 2^50 branch of main-openmp.cc:693-711: CRT_parallel :424-440,
 * crt_precompute :449-466, CRT_combine_parallel :468-485 (twins main.cc:813-951). */
int ntt_crt_multiply(ntt_ctx *ctx, const uint64_t *a, const uint64_t *b, uint64_t *ab, int n,
                     uint64_t p64);
/* The same for `batch` products at once: a, b are [batch][n], ab is [batch][2n-1] (contiguous rows).
 * One reduce launch, four batched residue products, one merge launch (the epoch loop of
 * main-openmp.cc:645-725 runs the branch above once per call; a batch amortises its launches). */
int ntt_crt_multiply_batched(ntt_ctx *ctx, const uint64_t *a, const uint64_t *b, uint64_t *ab, int n,
                             uint64_t p64, int batch);

/* Direct 64-bit path: products modulo ONE NTT-friendly prime p64 < 2^62 ((p64 - 1) divisible by
 * L = 2^ceil(log2 2n)) with 64-bit Montgomery lanes (R = 2^64): a, b are [batch][n] u64 with values
 * < p64, ab is [batch][2n-1] canonical residues with row stride ab_stride; host or device pointers.
 * Replaces the single-modulus branch of main-openmp.cc:678-692 (`montgomery m(R = 2^60, p_)`,
 * :650-651: toMontgomery, NTT_iterative x2, pointwise ModMul, inverse, fromMontgomery), which the
 * reference only takes for p_ < 2^50; with 64-bit lanes it also serves the 51-bit modulus of the
 * reference's fifth case, i.e. the caller of ntt_crt_multiply, for a prime of this shape.
 * A prime below 2^51 runs as exact integers in FP64 lanes (same results bit for bit, about 1.5x faster;
 * environment NTT_AMD_PLAN64_INT=1 at plan creation keeps the Montgomery lanes).  Operand values must be
 * < p64 (the reference's data are); values >= p64 are outside the contract of this entry point. */
typedef struct ntt_plan64 ntt_plan64;
int ntt_plan64_create(ntt_ctx *ctx, uint64_t p64, int n, int batch, ntt_plan64 **out);
int ntt_plan64_destroy(ntt_plan64 *plan);
int ntt_plan64_length(const ntt_plan64 *plan, uint32_t *L);
int ntt_poly_multiply64_batched(ntt_plan64 *plan, const uint64_t *a, const uint64_t *b, uint64_t *ab,
                                size_t ab_stride, int batch);
/* Negacyclic products with 64-bit lanes: c = a * b mod (x^n + 1, p64), n a power of two, 2n | p64 - 1;
 * c is [batch][n] with row stride c_stride.  The 64-bit twin of ntt_negacyclic_* below (FHE-style RNS
 * limbs are usually 50-60-bit primes). */
int ntt_negacyclic_plan64_create(ntt_ctx *ctx, uint64_t p64, int n, int batch, ntt_plan64 **out);
int ntt_negacyclic_multiply64_batched(ntt_plan64 *plan, const uint64_t *a, const uint64_t *b, uint64_t *c,
                                      size_t c_stride, int batch);

/* The two halves of ntt_crt_multiply as separate calls, for the prime-per-GPU sharding of the
 * reference's MPI driver (main-mpi.cc:657-660 gives each rank ceil(4/size) primes, :770 runs
 * process_mods_in_process on them, :772-796 gathers the residue polynomials and merges):
 *   ntt_crt_prime(k, &m)            the k-th fixed prime (k = 0..3), main-openmp.cc:654
 *   ntt_crt_residue_multiply        reduce a, b (< 2^64) mod prime k and multiply: res[0..2n-2] in [0, m_k)
 *                                   (CRT_parallel, main-openmp.cc:424-440)
 *   ntt_crt_merge                   out[i] = CRT(r0[i], r1[i], r2[i], r3[i]) mod p64 for `count`
 *                                   coefficients -- any slice of the product, so that every rank can
 *                                   merge its own slice after an all-to-all
 *                                   (CRT_combine_parallel, main-openmp.cc:468-485)
 * Host or device pointers. */
int ntt_crt_prime(int index, uint32_t *p);
int ntt_crt_residue_multiply(ntt_ctx *ctx, const uint64_t *a, const uint64_t *b, uint32_t *res, int n,
                             int prime_index);
int ntt_crt_merge(ntt_ctx *ctx, const uint32_t *r0, const uint32_t *r1, const uint32_t *r2,
                  const uint32_t *r3, uint64_t *out, size_t count, uint64_t p64);

/* One transform of length L = 2^lgL split over `world` GPUs (one process each), the pieces a rank
 * runs between the two all-to-all exchanges (BASELINE.json configs[4]; the reference has nothing
 * like it: its transforms live in static arrays of 300000, main-montgomery.cc:297).  Rank r holds
 * x[r*S .. (r+1)*S), S = L/world.  Forward:
 *   all-to-all #1 (rank r sends its elements with i2 in piece h to rank h; the receiver holds the
 *                  [world][S/world] array x[i1][i2 in its piece])
 *   ntt_dist_colpass(plan, data, 0)   world-point NTT over i1, times w_L^(i2*k1), canonical, in place
 *   all-to-all #2 (row k1 goes to rank k1, which now holds the full row Y[k1][0..S))
 *   ntt_dist_local(plan, row, 0)      S-point forward transform, natural order
 * after which rank k1 holds X[k1 + world*k2], k2 = 0..S-1.  The inverse runs the same steps
 * backwards with inverse = 1 (scaled by L^-1 in total).  Device pointers only for the column pass.
 * ntt_amd/shard.py::dist_transform drives the exchanges with torch.distributed (RCCL). */
typedef struct ntt_dist_plan ntt_dist_plan;
int ntt_dist_plan_create(ntt_ctx *ctx, uint32_t p, int lgL, int world, int rank, int modmul_kind,
                         ntt_dist_plan **out_plan);
int ntt_dist_plan_destroy(ntt_dist_plan *plan);
int ntt_dist_colpass(ntt_dist_plan *plan, uint32_t *data, int inverse);
int ntt_dist_local(ntt_dist_plan *plan, uint32_t *row, int inverse);
int ntt_dist_local_length(const ntt_dist_plan *plan);
/* The same transform with the exchanges INSIDE the library, for a plan made from a multi-device context
 * (ntt_dist_plan_create(ctx, p, lgL, world = ntt_ctx_device_count(ctx), rank ignored, ...); world 2, 4, 8 or 16).
 * in[r], out[r] (r < world) are device arrays of S = L / world u32 on the device of slot r, in[r] != out[r]:
 *   ntt_dist_forward   in[r] = x[r*S .. (r+1)*S)            ->  out[r][k2] = X[r + world*k2]
 *   ntt_dist_inverse   in[r][k2] = X[r + world*k2]          ->  out[r] = x[r*S .. (r+1)*S), scaled by L^-1
 * in[] is not modified.  The column kernel of slot h loads the world points of each of its columns from the arrays
 * of the slots that hold them and stores row k1 into slot k1's array: both all-to-alls of the decomposition above
 * are peer loads / stores over xGMI inside one kernel per device (transport "p2p", the default with peer access).
 * NTT_AMD_DIST_TRANSPORT (read at plan creation) = "copyin": the first exchange is hipMemcpyPeerAsync per peer slice
 * into a local array, cut into NTT_AMD_DIST_CHUNKS (4) column chunks so that the copies of chunk c + 1 run beside the
 * kernel of chunk c, the second stays peer stores; = "copy": both exchanges are peer copies (the default without
 * peer access); = "rccl": both exchanges are grouped ncclSend / ncclRecv pairs on one communicator per slot
 * (librccl is opened with dlopen for this transport only; it needs distinct devices, so a one-GPU rehearsal with
 * repeated ids is refused with NTT_ERR_UNSUPPORTED -- this transport has never moved data by us).  Queued on the slots' streams: ntt_ctx_synchronize(ctx) waits.  The arrays must be complete when the
 * call is made (it orders itself only against earlier calls on the same plan).
 * ntt_dist_plan_transport reports the transport in use (0 p2p, 1 copyin, 2 copy, 3 rccl; -1 for a single-device plan). */
int ntt_dist_forward(ntt_dist_plan *plan, const uint32_t *const *in, uint32_t *const *out);
int ntt_dist_inverse(ntt_dist_plan *plan, const uint32_t *const *in, uint32_t *const *out);
int ntt_dist_plan_transport(const ntt_dist_plan *plan, int *transport, int *chunks);
/* The product of two polynomials too long for one GPU to want, on a plan of a multi-device context: a and b have
 * n = L / 2 coefficients, block r (S = L / world of them) of each on the device of slot r for r < world / 2 (a[r], b[r]
 * for r >= world / 2 are not read: that half of the transform is the zero padding); ab[r], r < world, receives block r of
 * the L - 1 product coefficients (canonical; the last element of the last block is 0).  The harness multiply body
 * main-montgomery.cc:326-351 -- pad, NTT, NTT, pointwise product :341-344, inverse NTT -- with each transform split as
 * above, the exchanges inside the library, and no reordering pass anywhere: the spectra stay in the engine's own order
 * between the forward and the inverse transform.  a, b are not modified; queued like ntt_dist_forward. */
int ntt_dist_multiply(ntt_dist_plan *plan, const uint32_t *const *a, const uint32_t *const *b, uint32_t *const *ab);

/* ---- one process per GPU: the exchanges for a host program that keeps the reference's MPI shape ----------------
 * main-mpi.cc is one process per rank (:645-649 MPI_Init / rank / size, :685-688 the case reaches every rank by
 * MPI_Bcast, :657-660 rank r owns ceil(4 / size) primes, :770 multiplies their residues, :772-791 MPI_Send / MPI_Recv
 * of the residue polynomials to rank 0, :793-796 rank 0 merges).  A program that stays that way gives every rank an
 * ordinary single-device context on its GPU and ONE communicator; the collective calls below (every rank makes the
 * same calls in the same order, as with MPI collectives) then run what a multi-device context runs in one process,
 * with the exchange carried by
 *   ntt_comm_create_rccl   RCCL: grouped ncclSend / ncclRecv on device arrays over xGMI, librccl opened with dlopen.
 *                          ntt_comm_rccl_id on one rank makes the NTT_COMM_ID_BYTES-byte id, the host program hands it
 *                          to the others (MPI_Bcast), then every rank calls ntt_comm_create_rccl (itself collective).
 *                          One rank per device (RCCL refuses two ranks on one GPU).  The split transform is cut into
 *                          4 column chunks: the exchange of chunk c + 1 runs beside the column kernel of chunk c.
 *                          Exercised by us at world = 1 only (one GPU): parity unpinned on real links.
 *   ntt_comm_create_host   the host program's own all-to-all on HOST memory (ntt_alltoall_fn; with MPI one line:
 *                          MPI_Alltoall(send, bytes, MPI_BYTE, recv, bytes, MPI_BYTE, comm)); the library stages
 *                          through pinned buffers and calls it from the calling thread, synchronously.  Works for
 *                          ranks that share a GPU (how tests/test_gpu_comm.py runs 2 and 4 ranks on one).
 * ntt_comm_dist_forward / _inverse: the split transform of ntt_dist_plan_create(ctx, p, lgL, world, rank, ...) on
 *   this rank's block (device arrays of S = L / world words, in != out, in is not modified), same layouts as
 *   ntt_dist_forward / ntt_dist_inverse above; world <= 16.  Queued on the context's stream in RCCL mode
 *   (ntt_ctx_synchronize waits); complete on return in host mode.
 * ntt_comm_crt_multiply: the large-modulus product of ntt_crt_multiply with prime k on rank k / ceil(4 / world)
 *   (main-mpi.cc:657-660), slice h of every residue polynomial sent to rank h, merged there, and the merged slices
 *   gathered: a, b ([n] u64, host or device, the same on every rank as after the reference's MPI_Bcast), ab ([2n - 1]
 *   u64, host or device) complete on EVERY rank (the reference: rank 0 only).  Device ab in RCCL mode: queued.
 * Calls on one communicator are serialised; a communicator must be destroyed before its context. */
typedef struct ntt_comm ntt_comm;
#define NTT_COMM_ID_BYTES 128
/* send / recv: world blocks of bytes_per_rank bytes each; block h of send goes to rank h, block s of recv comes from
 * rank s.  Return 0 on success (anything else fails the collective call with NTT_ERR_IO). */
typedef int (*ntt_alltoall_fn)(void *user, const void *send, void *recv, size_t bytes_per_rank);
int ntt_comm_rccl_id(void *id /* NTT_COMM_ID_BYTES */);
int ntt_comm_create_rccl(ntt_ctx *ctx, const void *id, int world, int rank, ntt_comm **out_comm);
int ntt_comm_create_host(ntt_ctx *ctx, ntt_alltoall_fn alltoall, void *user, int world, int rank, ntt_comm **out_comm);
int ntt_comm_destroy(ntt_comm *comm);
int ntt_comm_size(const ntt_comm *comm, int *world, int *rank, int *kind /* 0 host, 1 rccl */);
int ntt_comm_dist_forward(ntt_comm *comm, ntt_dist_plan *plan, const uint32_t *in, uint32_t *out);
int ntt_comm_dist_inverse(ntt_comm *comm, ntt_dist_plan *plan, const uint32_t *in, uint32_t *out);
/* ntt_dist_multiply for one process per GPU: rank r < world / 2 passes block r (S words, device) of a and of b -- the
 * other ranks' a, b are not read (NULL allowed) -- and every rank receives block `rank` of the L - 1 product
 * coefficients in ab (device, S words).  No reordering pass; queued like ntt_comm_dist_forward. */
int ntt_comm_dist_multiply(ntt_comm *comm, ntt_dist_plan *plan, const uint32_t *a, const uint32_t *b, uint32_t *ab);
int ntt_comm_crt_multiply(ntt_comm *comm, const uint64_t *a, const uint64_t *b, uint64_t *ab, int n, uint64_t p64);

/* Per-kernel timing (HIP events on the context's stream) for bench.py's roofline block.
 * With timing enabled every kernel launch is bracketed by events; ntt_ctx_kernel_time returns
 * the accumulated milliseconds and launch count of kernel `index` (0 .. count-1) and its name;
 * returns NTT_ERR_INVALID past the end.  Reset with ntt_ctx_enable_kernel_timing(ctx, 1).
 * ntt_ctx_kernel_timing_select(ctx, index) restricts the events to kernel `index` (-1: all again), so
 * that a timed region pays for the events of the kernel it reports only. */
int ntt_ctx_enable_kernel_timing(ntt_ctx *ctx, int enable);
int ntt_ctx_kernel_timing_select(ntt_ctx *ctx, int index);
/* Bracket only every `every`-th launch of the selected kernel(s) (1: every launch): a sample of the launches of a timed
 * region instead of all of them -- the events of a launch cost about as much as 1.5 % of the row kernel it brackets.
 * ntt_ctx_kernel_time then reports the sampled launches (their count and their total). */
int ntt_ctx_kernel_timing_stride(ntt_ctx *ctx, int every);
int ntt_ctx_kernel_time(ntt_ctx *ctx, int index, const char **name, double *total_ms, long *launches);

/* Thread-local message of the last failure in this thread ("" if none). */
const char *ntt_last_error(void);

/* ---- 3. harness helpers (reference text formats) ------------------------ */

/* Replaces fRead(int*a,int*b,int*n,int*p,int id): main-montgomery.cc:15-32; u64 overload
 * main-openmp.cc:21-40.  Reads "<dir>/<id>.in" = "n p\n a[0..n) b[0..n)".  a and b must hold
 * `cap` elements; values are stored as u64 so the 51-bit case fits. */
int ntt_read_case(const char *dir, int id, uint64_t *a, uint64_t *b, size_t cap, int *n, uint64_t *p);
/* Replaces fCheck(int*ab,int n,int id): main-montgomery.cc:34-54.  Compares ab[0..2n-2] with
 * "<dir>/<id>.out", prints the reference's verdict line verbatim, and (unlike the reference,
 * which always exits 0) returns NTT_OK / NTT_ERR_MISMATCH / NTT_ERR_IO. */
int ntt_check_case(const char *dir, int id, const uint64_t *ab, int n);
/* Replaces fWrite(int*ab,int n,int id): main-montgomery.cc:56-69.  Writes "<dir>/<id>.out",
 * one decimal per line. */
int ntt_write_case(const char *dir, int id, const uint64_t *ab, int n);

/* Packed binary cases: "<dir>/<id>.inb" / "<id>.outb" = 32-byte header
 * {"NTTB", u32 version 1, u32 kind (1 inputs, 2 output), u32 width (4 | 8 bytes per coefficient),
 * u64 n, u64 p} + little-endian coefficients (a then b, or ab).  Same content as the text files of
 * fRead / fWrite (main-montgomery.cc:15-32, :56-69) without the decimal parsing; width 4 whenever
 * p < 2^32, so a case loads straight into the arrays the multiply takes.
 *   ntt_pack_case        text "<id>.in" (+ "<id>.out" when present) -> "<id>.inb" (+ "<id>.outb")
 *   ntt_read_case_bin    a, b receive n coefficients of `width` bytes each (cap_bytes per array)
 *   ntt_check_case_bin   compares ab (2n-1 coefficients of `width` bytes) with "<id>.outb", prints the
 *                        reference's verdict line, returns NTT_OK / NTT_ERR_MISMATCH / NTT_ERR_IO
 *   ntt_write_case_bin   writes "<id>.outb" */
int ntt_pack_case(const char *dir, int id);
int ntt_read_case_bin(const char *dir, int id, void *a, void *b, size_t cap_bytes, int *n, uint64_t *p,
                      int *width);
int ntt_check_case_bin(const char *dir, int id, const void *ab, int n, int width);
int ntt_write_case_bin(const char *dir, int id, const void *ab, int n, uint64_t p, int width);

#ifdef __cplusplus
}
#endif
#endif /* NTT_AMD_H */
