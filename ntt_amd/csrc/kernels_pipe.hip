// kernels_pipe.hip -- software-pipelined fused launch: one grid runs three INDEPENDENT roles,
//     F: forward column pass of chunk t      (memory-bound: 1 MiB in, 2 MiB out per product)
//     R: fused row multiply of chunk t-1     (VALU-bound:   ~2200 VALU instructions per wave)
//     I: inverse column pass of chunk t-2    (memory-bound: 1 MiB in, 1 MiB out per product)
// so that every CU holds memory-bound and VALU-bound workgroups at the same time.  The roles of
// one launch touch different chunks, hence there is no dependency inside a launch and nothing
// to wait for: the dependencies F(c) -> R(c) -> I(c) are carried by stream order between
// consecutive launches (a three-deep ring of work buffers).
//
// rocprofv3 on the three stand-alone kernels (profiles/r01b_summary.md) shows why: col_fwd waits
// on memory 64 % of its wave time, row_mul is issue-bound 71 % of its wave time; run back to
// back, each leaves the other resource idle.
//
// Block -> (group, role): blocks are dealt round-robin over the 8 XCDs (b % 8) and, inside an XCD,
// in order over its shader engines / CUs.  A role that is a fixed function of b % 4 (or of the
// slot index b / 8 modulo 4) therefore lands on the same quarter of the machine every time:
// measured, the VALU-heavy R role then runs 3x slower because a quarter of the CUs get all of it.
// With b = 32*Gq + 8*s + x (x = b % 8):  role = (s + x + rot(Gq)) % 4,  group = 8*Gq + x, where
// rot() changes with every bit pair of Gq -- still a bijection for fixed (Gq, x), and every
// (XCD, slot class) sees all four roles in rotation.
// Roles: 0 = F operand a, 1 = F operand b, 2 = R, 3 = I.
// A group is one 4096-point row (R) or one column tile (F, I); both number L/4096 per product.
#include "col16_body.hpp"
#include "row_body.hpp"

namespace ntt {

template <class MM, bool LAZY, int LOGR>
__global__ void __launch_bounds__(256) pipe_kernel(const PipeArgs<MM> a, const ModParams m)
{
    __shared__ uint32_t lds[kLdsWords];
    const uint32_t b = blockIdx.x;
    const uint32_t x = b & 7, s = (b >> 3) & 3;
    const uint32_t gq = b >> 5;
    const uint32_t role = (s + x + gq + (gq >> 2) + (gq >> 4) + (gq >> 6) + (gq >> 8)) & 3;
    const uint32_t g = (gq << 3) + x;
    const uint32_t per = a.groups_per_product;          // L / 4096
    if (role < 2) {
        if (g >= a.groups_f) return;
        col16_body<MM, LAZY, LOGR, false, true>(a.f, m, lds, g % per, g / per, (int)role);
    } else if (role == 2) {
        if (g >= a.groups_r) return;
        row_mul_body<MM, LAZY>(a.r, m, lds, g);
    } else {
        if (g >= a.groups_i) return;
        col16_body<MM, LAZY, LOGR, true, false>(a.i, m, lds, g % per, g / per, 0);
    }
}

template <class MM, bool LAZY>
static void launch_pipe_l(int logN1, const PipeArgs<MM> &a, const ModParams &mp, unsigned groups, hipStream_t s)
{
    const dim3 grid(4 * groups);
    if (logN1 == 6) hipLaunchKernelGGL((pipe_kernel<MM, LAZY, 2>), grid, dim3(256), 0, s, a, mp);
    else hipLaunchKernelGGL((pipe_kernel<MM, LAZY, 1>), grid, dim3(256), 0, s, a, mp);
}

void launch_pipe(int kind, bool lazy, int logN1, const void *args, const ModParams &mp, unsigned groups,
                 hipStream_t s)
{
    if (kind == 0) {
        const auto &a = *static_cast<const PipeArgs<Mont32> *>(args);
        if (lazy) launch_pipe_l<Mont32, true>(logN1, a, mp, groups, s);
        else launch_pipe_l<Mont32, false>(logN1, a, mp, groups, s);
    } else {
        const auto &a = *static_cast<const PipeArgs<Barrett32> *>(args);
        if (lazy) launch_pipe_l<Barrett32, true>(logN1, a, mp, groups, s);
        else launch_pipe_l<Barrett32, false>(logN1, a, mp, groups, s);
    }
}

}  // namespace ntt
