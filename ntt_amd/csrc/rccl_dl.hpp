// rccl_dl.hpp -- RCCL, loaded on demand (multidev.cpp: transport "rccl" of the split transform on a multi-device
// context; comm.cpp: the process-per-GPU communicator).  The library does not link librccl: a Python process that also
// runs torch.distributed already holds torch's own copy, and every default path here is plain HIP.  dlopen with local
// symbols, and only the calls below.
#pragma once
#include <hip/hip_runtime.h>

namespace ntt {

struct RcclUniqueId { char internal[128]; };             // ncclUniqueId (rccl.h: NCCL_UNIQUE_ID_BYTES = 128), passed by value

struct Rccl {
    void *lib = nullptr;
    int (*CommInitAll)(void **comms, int ndev, const int *devlist) = nullptr;
    int (*CommDestroy)(void *comm) = nullptr;
    int (*GroupStart)() = nullptr;
    int (*GroupEnd)() = nullptr;
    int (*Send)(const void *buf, size_t count, int dtype, int peer, void *comm, hipStream_t stream) = nullptr;
    int (*Recv)(void *buf, size_t count, int dtype, int peer, void *comm, hipStream_t stream) = nullptr;
    const char *(*GetErrorString)(int) = nullptr;
    int (*GetUniqueId)(RcclUniqueId *id) = nullptr;
    int (*CommInitRank)(void **comm, int nranks, RcclUniqueId id, int rank) = nullptr;
    static constexpr int kUint32 = 3;                    // ncclUint32 (rccl.h)
};
Rccl *rccl_load();                                       // nullptr when librccl cannot be opened

#define RCCL_TRY(R, expr)                                                                                          \
    do {                                                                                                           \
        const int e_ = (expr);                                                                                     \
        if (e_ != 0) return ntt::fail(NTT_ERR_HIP, std::string(#expr) + ": " + (R)->GetErrorString(e_));           \
    } while (0)

}  // namespace ntt
