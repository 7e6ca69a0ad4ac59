// broadcast.cpp -- the ONE collective of a multi-GPU deployment, inside the library: the device-format weight arena goes from
// rank 0 to every rank once at start-up over RCCL / xGMI.  Utterances are then dealt to the ranks; no step of the
// data path communicates.  A host without PyTorch (the reference's Go server) needs nothing but this: rank 0 makes the 128-byte id,
// hands it to the other processes over whatever channel started them (environment, file, socket), every rank calls
// ptts_rccl_broadcast on the arena it allocated for ptts_model_open_planned.
//
// librccl is loaded on first use (dlopen), so that hosts which never broadcast carry no dependency on it.
#include <dlfcn.h>

#include "runtime.h"

namespace ptts {
namespace {

struct Rccl {
    typedef struct { char internal[128]; } UniqueId;   // ncclUniqueId (NCCL_UNIQUE_ID_BYTES = 128)
    typedef void* Comm;
    int (*GetUniqueId)(UniqueId*) = nullptr;
    int (*CommInitRank)(Comm*, int, UniqueId, int) = nullptr;
    int (*Broadcast)(const void*, void*, size_t, int, int, Comm, hipStream_t) = nullptr;
    int (*CommDestroy)(Comm) = nullptr;
    const char* (*GetErrorString)(int) = nullptr;
    void* handle = nullptr;
};

Rccl& rccl() {
    static Rccl r;
    static std::once_flag once;
    std::call_once(once, [] {
        const char* env = getenv("PTTS_RCCL_LIB");
        const char* names[] = {env ? env : "librccl.so.1", "librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
        for (const char* n : names) {
            r.handle = dlopen(n, RTLD_NOW | RTLD_LOCAL);
            if (r.handle) break;
        }
        if (!r.handle) return;
        r.GetUniqueId = (int (*)(Rccl::UniqueId*))dlsym(r.handle, "ncclGetUniqueId");
        r.CommInitRank = (int (*)(Rccl::Comm*, int, Rccl::UniqueId, int))dlsym(r.handle, "ncclCommInitRank");
        r.Broadcast = (int (*)(const void*, void*, size_t, int, int, Rccl::Comm, hipStream_t))dlsym(r.handle, "ncclBroadcast");
        r.CommDestroy = (int (*)(Rccl::Comm))dlsym(r.handle, "ncclCommDestroy");
        r.GetErrorString = (const char* (*)(int))dlsym(r.handle, "ncclGetErrorString");
    });
    if (!r.handle || !r.GetUniqueId || !r.CommInitRank || !r.Broadcast || !r.CommDestroy)
        throw Error(PTTS_ENODEVICE, "ptts-hip: librccl is not available (set PTTS_RCCL_LIB to its path)");
    return r;
}

void check(Rccl& r, int rc, const char* what) {
    if (rc != 0) throw Error(PTTS_ENODEVICE, strfmt("rccl: %s failed: %s", what, r.GetErrorString ? r.GetErrorString(rc) : "error"));
}

}  // namespace

void rccl_unique_id(uint8_t out[128]) {
    Rccl& r = rccl();
    Rccl::UniqueId id;
    check(r, r.GetUniqueId(&id), "ncclGetUniqueId");
    std::memcpy(out, id.internal, 128);
}

void rccl_broadcast(void* device_buf, size_t bytes, int rank, int n_ranks, const uint8_t id_bytes[128], int device) {
    if (!device_buf || !id_bytes || n_ranks <= 0 || rank < 0 || rank >= n_ranks) throw Error(PTTS_EINVAL, "ptts-hip: bad broadcast arguments");
    Rccl& r = rccl();
    PTTS_HIP(hipSetDevice(device));
    Rccl::UniqueId id;
    std::memcpy(id.internal, id_bytes, 128);
    Rccl::Comm comm = nullptr;
    check(r, r.CommInitRank(&comm, n_ranks, id, rank), "ncclCommInitRank");
    hipStream_t s = nullptr;
    hipError_t e = hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
    int rc = 0;
    if (e == hipSuccess) {
        rc = r.Broadcast(device_buf, device_buf, bytes, /* ncclUint8 */ 1, /* root */ 0, comm, s);
        e = hipStreamSynchronize(s);
        (void)hipStreamDestroy(s);
    }
    (void)r.CommDestroy(comm);
    if (e != hipSuccess) throw Error(PTTS_ENODEVICE, strfmt("hip: broadcast stream failed: %s", hipGetErrorString(e)));
    check(r, rc, "ncclBroadcast");
}

}  // namespace ptts
