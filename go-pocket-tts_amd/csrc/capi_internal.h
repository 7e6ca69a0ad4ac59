// capi_internal.h -- what the two extern "C" translation units share: capi.cpp (the product ABI, include/ptts.h) and capi_hooks.cpp (the test and
// measurement hooks of include/ptts_debug.h, built into libptts_hooks.so only -- libptts_hip.so does not contain them).
#pragma once

#include <cmath>
#include <map>

#include "runtime.h"
#include "../../include/ptts.h"

struct ptts_model { ptts::Model* m; };
struct ptts_plan { ptts::Plan p; };
struct ptts_batch { ptts::Batch* b; ptts::Model* m; };
// ptts_voice is ptts::Voice (opaque to C)

namespace ptts {
const std::string& last_error_ref();

namespace capi {

template <class F> int guard(F&& fn) {
    try {
        fn();
        return PTTS_OK;
    } catch (const Error& e) {
        set_last_error(e.what());
        return e.code;
    } catch (const std::bad_alloc&) {
        set_last_error("ptts-hip: out of host memory");
        return PTTS_ENOMEM;
    } catch (const std::exception& e) {
        set_last_error(std::string("ptts-hip: ") + e.what());
        return PTTS_EINVAL;
    }
}

inline ptts_opts resolve_opts(const ptts_opts* o) {
    ptts_opts r;
    ptts_default_opts(&r);
    if (o) r = *o;
    if (r.max_batch <= 0) r.max_batch = 64;
    if (r.max_batch > kStepMaxRows) throw Error(PTTS_EINVAL, strfmt("ptts-hip: max_batch %d exceeds the %d utterances one AR step takes", r.max_batch, kStepMaxRows));
    if (r.weights != PTTS_WEIGHTS_F32 && r.weights != PTTS_WEIGHTS_BF16 && r.weights != PTTS_WEIGHTS_INT8) throw Error(PTTS_EINVAL, "ptts-hip: unknown weights mode");
    if (r.kv != PTTS_KV_F32 && r.kv != PTTS_KV_BF16) throw Error(PTTS_EINVAL, "ptts-hip: unknown kv mode");
    return r;
}

inline void require_device() {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) throw Error(PTTS_ENODEVICE, "ptts-hip: no HIP device available (this library has no CPU fallback)");
}

// scratch for the op-level entry points: device 0, default stream, synchronous copies
struct Tmp {
    void* p = nullptr;
    explicit Tmp(size_t bytes) { PTTS_HIP(hipMalloc(&p, bytes ? bytes : 256)); }
    ~Tmp() { if (p) (void)hipFree(p); }
    template <class T> T* as() { return reinterpret_cast<T*>(p); }
};
inline void up(void* d, const void* h, size_t n) { if (n) PTTS_HIP(hipMemcpy(d, h, n, hipMemcpyHostToDevice)); }
inline void down(void* h, const void* d, size_t n) { if (n) PTTS_HIP(hipMemcpy(h, d, n, hipMemcpyDeviceToHost)); }

// LatentToMimi + MimiDecode (model.go:141,410) with an optional staged observation point (the decoder transformer's output rows): ptts_decode_latents
// passes null; the test hook ptts_decode_stages (capi_hooks.cpp) asks for it
int decode_stages(ptts_model* h, const float* latents, int32_t n_utt, int32_t frames, float* pcm, float* mimi_latent, float* transformer_out);

}  // namespace capi
}  // namespace ptts
