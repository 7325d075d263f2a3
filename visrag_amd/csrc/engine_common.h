// Host-side helpers shared by the engines of libvisrag_hip.so (engine.hip: the VisRAG-Ret encoder; gen.hip: the
// EVisRAG generator's language model): error reporting, owning device buffers, packed weights and their loaders.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdio>
#include <initializer_list>
#include <string>
#include <vector>

#include "../../include/visrag_hip.h"
#include "kernels.h"
#include "pack.h"

using namespace vr;

// ------------------------------------------------------------------------------ errors ---
inline thread_local std::string g_err;

static inline int fail(int code, const char* fmt, ...) {
    char buf[1024];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    g_err = buf;
    return code;
}

#define HIPCHK(expr)                                                                         \
    do {                                                                                     \
        hipError_t e_ = (expr);                                                              \
        if (e_ != hipSuccess)                                                                \
            return fail(VR_ERR_HIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_),   \
                        __FILE__, __LINE__);                                                 \
    } while (0)
#define VRCHK(expr)            \
    do {                       \
        int r_ = (expr);       \
        if (r_ != VR_OK) return r_; \
    } while (0)

static inline int pad128(int x) { return (x + 127) / 128 * 128; }
static inline int64_t pad128l(int64_t x) { return (x + 127) / 128 * 128; }
static inline int pad256(int x) { return (x + 255) / 256 * 256; }
static inline int64_t pad256l(int64_t x) { return (x + 255) / 256 * 256; }

// ------------------------------------------------------------------------- device bufs ---
// Owning device buffer.  A COPY is a non-owning alias (vr_model_clone shares the weight buffers of
// its source this way); moves transfer ownership; the destructor frees what the buffer owns, so
// temporaries are released on every return path.
struct DevBuf {
    void* p = nullptr;
    size_t bytes = 0;
    bool owned = true;
    DevBuf() = default;
    DevBuf(const DevBuf& o) : p(o.p), bytes(o.bytes), owned(false) {}
    DevBuf& operator=(const DevBuf& o) {
        if (this != &o) { free(); p = o.p; bytes = o.bytes; owned = false; }
        return *this;
    }
    DevBuf(DevBuf&& o) noexcept : p(o.p), bytes(o.bytes), owned(o.owned) { o.p = nullptr; o.bytes = 0; o.owned = true; }
    DevBuf& operator=(DevBuf&& o) noexcept {
        if (this != &o) { free(); p = o.p; bytes = o.bytes; owned = o.owned; o.p = nullptr; o.bytes = 0; o.owned = true; }
        return *this;
    }
    ~DevBuf() { free(); }
    int alloc(size_t n) {
        free();
        if (n == 0) n = 16;
        hipError_t e = hipMalloc(&p, n);
        if (e != hipSuccess) { p = nullptr; return fail(VR_ERR_HIP, "hipMalloc(%zu) failed: %s", n, hipGetErrorString(e)); }
        bytes = n;
        e = hipMemset(p, 0, n);
        if (e != hipSuccess) return fail(VR_ERR_HIP, "hipMemset failed: %s", hipGetErrorString(e));
        // the memset runs on the NULL stream; callers use the buffer on their own (possibly
        // non-blocking) stream right away, which is not ordered after it: finish it here
        e = hipStreamSynchronize(nullptr);
        if (e != hipSuccess) return fail(VR_ERR_HIP, "hipStreamSynchronize failed: %s", hipGetErrorString(e));
        return VR_OK;
    }
    // grow-only scratch: contents undefined, no clear, no synchronisation unless it has to grow
    int reserve(size_t n) {
        if (p && owned && bytes >= n) return VR_OK;
        free();
        if (n == 0) n = 16;
        hipError_t e = hipMalloc(&p, n);
        if (e != hipSuccess) { p = nullptr; return fail(VR_ERR_HIP, "hipMalloc(%zu) failed: %s", n, hipGetErrorString(e)); }
        bytes = n;
        return VR_OK;
    }
    void free() {
        if (p && owned) (void)hipFree(p);
        p = nullptr;
        bytes = 0;
        owned = true;
    }
    template <typename T> T* as() const { return reinterpret_cast<T*>(p); }
};

struct Linear {            // bf16 [n_pad][k_pad] (+ f32 bias [n_pad])
    DevBuf w, b;
    int n = 0, k = 0, n_pad = 0, k_pad = 0;
    bool has_w = false, has_b = false;
};
struct Vec { DevBuf v; bool ok = false; };   // f32 vector (norm weights / biases)


static inline int set_dev(int dev) {
    HIPCHK(hipSetDevice(dev));
    return VR_OK;
}


// ------------------------------------------------------------------------- weight load ---
// Stage `data` (host or device, f32 or bf16) on the device; returns a device pointer that is
// valid until the next call (tmp buffer) or `data` itself when it is already on the device.
struct Staged { const void* dev = nullptr; DevBuf tmp; };

static inline int stage(const void* data, size_t bytes, int on_device, Staged& st) {
    if (on_device) { st.dev = data; return VR_OK; }
    VRCHK(st.tmp.alloc(bytes));
    HIPCHK(hipMemcpy(st.tmp.p, data, bytes, hipMemcpyHostToDevice));
    st.dev = st.tmp.p;
    return VR_OK;
}

// rows of a [n][k] weight go to rows (r/blk)*blk_stride + blk_off + r%blk of the padded dst
static inline int load_linear_part(Linear& L, int n_total, int k, const void* dev_src, int is_bf16, int rows, int src_ld,
                            int transpose, int blk, int blk_stride, int blk_off, int lo_part = 0) {
    if (!L.w.p) {
        L.n = n_total; L.k = k; L.n_pad = pad128(n_total); L.k_pad = pad128(k);
        VRCHK(L.w.alloc((size_t)pad256(n_total) * L.k_pad * 2));   // rows readable by a 256-row tile
    } else if (L.n != n_total || L.k != k) {
        return fail(VR_ERR_INVALID, "inconsistent shapes for a packed weight");
    }
    HIPCHK(launch_pack_weight(dev_src, is_bf16, rows, k, src_ld, transpose, L.w.p, L.k_pad, blk, blk_stride, blk_off, 0, lo_part));
    HIPCHK(hipDeviceSynchronize());
    L.has_w = true;
    return VR_OK;
}

static inline int load_bias_part(Linear& L, int n_total, const void* dev_src, int is_bf16, int rows, int off) {
    if (!L.b.p) VRCHK(L.b.alloc((size_t)pad256(n_total) * 4));
    HIPCHK(launch_to_f32(dev_src, is_bf16, L.b.as<float>() + off, rows, 0));
    HIPCHK(hipDeviceSynchronize());
    L.has_b = true;
    return VR_OK;
}

static inline int load_vec(Vec& v, const void* dev_src, int is_bf16, int n, int n_alloc) {
    VRCHK(v.v.alloc((size_t)n_alloc * 4));
    HIPCHK(launch_to_f32(dev_src, is_bf16, v.v.as<float>(), n, 0));
    HIPCHK(hipDeviceSynchronize());
    v.ok = true;
    return VR_OK;
}

static inline int to_host_f32(const void* dev_src, int is_bf16, size_t n, std::vector<float>& out) {
    DevBuf t;
    VRCHK(t.alloc(n * 4));
    HIPCHK(launch_to_f32(dev_src, is_bf16, t.as<float>(), n, 0));
    out.resize(n);
    HIPCHK(hipMemcpy(out.data(), t.p, n * 4, hipMemcpyDeviceToHost));
    t.free();
    return VR_OK;
}

static inline bool shape_is(const int64_t* s, int nd, std::initializer_list<int64_t> want) {
    if (nd != (int)want.size()) return false;
    int i = 0;
    for (int64_t w : want) if (s[i++] != w) return false;
    return true;
}

