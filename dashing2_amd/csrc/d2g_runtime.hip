// d2g_runtime.hip -- context, device memory and timing plumbing of libd2g.
#include "d2g_internal.h"
#include <cstdlib>
#include <cstring>
#include <new>

// the library's run-time switches (DESIGN.md section 4 lists what each one does).  Host-side switches that need no context
// (D2G_MAX_RUN and D2G_NO_AVX512 in d2g_host.cpp, D2G_RCCL_LIB, D2G_COMM_LOOPBACK) are read where they apply.
static const char *const kTuningNames[] = {
    "D2G_BS_SORT", "D2G_BS_NSPLIT", "D2G_BS_TAGBITS",
    "D2G_BS_SPARSE", "D2G_BS_SPARSE_MIN_N", "D2G_SP_LINK", "D2G_SP_TILE_FRAC", "D2G_SP_LIST_DIV", "D2G_SP_LONG_LIST", "D2G_SP_LIST_FORM", "D2G_SP_PREDICT", "D2G_SP_REMEMBER", "D2G_SP_EMIT_BIG", "D2G_SP_OLINK", "D2G_SP_RIDE",
    "D2G_MGPU_CHUNKS",
    "D2G_K3_COMPACT", "D2G_K3_L1BITS", "D2G_K3_BUCKET_KEYS", "D2G_K3_SUB_KEYS", "D2G_K3_SPLIT_MIN", "D2G_K3_SUBBATCH", "D2G_K3_ROUND_KEYS",
    "D2G_K3_GUESS_SCALE", "D2G_K3_GRID_PER_CU", "D2G_K3_LIGHT", "D2G_K3_GQ_SCALE",
};
void d2g_tuning_load(d2g_tuning &t) {
    t.kv.clear();
    for (const char *n : kTuningNames)
        if (const char *v = std::getenv(n)) t.kv.emplace_back(n, v);
}

extern "C" {

int d2g_ctx_reload_tuning(d2g_ctx *c) {
    if (!c) return D2G_ERR_INVALID;
    d2g_tuning_load(c->tune);
    return D2G_OK;
}

// {"D2G_X": "value", ...} of the switches this context resolved; returns the length needed (excluding the terminator)
int d2g_ctx_tuning(const d2g_ctx *c, char *buf, size_t cap) {
    if (!c) return D2G_ERR_INVALID;
    std::string j = "{";
    for (size_t i = 0; i < c->tune.kv.size(); ++i) {
        if (i) j += ", ";
        j += "\"" + c->tune.kv[i].first + "\": \"";
        for (char ch : c->tune.kv[i].second) { if (ch == '"' || ch == '\\') j += '\\'; if ((unsigned char)ch >= 0x20) j += ch; }
        j += "\"";
    }
    // ... and what the switches that select K2's kernels and thresholds RESOLVED to, set or not (the bench line's `tuning` names the defaults)
    j += std::string(c->tune.kv.empty() ? "" : ", ") + "\"resolved\": " + d2g_k2_tuning_json(c);
    j += "}";
    if (buf && cap) { std::snprintf(buf, cap, "%s", j.c_str()); }
    return (int)j.size();
}

int d2g_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

int d2g_ctx_create(int device, d2g_ctx **out) {
    if (!out) return D2G_ERR_INVALID;
    *out = nullptr;
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) return D2G_ERR_NODEVICE;
    if (device < 0 || device >= n) return D2G_ERR_INVALID;
    if (hipSetDevice(device) != hipSuccess) return D2G_ERR_NODEVICE;
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) != hipSuccess) return D2G_ERR_NODEVICE;
    // this library carries gfx950 code objects only
    if (std::strncmp(prop.gcnArchName, "gfx950", 6) != 0) return D2G_ERR_NODEVICE;
    d2g_ctx *c = new (std::nothrow) d2g_ctx();
    if (!c) return D2G_ERR_NOMEM;
    c->device = device;
    c->num_cus = prop.multiProcessorCount;
    d2g_tuning_load(c->tune);
    *out = c;
    return D2G_OK;
}

void d2g_ctx_destroy(d2g_ctx *c) {
    if (!c) return;
    (void)hipSetDevice(c->device);
    for (d2g_evlog *e : {&c->ev_k1, &c->ev_k2, &c->ev_k2prep, &c->ev_k3, &c->ev_k0}) {
        for (hipEvent_t x : e->a) (void)hipEventDestroy(x);
        for (hipEvent_t x : e->b) (void)hipEventDestroy(x);
    }
    if (c->k3) d2g_k3_state_destroy(c->k3);
    delete c;
}

const char *d2g_last_error(const d2g_ctx *c) { return c ? c->last_error.c_str() : "null ctx"; }
int d2g_ctx_device(const d2g_ctx *c) { return c ? c->device : -1; }

int d2g_sync(d2g_ctx *c, void *stream) {
    if (!c) return D2G_ERR_INVALID;
    D2G_HIP(c, hipSetDevice(c->device));
    D2G_HIP(c, hipStreamSynchronize(as_stream(stream)));
    return D2G_OK;
}

int d2g_malloc(d2g_ctx *c, size_t nbytes, void **dptr) {
    if (!c || !dptr) return D2G_ERR_INVALID;
    D2G_HIP(c, hipSetDevice(c->device));
    D2G_HIP(c, hipMalloc(dptr, nbytes ? nbytes : 1));
    return D2G_OK;
}
int d2g_free(d2g_ctx *c, void *dptr) {
    if (!c) return D2G_ERR_INVALID;
    if (!dptr) return D2G_OK;
    D2G_HIP(c, hipSetDevice(c->device));
    D2G_HIP(c, hipFree(dptr));
    return D2G_OK;
}
int d2g_malloc_host(d2g_ctx *c, size_t nbytes, void **hptr) {
    if (!c || !hptr) return D2G_ERR_INVALID;
    D2G_HIP(c, hipSetDevice(c->device));
    D2G_HIP(c, hipHostMalloc(hptr, nbytes ? nbytes : 1, hipHostMallocDefault));
    return D2G_OK;
}
// page-lock memory the caller already owns (e.g. staging buffers that were being filled before the context existed)
int d2g_host_register(d2g_ctx *c, void *hptr, size_t nbytes) {
    if (!c || !hptr || !nbytes) return D2G_ERR_INVALID;
    D2G_HIP(c, hipSetDevice(c->device));
    D2G_HIP(c, hipHostRegister(hptr, nbytes, hipHostRegisterDefault));
    return D2G_OK;
}
int d2g_host_unregister(d2g_ctx *c, void *hptr) {
    if (!c || !hptr) return D2G_ERR_INVALID;
    D2G_HIP(c, hipHostUnregister(hptr));
    return D2G_OK;
}
int d2g_free_host(d2g_ctx *c, void *hptr) {
    if (!c) return D2G_ERR_INVALID;
    if (!hptr) return D2G_OK;
    D2G_HIP(c, hipHostFree(hptr));
    return D2G_OK;
}
int d2g_memcpy_h2d(d2g_ctx *c, void *dst, const void *src, size_t n, void *stream) {
    if (!c || (n && (!dst || !src))) return D2G_ERR_INVALID;
    D2G_HIP(c, hipSetDevice(c->device));
    D2G_HIP(c, hipMemcpyAsync(dst, src, n, hipMemcpyHostToDevice, as_stream(stream)));
    D2G_HIP(c, hipStreamSynchronize(as_stream(stream)));
    return D2G_OK;
}
int d2g_memcpy_d2h(d2g_ctx *c, void *dst, const void *src, size_t n, void *stream) {
    if (!c || (n && (!dst || !src))) return D2G_ERR_INVALID;
    D2G_HIP(c, hipSetDevice(c->device));
    D2G_HIP(c, hipMemcpyAsync(dst, src, n, hipMemcpyDeviceToHost, as_stream(stream)));
    D2G_HIP(c, hipStreamSynchronize(as_stream(stream)));
    return D2G_OK;
}

// see d2g.h
int d2g_warmup(d2g_ctx *c, int what) {
    if (!c) return D2G_ERR_INVALID;
    D2G_HIP(c, hipSetDevice(c->device));
    if (what & D2G_WARM_COPY) {
        // the first host<->device copy of a process sets up the runtime's copy machinery (~30 ms on MI355X / ROCm 7: measured with any
        // size and with pinned or pageable memory alike; a 4 KB copy leaves part of it to the first large one, 1 MB does not)
        std::vector<char> src((size_t)1 << 20);                 // per call: two helper threads may warm two contexts at once
        void *d = nullptr;
        D2G_HIP(c, hipMalloc(&d, src.size()));
        hipError_t e = hipMemcpy(d, src.data(), src.size(), hipMemcpyHostToDevice);
        if (e == hipSuccess) e = hipMemcpy(src.data(), d, 4096, hipMemcpyDeviceToHost);
        (void)hipFree(d);
        D2G_HIP(c, e);
    }
    if (what & D2G_WARM_K0) d2g_warm_k0();
    if (what & D2G_WARM_K1) d2g_warm_k1();
    if (what & D2G_WARM_K2) { d2g_warm_k2(); d2g_warm_k2_bitslice(); }
    if (what & D2G_WARM_K3) d2g_warm_k3();
    return D2G_OK;
}
int d2g_device_name(int device, char *buf, size_t cap) {
    if (!buf || !cap) return D2G_ERR_INVALID;
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) != hipSuccess) return D2G_ERR_NODEVICE;
    std::snprintf(buf, cap, "%s (%s, %d CUs)", prop.name, prop.gcnArchName, prop.multiProcessorCount);
    return D2G_OK;
}

int d2g_set_timing(d2g_ctx *c, int enabled) {
    if (!c) return D2G_ERR_INVALID;
    c->timing = enabled == 1 ? (D2G_TIME_K1 | D2G_TIME_K2 | D2G_TIME_K2PREP | D2G_TIME_K3) : (enabled & ~1);
    return D2G_OK;
}

int d2g_kernel_ms(d2g_ctx *c, const char *which, int reset, int *count, float *avg_ms, float *last_ms) {
    if (!c || !which) return D2G_ERR_INVALID;
    d2g_evlog *e = nullptr;
    if (!std::strcmp(which, "k1")) e = &c->ev_k1;
    else if (!std::strcmp(which, "k2")) e = &c->ev_k2;
    else if (!std::strcmp(which, "k2prep")) e = &c->ev_k2prep;
    else if (!std::strcmp(which, "k3")) e = &c->ev_k3;
    else if (!std::strcmp(which, "k0")) e = &c->ev_k0;
    D2G_CHECK(c, e != nullptr, "d2g_kernel_ms: unknown kernel name");
    D2G_HIP(c, hipSetDevice(c->device));
    double sum = 0;
    float last = 0.f;
    for (size_t i = 0; i < e->a.size(); ++i) {
        D2G_HIP(c, hipEventSynchronize(e->b[i]));
        D2G_HIP(c, hipEventElapsedTime(&last, e->a[i], e->b[i]));
        sum += last;
    }
    const int n = (int)e->a.size();
    if (count) *count = n;
    if (avg_ms) *avg_ms = n ? (float)(sum / n) : 0.f;
    if (last_ms) *last_ms = last;
    if (reset) {
        for (hipEvent_t x : e->a) (void)hipEventDestroy(x);
        for (hipEvent_t x : e->b) (void)hipEventDestroy(x);
        e->a.clear(); e->b.clear();
    }
    return D2G_OK;
}

}  // extern "C"
