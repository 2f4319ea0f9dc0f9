// d2g_internal.h -- shared between the HIP translation units of libd2g (not installed).
#pragma once
#include "../../include/d2g.h"
#include <hip/hip_runtime.h>
#include <cstdio>
#include <string>
#include <utility>

#include <vector>
// every timed launch appends a (start, stop) pair; nothing synchronises until the caller asks
struct d2g_evlog {
    std::vector<hipEvent_t> a, b;
};

// Every D2G_* switch that selects a kernel, a threshold or a test hook inside the library is read from the environment ONCE per context
// (d2g_ctx_create; again only on d2g_ctx_reload_tuning) and used from this snapshot: a run's timing does not depend on who changed
// the environment in between, d2g_ctx_tuning reports the resolved set, and the multi-GPU engine compares it across ranks.
struct d2g_tuning {
    std::vector<std::pair<std::string, std::string>> kv;      // the switches that were set, in the order of d2g_tuning_names()
    const char *get(const char *name) const {                 // nullptr = not set
        for (const auto &p : kv) if (p.first == name) return p.second.c_str();
        return nullptr;
    }
};
void d2g_tuning_load(d2g_tuning &t);
std::string d2g_k2_tuning_json(const d2g_ctx *ctx);   // (d2g_k2_bitslice.hip) {"D2G_BS_SPARSE_MIN_N": 8192, ...}: the resolved values of the same switches
uint64_t d2g_k2_tuning_hash(const d2g_ctx *ctx);   // (d2g_k2_bitslice.hip) FNV-1a over the RESOLVED values of the switches that select K2 kernels and thresholds: what the ranks of one job must agree on

struct d2g_ctx {
    int device = -1;
    d2g_tuning tune;
    int num_cus = 0;
    std::string last_error;
    int timing = 0;                         // D2G_TIME_* mask (d2g_set_timing)
    d2g_evlog ev_k1, ev_k2, ev_k2prep, ev_k3, ev_k0;
    struct d2g_k3_state *k3 = nullptr;      // work buffers of d2g_bmh_sketch_dev (d2g_k3_bmh.hip)
};
void d2g_k3_state_destroy(struct d2g_k3_state *st);

#define D2G_HIP(ctx, call)                                                            \
    do {                                                                              \
        hipError_t e__ = (call);                                                      \
        if (e__ != hipSuccess) {                                                      \
            (ctx)->last_error = std::string(#call) + ": " + hipGetErrorString(e__);   \
            return e__ == hipErrorOutOfMemory ? D2G_ERR_NOMEM : D2G_ERR_HIP;          \
        }                                                                             \
    } while (0)

#define D2G_CHECK(ctx, cond, msg)                      \
    do {                                               \
        if (!(cond)) {                                 \
            (ctx)->last_error = (msg);                 \
            return D2G_ERR_INVALID;                    \
        }                                              \
    } while (0)

// hipEvent bracket around the dominant kernel of a path (enabled with d2g_set_timing);
// elapsed times are read lazily by d2g_kernel_ms (which synchronises on the stop events).
struct d2g_timer {
    d2g_evlog *ev; hipStream_t s; bool on;
    d2g_timer(d2g_ctx *c, d2g_evlog *e, hipStream_t st) : ev(e), s(st), on(false) {
        const int bit = e == &c->ev_k1 ? D2G_TIME_K1 : e == &c->ev_k2 ? D2G_TIME_K2 : e == &c->ev_k2prep ? D2G_TIME_K2PREP : e == &c->ev_k0 ? D2G_TIME_K0 : D2G_TIME_K3;
        on = (c->timing & bit) != 0;
        if (on) {
            hipEvent_t x = nullptr, y = nullptr;
            if (hipEventCreate(&x) != hipSuccess || hipEventCreate(&y) != hipSuccess) { on = false; return; }
            ev->a.push_back(x); ev->b.push_back(y);
            (void)hipEventRecord(x, s);
        }
    }
    void stop() { if (on) (void)hipEventRecord(ev->b.back(), s); }
};

// one per translation unit with kernels: makes the runtime load that unit's code object now (hipFuncGetAttributes on one of
// its kernels) instead of at its first launch
void d2g_warm_k0(); void d2g_warm_k1(); void d2g_warm_k2(); void d2g_warm_k2_bitslice(); void d2g_warm_k3();

static inline hipStream_t as_stream(void *s) { return reinterpret_cast<hipStream_t>(s); }

template <class T> static inline T div_up(T a, T b) { return (a + b - 1) / b; }
