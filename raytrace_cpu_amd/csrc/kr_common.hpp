// kr_common.hpp -- host-side plumbing shared by the .hip translation units of libkrtrace.so.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include <string>

#include "../../include/kr_trace.h"

namespace kr {

void set_error(const std::string& msg);
int hip_fail(hipError_t e, const char* what, const char* file, int line);
int require_device();   // KR_OK or KR_ENODEVICE (message set)
int cu_count();         // compute units of the current device

#define KR_HIP(call)                                                      \
    do {                                                                  \
        hipError_t e__ = (call);                                          \
        if (e__ != hipSuccess) return kr::hip_fail(e__, #call, __FILE__, __LINE__); \
    } while (0)

// RAII device scratch used by the host-buffer entry points
struct DeviceBuffer {
    void* p = nullptr;
    ~DeviceBuffer() { if (p) (void) hipFree(p); }
    int alloc(size_t bytes);
};

// device-pointer implementations (defined in kr_trace.hip / kr_post.hip); stream may be null
int trace_dev(const kr_params* p, void* d_rays, int64_t n, hipStream_t stream, kr_stats* stats, bool f32);
int trace_async(const kr_params* p, void* d_rays, int64_t n, hipStream_t stream, bool f32, void** ticket);
int trace_batch_async(int count, const kr_params* const* p, void* const* d_rays, const int64_t* n, void* const* streams, void** tickets);
int trace_wait(void* ticket, kr_stats* stats);
int trace_poll(void* ticket, int64_t* rays_started, int32_t* finished);
void trace_release(void* ticket);
void side_stream_forget(hipStream_t user);
int trace_shutdown();
void source_tables_shutdown();
void angle_values(int kind, double x0, double dx, int n, double* sincos_pairs);     // kind 0: x = cos(alpha); 1: x = beta

}  // namespace kr
