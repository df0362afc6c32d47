// apps/app_common.h -- small helpers shared by the device-resident applications.
#ifndef KR_APP_COMMON_H_
#define KR_APP_COMMON_H_

#include <chrono>
#include <cstdlib>
#include <cstring>
#include <stdexcept>
#include <string>
#include <thread>
#include <vector>

#include "../../include/kr_trace.h"

namespace krapp {

inline void check(int rc, const char* what)
{
    if (rc != KR_OK) throw std::runtime_error(std::string(what) + ": " + kr_last_error());
}

// KRTRACE_ARITHMETIC = hybrid | strict | fast, as in the host mirror of the class API; unset: hybrid for the
// fixed-step integrators, strict for RK45
inline int arithmetic_flags(const std::string& choice, int integrator)
{
    if (choice.empty()) return integrator == KR_RK45 ? 0 : KR_FLAG_HYBRID;
    if (choice == "hybrid") return KR_FLAG_HYBRID;
    if (choice == "strict") return 0;
    if (choice == "fast") return KR_FLAG_FAST_MATH;
    throw std::invalid_argument("arithmetic: expected hybrid, strict or fast, got '" + choice + "'");
}
inline std::string arithmetic_from_env()
{
    const char* e = std::getenv("KRTRACE_ARITHMETIC");
    return e ? std::string(e) : std::string();
}

inline int integrator_code(const std::string& name, int fallback)
{
    if (name == "euler") return KR_EULER;
    if (name == "rk4") return KR_RK4;
    if (name == "rk45") return KR_RK45;
    return fallback;
}

// device buffer that frees itself
class DeviceBuffer {
public:
    explicit DeviceBuffer(int64_t bytes) : bytes_(bytes) { check(kr_malloc(&ptr_, bytes), "kr_malloc"); }
    ~DeviceBuffer() { if (ptr_) kr_free(ptr_); }
    DeviceBuffer(const DeviceBuffer&) = delete;
    DeviceBuffer& operator=(const DeviceBuffer&) = delete;
    void* get() const { return ptr_; }
    void zero() { check(kr_memset(ptr_, 0, bytes_), "kr_memset"); }

private:
    void* ptr_ = nullptr;
    int64_t bytes_;
};

// page-locked host buffer of doubles (read-back target)
class PinnedDoubles {
public:
    explicit PinnedDoubles(int64_t count) : count_(count)
    {
        void* p = nullptr;
        check(kr_host_alloc(&p, count * (int64_t) sizeof(double)), "kr_host_alloc");
        ptr_ = static_cast<double*>(p);
    }
    ~PinnedDoubles() { if (ptr_) kr_host_free(ptr_); }
    PinnedDoubles(const PinnedDoubles&) = delete;
    PinnedDoubles& operator=(const PinnedDoubles&) = delete;
    double* data() const { return ptr_; }
    int64_t size() const { return count_; }
    double& operator[](int64_t i) const { return ptr_[i]; }

private:
    double* ptr_ = nullptr;
    int64_t count_;
};

class Stopwatch {
public:
    Stopwatch() : t0_(std::chrono::steady_clock::now()) {}
    double lap_ms()
    {
        const auto t1 = std::chrono::steady_clock::now();
        const double ms = std::chrono::duration<double, std::milli>(t1 - t0_).count();
        t0_ = t1;
        return ms;
    }

private:
    std::chrono::steady_clock::time_point t0_;
};

// fn(begin, end) over [0, n) in contiguous pieces on min(hardware threads, 16, KRTRACE_HOST_THREADS) threads (the calling thread takes the first)
template <class F>
inline void parallel_for(int64_t n, F fn)
{
    unsigned hw = std::thread::hardware_concurrency();
    int64_t t = hw == 0 ? 4 : (hw < 16 ? hw : 16);
    if (const char* e = getenv("KRTRACE_HOST_THREADS")) {
        const int v = atoi(e);
        if (v >= 1 && v < t) t = v;
    }
    if (n < (int64_t) 1 << 16) t = 1;
    if (t <= 1) { fn((int64_t) 0, n); return; }
    std::vector<std::thread> others;
    others.reserve((size_t) t - 1);
    for (int64_t i = 1; i < t; ++i) others.emplace_back(fn, n * i / t, n * (i + 1) / t);
    fn((int64_t) 0, n / t);
    for (std::thread& th : others) th.join();
}

}   // namespace krapp

#endif /* KR_APP_COMMON_H_ */
