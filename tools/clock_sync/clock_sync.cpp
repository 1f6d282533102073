// Host <-> device clock offset for tools/live_trace.py: a one-wave kernel keeps storing wall_clock64() (the constant 100 MHz counter the
// RC_LIVE_TRACE stamps use) into pinned host memory; the host pairs what it reads there with CLOCK_MONOTONIC. The freshest pair has the
// smallest host - device difference: offset = min(host_ns - ticks * ns_per_tick), which still contains one posted write over PCIe (~0.5-1 us).
//   hipcc --offload-arch=gfx950 -O2 -fPIC -shared -o tools/clock_sync/libclock_sync.so tools/clock_sync/clock_sync.cpp
#include <hip/hip_runtime.h>
#include <time.h>

extern "C" __global__ void k_clock(volatile unsigned long long* out, volatile int* stop, int iters) {
    for (int i = 0; i < iters; ++i) {
        out[0] = wall_clock64();
        __threadfence_system();
        if (*stop) break;
    }
}

static double now_ns() {
    timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return (double)ts.tv_sec * 1e9 + (double)ts.tv_nsec;
}

// out[0] = offset_ns (host_ns = ticks * out[1] + offset_ns), out[1] = ns per tick, out[2] = samples that moved
extern "C" int clock_sync(double* out) {
    int rate_khz = 0;
    if (hipDeviceGetAttribute(&rate_khz, hipDeviceAttributeWallClockRate, 0) != hipSuccess || rate_khz <= 0) return -1;
    const double ns_per_tick = 1e6 / (double)rate_khz;
    unsigned long long* w = nullptr;
    int* stop = nullptr;
    if (hipHostMalloc((void**)&w, 64, hipHostMallocMapped) != hipSuccess || hipHostMalloc((void**)&stop, 64, hipHostMallocMapped) != hipSuccess) return -2;
    *w = 0; *stop = 0;
    hipStream_t s;
    if (hipStreamCreateWithFlags(&s, hipStreamNonBlocking) != hipSuccess) return -3;
    hipLaunchKernelGGL(k_clock, dim3(1), dim3(1), 0, s, (volatile unsigned long long*)w, (volatile int*)stop, 400000);   // bounded: ends by itself
    double best = 1e300;
    int moved = 0;
    unsigned long long last = 0;
    const double t_end = now_ns() + 50e6;                                    // 50 ms of samples
    while (now_ns() < t_end) {
        const unsigned long long v = __atomic_load_n(w, __ATOMIC_ACQUIRE);
        const double t = now_ns();
        if (v != 0 && v != last) {
            last = v;
            ++moved;
            const double d = t - (double)v * ns_per_tick;
            if (d < best) best = d;
        }
    }
    __atomic_store_n(stop, 1, __ATOMIC_RELEASE);
    hipStreamSynchronize(s);
    hipStreamDestroy(s);
    hipHostFree(w);
    hipHostFree(stop);
    if (moved < 100) return -4;
    out[0] = best; out[1] = ns_per_tick; out[2] = (double)moved;
    return 0;
}
