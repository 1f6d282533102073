// kernels of tools/launch_probe: a chain link that holds the GPU for a given number of 100 MHz ticks and bumps a counter
#include <hip/hip_runtime.h>
extern "C" __global__ __launch_bounds__(256) void probe_link(unsigned long long* out, int ticks, int seq) {
    const unsigned long long t0 = wall_clock64();
    while ((long long)(wall_clock64() - t0) < ticks) { }
    if (threadIdx.x == 0 && blockIdx.x == 0) out[0] = (unsigned long long)seq;
}
