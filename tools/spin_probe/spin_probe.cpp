// Micro-probe (gfx950): how fast does a kernel that is ALREADY RUNNING notice a word the host writes into device memory (large BAR)?
// The question behind it: a live frame pays ~6.8 us between rc_live_step's call and K1's first instruction even with the queue armed
// (profiles/r05_live_host_device_split.txt). Would a K1 launched ahead of the frame and spinning on a mailbox start sooner?
//   a) one workgroup spins on a mailbox word in host-writable device memory (extended-scope fine-grained pool), then stores a pinned host flag
//   b) 112 workgroups: workgroup 0's lane 0 is the arbiter (polls the mailbox, publishes a decision word), all others poll the decision;
//      the last one to see it (device counter) stores the flag -- the shape a spinning rc_live_k1 would have
//   c) for scale: the same 112-workgroup kernel launched when the "frame arrives" (hipLaunchKernelGGL, no spin)
// Every spin is bounded (20 ms of the 100 MHz counter): the probe cannot hang.
//   hipcc --offload-arch=gfx950 -O2 -o spin_probe tools/spin_probe/spin_probe.cpp -lhsa-runtime64 && ./spin_probe
#include <hip/hip_runtime.h>
#include <hsa/hsa.h>
#include <hsa/hsa_ext_amd.h>
#include <immintrin.h>
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <vector>

__device__ __forceinline__ unsigned ld_sys(const unsigned* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); }

// mail[0]: written by the host; mail[16]: decision (written by the arbiter); mail[32]: arrival counter
extern "C" __global__ void k_spin(unsigned* mail, unsigned expect, unsigned* done, int arbiter) {
    const unsigned long long t0 = wall_clock64();
    __shared__ unsigned s_go;
    if (threadIdx.x == 0) {
        unsigned v = 0;
        if (!arbiter || blockIdx.x == 0) {
            while ((v = ld_sys(mail)) != expect && wall_clock64() - t0 < 2000000ull) __builtin_amdgcn_s_sleep(1);
            if (arbiter) __hip_atomic_store(mail + 16, v, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        } else {
            while ((v = ld_sys(mail + 16)) != expect && wall_clock64() - t0 < 2000000ull) __builtin_amdgcn_s_sleep(1);
        }
        s_go = v;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        const unsigned n = atomicAdd(mail + 32, 1u);
        if (n == gridDim.x - 1) {
            mail[32] = 0;
            __threadfence_system();
            __hip_atomic_store(done, s_go == expect ? expect : 0xdeadu, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
}
extern "C" __global__ void k_plain(unsigned* mail, unsigned expect, unsigned* done) {
    if (threadIdx.x == 0) {
        const unsigned n = atomicAdd(mail + 32, 1u);
        if (n == gridDim.x - 1) {
            mail[32] = 0;
            __threadfence_system();
            __hip_atomic_store(done, expect, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
}

struct Find { hsa_agent_t gpu{}, cpu{}; bool have_gpu = false, have_cpu = false; };
static hsa_status_t agent_cb(hsa_agent_t a, void* d) {
    Find* f = (Find*)d;
    hsa_device_type_t t;
    hsa_agent_get_info(a, HSA_AGENT_INFO_DEVICE, &t);
    if (t == HSA_DEVICE_TYPE_GPU && !f->have_gpu) { f->gpu = a; f->have_gpu = true; }
    if (t == HSA_DEVICE_TYPE_CPU && !f->have_cpu) { f->cpu = a; f->have_cpu = true; }
    return HSA_STATUS_SUCCESS;
}
struct Pools { std::vector<hsa_amd_memory_pool_t> v; };
static hsa_status_t pool_cb(hsa_amd_memory_pool_t p, void* d) { ((Pools*)d)->v.push_back(p); return HSA_STATUS_SUCCESS; }

static double now_us() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

int main() {
    hipSetDevice(0);
    float* warm; hipMalloc(&warm, 4);
    hsa_init();
    Find f; hsa_iterate_agents(agent_cb, &f);
    Pools ps; hsa_amd_agent_iterate_memory_pools(f.gpu, pool_cb, &ps);
    unsigned* done = nullptr; unsigned* done_d = nullptr;
    hipHostMalloc((void**)&done, 64, hipHostMallocMapped); hipHostGetDevicePointer((void**)&done_d, done, 0);
    hipStream_t s; hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
    for (hsa_amd_memory_pool_t p : ps.v) {
        hsa_amd_segment_t seg; uint32_t flags = 0; bool alloc = false;
        hsa_amd_memory_pool_get_info(p, HSA_AMD_MEMORY_POOL_INFO_SEGMENT, &seg);
        if (seg != HSA_AMD_SEGMENT_GLOBAL) continue;
        hsa_amd_memory_pool_get_info(p, HSA_AMD_MEMORY_POOL_INFO_GLOBAL_FLAGS, &flags);
        hsa_amd_memory_pool_get_info(p, HSA_AMD_MEMORY_POOL_INFO_RUNTIME_ALLOC_ALLOWED, &alloc);
        hsa_amd_memory_pool_access_t acc;
        hsa_amd_agent_memory_pool_get_info(f.cpu, p, HSA_AMD_AGENT_MEMORY_POOL_INFO_ACCESS, &acc);
        if (!alloc || acc == HSA_AMD_MEMORY_POOL_ACCESS_NEVER_ALLOWED) continue;
        void* ptr = nullptr;
        if (hsa_amd_memory_pool_allocate(p, 4096, 0, &ptr) != HSA_STATUS_SUCCESS) continue;
        hsa_agent_t both[2] = {f.cpu, f.gpu};
        if (hsa_amd_agents_allow_access(2, both, nullptr, ptr) != HSA_STATUS_SUCCESS) { hsa_amd_memory_pool_free(ptr); continue; }
        volatile unsigned* mail = (volatile unsigned*)ptr;
        for (int i = 0; i < 64; ++i) mail[i] = 0;
        _mm_sfence();
        hipDeviceSynchronize();
        std::printf("device pool flags 0x%x (fine %d coarse %d ext-fine %d)\n", flags, !!(flags & 2), !!(flags & 4), !!(flags & 8));
        for (int mode = 0; mode < 3; ++mode) {
            const int wg = mode == 0 ? 1 : 112;
            std::vector<double> rt;
            int bad = 0;
            for (unsigned it = 1; it <= 300; ++it) {
                *done = 0;
                if (mode < 2) {
                    hipLaunchKernelGGL(k_spin, dim3(wg), dim3(256), 0, s, (unsigned*)ptr, it, done_d, mode == 1 ? 1 : 0);
                    const double tw = now_us() + (it % 3 == 0 ? 2000.0 : 200.0);     // the kernel is surely spinning (a third of them after 2 ms)
                    while (now_us() < tw) {}
                }
                const double t0 = now_us();
                if (mode < 2) { mail[0] = it; _mm_sfence(); }
                else hipLaunchKernelGGL(k_plain, dim3(wg), dim3(256), 0, s, (unsigned*)ptr, it, done_d);
                unsigned v = 0;
                while ((v = __atomic_load_n(done, __ATOMIC_ACQUIRE)) == 0 && now_us() - t0 < 50000.0) {}
                rt.push_back(now_us() - t0);
                if (v != it) ++bad;
                hipStreamSynchronize(s);
            }
            std::sort(rt.begin(), rt.end());
            std::printf("  %-64s round trip p50 %6.2f  p90 %6.2f  p99 %6.2f us   wrong / timed out %d of 300\n",
                        mode == 0 ? "a) 1 spinning workgroup: host write -> pinned flag" : (mode == 1 ? "b) 112 spinning workgroups, arbiter + decision word" : "c) 112 workgroups launched at the arrival (hipLaunchKernelGGL)"),
                        rt[rt.size() / 2], rt[rt.size() * 9 / 10], rt[rt.size() * 99 / 100], bad);
        }
        hsa_amd_memory_pool_free(ptr);
    }
    return 0;
}
