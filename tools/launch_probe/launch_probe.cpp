// Host-side cost of getting a chain of N dependent kernels onto an MI355X and knowing that it finished (tools/launch_probe):
//   graph   hipGraphLaunch of a captured chain + hipStreamQuery polling (what rc_live_step does)
//   direct  N hipLaunchKernelGGL + polling
//   aql     the same chain as N AQL packets written by this process into its own HSA queue (barrier bit on every packet, completion
//           signal on the last), polled with hsa_signal_load
// Every link spins `ticks` x 10 ns on the device, so: overhead = wall - N * link time.
#include <hip/hip_runtime.h>
#include <hsa/hsa.h>
#include <hsa/hsa_ext_amd.h>
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <vector>

extern "C" __global__ void probe_link(unsigned long long* out, int ticks, int seq);
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); std::exit(1); } } while (0)
#define HK(x) do { hsa_status_t s_ = (x); if (s_ != HSA_STATUS_SUCCESS) { const char* m_ = nullptr; hsa_status_string(s_, &m_); std::fprintf(stderr, "%s: %s\n", #x, m_ ? m_ : "?"); std::exit(1); } } while (0)

static double now_us() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
static void report(const char* name, std::vector<double>& v, int N, double link_us) {
    std::sort(v.begin(), v.end());
    const double p50 = v[v.size() / 2], p99 = v[(size_t)(v.size() * 0.99)];
    std::printf("%-8s N=%d link=%.1f us: p50 %.1f us  p99 %.1f us  -> overhead p50 %.1f us\n", name, N, link_us, p50, p99, p50 - N * link_us);
}

struct Agents { hsa_agent_t gpu{}; hsa_agent_t cpu{}; bool have_gpu = false, have_cpu = false; };
static hsa_status_t agent_cb(hsa_agent_t a, void* d) {
    Agents* g = (Agents*)d; hsa_device_type_t t;
    hsa_agent_get_info(a, HSA_AGENT_INFO_DEVICE, &t);
    if (t == HSA_DEVICE_TYPE_GPU && !g->have_gpu) { g->gpu = a; g->have_gpu = true; }
    if (t == HSA_DEVICE_TYPE_CPU && !g->have_cpu) { g->cpu = a; g->have_cpu = true; }
    return HSA_STATUS_SUCCESS;
}
struct Pools { hsa_amd_memory_pool_t kernarg{}; bool have = false; };
static hsa_status_t pool_cb(hsa_amd_memory_pool_t p, void* d) {
    Pools* q = (Pools*)d; hsa_amd_segment_t seg; uint32_t fl = 0;
    hsa_amd_memory_pool_get_info(p, HSA_AMD_MEMORY_POOL_INFO_SEGMENT, &seg);
    if (seg != HSA_AMD_SEGMENT_GLOBAL) return HSA_STATUS_SUCCESS;
    hsa_amd_memory_pool_get_info(p, HSA_AMD_MEMORY_POOL_INFO_GLOBAL_FLAGS, &fl);
    if ((fl & HSA_AMD_MEMORY_POOL_GLOBAL_FLAG_KERNARG_INIT) && !q->have) { q->kernarg = p; q->have = true; }
    return HSA_STATUS_SUCCESS;
}

int main(int argc, char** argv) {
    const int N = argc > 1 ? std::atoi(argv[1]) : 7, ticks = argc > 2 ? std::atoi(argv[2]) : 500, iters = argc > 3 ? std::atoi(argv[3]) : 2000;
    const char* hsaco = argc > 4 ? argv[4] : "tools/launch_probe/probe_kernels.hsaco";
    const double link_us = ticks * 0.01;
    unsigned long long* out_d; CK(hipMalloc(&out_d, 64));
    hipStream_t st; CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    // ---- graph
    hipGraph_t g; hipGraphExec_t ge;
    CK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
    for (int i = 0; i < N; ++i) hipLaunchKernelGGL(probe_link, dim3(64), dim3(256), 0, st, out_d, ticks, i);
    CK(hipStreamEndCapture(st, &g)); CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    std::vector<double> v;
    for (int it = 0; it < iters; ++it) {
        const double t0 = now_us();
        CK(hipGraphLaunch(ge, st));
        while (hipStreamQuery(st) == hipErrorNotReady) { }
        v.push_back(now_us() - t0);
    }
    report("graph", v, N, link_us);
    v.clear();
    for (int it = 0; it < iters; ++it) {
        const double t0 = now_us();
        for (int i = 0; i < N; ++i) hipLaunchKernelGGL(probe_link, dim3(64), dim3(256), 0, st, out_d, ticks, i);
        while (hipStreamQuery(st) == hipErrorNotReady) { }
        v.push_back(now_us() - t0);
    }
    report("direct", v, N, link_us);
    // ---- AQL on a queue of our own
    HK(hsa_init());
    Agents ag; HK(hsa_iterate_agents(agent_cb, &ag));
    hsa_queue_t* q = nullptr;
    HK(hsa_queue_create(ag.gpu, 256, HSA_QUEUE_TYPE_SINGLE, nullptr, nullptr, UINT32_MAX, UINT32_MAX, &q));
    std::ifstream f(hsaco, std::ios::binary); std::vector<char> blob((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
    if (blob.empty()) { std::fprintf(stderr, "cannot read %s\n", hsaco); return 1; }
    hsa_code_object_reader_t rd; HK(hsa_code_object_reader_create_from_memory(blob.data(), blob.size(), &rd));
    hsa_executable_t ex; HK(hsa_executable_create_alt(HSA_PROFILE_FULL, HSA_DEFAULT_FLOAT_ROUNDING_MODE_DEFAULT, nullptr, &ex));
    HK(hsa_executable_load_agent_code_object(ex, ag.gpu, rd, nullptr, nullptr));
    HK(hsa_executable_freeze(ex, nullptr));
    hsa_executable_symbol_t sym; HK(hsa_executable_get_symbol_by_name(ex, "probe_link.kd", &ag.gpu, &sym));
    uint64_t kobj = 0; uint32_t kseg = 0, gseg = 0, pseg = 0;
    HK(hsa_executable_symbol_get_info(sym, HSA_EXECUTABLE_SYMBOL_INFO_KERNEL_OBJECT, &kobj));
    HK(hsa_executable_symbol_get_info(sym, HSA_EXECUTABLE_SYMBOL_INFO_KERNEL_KERNARG_SEGMENT_SIZE, &kseg));
    HK(hsa_executable_symbol_get_info(sym, HSA_EXECUTABLE_SYMBOL_INFO_KERNEL_GROUP_SEGMENT_SIZE, &gseg));
    HK(hsa_executable_symbol_get_info(sym, HSA_EXECUTABLE_SYMBOL_INFO_KERNEL_PRIVATE_SEGMENT_SIZE, &pseg));
    std::printf("aql: kernarg %u B, group %u B, private %u B\n", kseg, gseg, pseg);
    Pools pl; HK(hsa_amd_agent_iterate_memory_pools(ag.cpu, pool_cb, &pl));
    char* kargs = nullptr; HK(hsa_amd_memory_pool_allocate(pl.kernarg, (size_t)N * 256, 0, (void**)&kargs));
    HK(hsa_amd_agents_allow_access(1, &ag.gpu, nullptr, kargs));
    for (int i = 0; i < N; ++i) {
        char* k = kargs + (size_t)i * 256; std::memset(k, 0, 256);
        std::memcpy(k, &out_d, 8); std::memcpy(k + 8, &ticks, 4); std::memcpy(k + 12, &i, 4);
        // code object v5 hidden arguments behind the explicit ones (8-byte aligned): block counts (3 x u32), group sizes (3 x u16)
        const uint32_t bc[3] = {64, 1, 1}; const uint16_t gs[3] = {256, 1, 1};
        std::memcpy(k + 16, bc, 12); std::memcpy(k + 28, gs, 6);
    }
    hsa_signal_t done; HK(hsa_signal_create(1, 0, nullptr, &done));
    const uint32_t mask = q->size - 1;
  for (int scopes = 0; scopes < 2; ++scopes) {
    v.clear();
    for (int it = 0; it < iters; ++it) {
        const double t0 = now_us();
        hsa_signal_store_relaxed(done, 1);
        const uint64_t base = hsa_queue_add_write_index_relaxed(q, N);
        for (int i = 0; i < N; ++i) {
            hsa_kernel_dispatch_packet_t* p = (hsa_kernel_dispatch_packet_t*)q->base_address + ((base + i) & mask);
            p->setup = 1 << HSA_KERNEL_DISPATCH_PACKET_SETUP_DIMENSIONS;
            p->workgroup_size_x = 256; p->workgroup_size_y = 1; p->workgroup_size_z = 1;
            p->grid_size_x = 64 * 256; p->grid_size_y = 1; p->grid_size_z = 1;
            p->private_segment_size = pseg; p->group_segment_size = gseg;
            p->kernel_object = kobj; p->kernarg_address = kargs + (size_t)i * 256;
            p->completion_signal = i == N - 1 ? done : hsa_signal_t{0};
            // fences: what the chain needs and no more -- agent scope between the links (same device), system scope only where the host
            // is on the other side (acquire of the first packet, release of the last); `scopes` = 1 puts system scope everywhere
            const int acq = (scopes == 1 || i == 0) ? HSA_FENCE_SCOPE_SYSTEM : HSA_FENCE_SCOPE_AGENT;
            const int rel = (scopes == 1 || i == N - 1) ? HSA_FENCE_SCOPE_SYSTEM : HSA_FENCE_SCOPE_AGENT;
            const uint16_t hdr = (HSA_PACKET_TYPE_KERNEL_DISPATCH << HSA_PACKET_HEADER_TYPE) | (1 << HSA_PACKET_HEADER_BARRIER) |
                                 (acq << HSA_PACKET_HEADER_SCACQUIRE_FENCE_SCOPE) | (rel << HSA_PACKET_HEADER_SCRELEASE_FENCE_SCOPE);
            __atomic_store_n((uint16_t*)p, hdr, __ATOMIC_RELEASE);
        }
        hsa_signal_store_screlease(q->doorbell_signal, base + N - 1);
        while (hsa_signal_load_scacquire(done) != 0) { }
        v.push_back(now_us() - t0);
    }
    report(scopes ? "aql-sys" : "aql", v, N, link_us);
  }
    unsigned long long h = 0; CK(hipMemcpy(&h, out_d, 8, hipMemcpyDeviceToHost));
    std::printf("last link wrote %llu (expect %d)\n", h, N - 1);
    return 0;
}
