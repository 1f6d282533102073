// The lean live frame as a chain of AQL packets on an HSA queue of the context's own (host code; the kernels are rc_live.hip's).
//
// hipGraphLaunch costs ~7 us of host time per replay on ROCm 7.2 and a further ~10 us between its return, the first kernel and the
// host noticing the end (tools/launch_probe: a chain of seven 5-us kernels -- graph 58 us, direct launches 53.5 us, this 50.8 us;
// one kernel: 23.9 / 20.7 / 16.2 us). A live frame is the same seven dispatches every time, with the same arguments (the inputs
// arrive at fixed pinned addresses): the packets are built ONCE; a frame copies them into the ring, rings the doorbell and polls the
// last packet's completion signal. Kernel objects are found by symbol among the code objects HIP has loaded (AMD loader
// extension), kernel arguments live in device memory. Fences: agent scope between the links of the chain, system scope where
// the host is on the other side (first acquire, last release).
#include "rc_internal.h"

#include <hsa/hsa.h>
#include <hsa/hsa_ext_amd.h>
#include <hsa/hsa_ven_amd_loader.h>

#if defined(__x86_64__) || defined(_M_X64)
#include <immintrin.h>
#define RC_STORE_FENCE() _mm_sfence()          // posted writes to the device's BAR leave the write-combining buffers in program order
#else
#define RC_STORE_FENCE() __atomic_thread_fence(__ATOMIC_SEQ_CST)
#endif

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

struct AqlProgram {
    hsa_kernel_dispatch_packet_t pkt[RC_LIVE_KERNELS]{};
    uint16_t hdr[RC_LIVE_KERNELS]{};
    int n = 0;
    bool frame = false;                          // a live frame (completion word / frame signal) or a background program (pre-step)
    char* kargs = nullptr;                       // device memory: n blocks of kKargStride bytes
};

struct AqlChain {
    hsa_agent_t gpu{};
    hsa_queue_t* q = nullptr;
    hsa_signal_t done{};
    hsa_signal_t bg_done{};                      // background programs: decremented by the last packet of each
    unsigned long long bg_seq = 0;
    // completion by a word the last kernel stores itself (one row; RC_LIVE_DONE_FLAG=0 switches it off): the sequence number of the frame
    unsigned* flag_h = nullptr;                  // pinned host word (the device sees the same address)
    unsigned* seq_d = nullptr;                   // device counter of frames
    unsigned long long seq = 0;                  // frames submitted (64 bits: the completion signal counts down once per frame for the life of
                                                 // the chain; the flag word K7 stores is its low 32 bits)
    // RC_LIVE_ARM: a barrier-AND packet left at the head of the ring while the caller is away, waiting on arm[armed]; the next push releases it
    hsa_signal_t arm[2]{};
    int arm_next = 0, armed = -1;
    hsa_agent_t cpu{};
    bool have_cpu = false;
    std::vector<void*> shared;                   // rc_aql_alloc_shared
    volatile unsigned* mailbox = nullptr;        // of a spinning K1 (rc_aql_set_mailbox)
    bool dead = false;                           // a frame did not complete in time: the chain takes no further frame (rc_live_step falls back)
    long long sig0 = 0;                          // value of `done` / `bg_done` before any program: every retired one decrements it
    std::vector<AqlProgram> prog;
    void* loader_fs = nullptr;
    bool hsa_up = false;
};

namespace {
constexpr size_t kKargStride = 2048;

struct FindAgent { uint32_t bdf; uint32_t domain = 0; int count = 0; hsa_agent_t first{}, match{}; bool have_match = false; hsa_agent_t cpu{}; bool have_cpu = false; };
hsa_status_t agent_cb(hsa_agent_t a, void* d) {
    FindAgent* f = (FindAgent*)d;
    hsa_device_type_t t;
    if (hsa_agent_get_info(a, HSA_AGENT_INFO_DEVICE, &t) != HSA_STATUS_SUCCESS) return HSA_STATUS_SUCCESS;
    if (t == HSA_DEVICE_TYPE_CPU && !f->have_cpu) { f->cpu = a; f->have_cpu = true; }
    if (t != HSA_DEVICE_TYPE_GPU) return HSA_STATUS_SUCCESS;
    if (f->count++ == 0) f->first = a;
    uint32_t bdf = 0, dom = 0;
    if (hsa_agent_get_info(a, (hsa_agent_info_t)HSA_AMD_AGENT_INFO_DOMAIN, &dom) != HSA_STATUS_SUCCESS) dom = f->domain;   // (older runtimes: bus / device / function only)
    if (hsa_agent_get_info(a, (hsa_agent_info_t)HSA_AMD_AGENT_INFO_BDFID, &bdf) == HSA_STATUS_SUCCESS && bdf == f->bdf && dom == f->domain &&
        !f->have_match) {
        f->match = a; f->have_match = true;
    }
    return HSA_STATUS_SUCCESS;
}

struct FindSyms { hsa_agent_t gpu; const LiveKernel* k; int n; uint64_t kobj[RC_LIVE_KERNELS]; uint32_t group[RC_LIVE_KERNELS], priv[RC_LIVE_KERNELS], karg[RC_LIVE_KERNELS]; };
hsa_status_t sym_cb(hsa_executable_t, hsa_agent_t, hsa_executable_symbol_t s, void* d) {
    FindSyms* f = (FindSyms*)d;
    hsa_symbol_kind_t kind;
    if (hsa_executable_symbol_get_info(s, HSA_EXECUTABLE_SYMBOL_INFO_TYPE, &kind) != HSA_STATUS_SUCCESS || kind != HSA_SYMBOL_KIND_KERNEL) return HSA_STATUS_SUCCESS;
    uint32_t len = 0;
    hsa_executable_symbol_get_info(s, HSA_EXECUTABLE_SYMBOL_INFO_NAME_LENGTH, &len);
    if (len == 0 || len > 200) return HSA_STATUS_SUCCESS;
    char name[256];
    hsa_executable_symbol_get_info(s, HSA_EXECUTABLE_SYMBOL_INFO_NAME, name);
    name[len] = 0;
    for (int i = 0; i < f->n; ++i) {
        const std::string want = std::string(f->k[i].name) + ".kd";
        if (want != name) continue;
        hsa_executable_symbol_get_info(s, HSA_EXECUTABLE_SYMBOL_INFO_KERNEL_OBJECT, &f->kobj[i]);
        hsa_executable_symbol_get_info(s, HSA_EXECUTABLE_SYMBOL_INFO_KERNEL_GROUP_SEGMENT_SIZE, &f->group[i]);
        hsa_executable_symbol_get_info(s, HSA_EXECUTABLE_SYMBOL_INFO_KERNEL_PRIVATE_SEGMENT_SIZE, &f->priv[i]);
        hsa_executable_symbol_get_info(s, HSA_EXECUTABLE_SYMBOL_INFO_KERNEL_KERNARG_SEGMENT_SIZE, &f->karg[i]);
    }
    return HSA_STATUS_SUCCESS;
}
hsa_status_t exec_cb(hsa_executable_t e, void* d) {
    FindSyms* f = (FindSyms*)d;
    (void)hsa_executable_iterate_agent_symbols(e, f->gpu, sym_cb, d);
    return HSA_STATUS_SUCCESS;
}

int bail(AqlChain* c, char* err, int err_len, const char* what, hsa_status_t st = HSA_STATUS_SUCCESS) {
    if (err && err_len > 0) {
        const char* m = nullptr;
        if (st != HSA_STATUS_SUCCESS) hsa_status_string(st, &m);
        std::snprintf(err, (size_t)err_len, "%s%s%s", what, m ? ": " : "", m ? m : "");
    }
    rc_aql_destroy(c);
    return -1;
}
}  // namespace

int rc_aql_create(int hip_device, AqlChain** out, char* err, int err_len) {
    if (!out) return -1;
    *out = nullptr;
    AqlChain* c = new AqlChain();
    hsa_status_t st = hsa_init();                                    // reference-counted: HIP runs on the same runtime
    if (st != HSA_STATUS_SUCCESS) return bail(c, err, err_len, "hsa_init", st);
    c->hsa_up = true;
    // the HSA agent of the HIP device: by PCI domain / bus / device / function (an 8-GPU node may repeat bus numbers across domains)
    FindAgent fa{};
    {
        char bus[64] = {0};
        unsigned dom = 0, b = 0, dv = 0, fn = 0;
        if (hipDeviceGetPCIBusId(bus, sizeof(bus), hip_device) == hipSuccess && std::sscanf(bus, "%x:%x:%x.%x", &dom, &b, &dv, &fn) == 4)
            { fa.bdf = (b << 8) | (dv << 3) | fn; fa.domain = dom; }
        else fa.bdf = 0xffffffffu;
    }
    if ((st = hsa_iterate_agents(agent_cb, &fa)) != HSA_STATUS_SUCCESS || fa.count == 0) return bail(c, err, err_len, "no GPU agent", st);
    if (!fa.have_match && fa.count != 1) return bail(c, err, err_len, "cannot match the HIP device to an HSA agent");
    c->gpu = fa.have_match ? fa.match : fa.first;
    c->cpu = fa.cpu; c->have_cpu = fa.have_cpu;
    if ((st = hsa_queue_create(c->gpu, 64, HSA_QUEUE_TYPE_SINGLE, nullptr, nullptr, UINT32_MAX, UINT32_MAX, &c->q)) != HSA_STATUS_SUCCESS)
        return bail(c, err, err_len, "hsa_queue_create", st);
    c->sig0 = 1ll << 62;                                                     // counts down once per frame: ~10^14 years of frames
    if ((st = hsa_signal_create(c->sig0, 0, nullptr, &c->done)) != HSA_STATUS_SUCCESS) return bail(c, err, err_len, "hsa_signal_create", st);
    if ((st = hsa_signal_create(c->sig0, 0, nullptr, &c->bg_done)) != HSA_STATUS_SUCCESS) return bail(c, err, err_len, "hsa_signal_create", st);
    for (int a = 0; a < 2; ++a)
        if ((st = hsa_signal_create(0, 0, nullptr, &c->arm[a])) != HSA_STATUS_SUCCESS) return bail(c, err, err_len, "hsa_signal_create", st);
    *out = c;
    return 0;
}

int rc_aql_add(AqlChain* c, const LiveKernel* k, int n, int frame, char* err, int err_len) {
    auto fail_add = [&](const char* what, hsa_status_t st = HSA_STATUS_SUCCESS) {
        if (err && err_len > 0) {
            const char* m = nullptr;
            if (st != HSA_STATUS_SUCCESS) hsa_status_string(st, &m);
            std::snprintf(err, (size_t)err_len, "%s%s%s", what, m ? ": " : "", m ? m : "");
        }
        return -1;
    };
    if (!c || !c->q || !k || n < 1 || n > RC_LIVE_KERNELS) return fail_add("bad program");
    hsa_status_t st;
    // kernel objects of the (already loaded) kernels
    hsa_ven_amd_loader_1_03_pfn_t ld{};
    if ((st = hsa_system_get_major_extension_table(HSA_EXTENSION_AMD_LOADER, 1, sizeof(ld), &ld)) != HSA_STATUS_SUCCESS ||
        !ld.hsa_ven_amd_loader_iterate_executables)
        return fail_add("AMD loader extension 1.03", st);
    FindSyms fs{};
    fs.gpu = c->gpu; fs.k = k; fs.n = n;
    if ((st = ld.hsa_ven_amd_loader_iterate_executables(exec_cb, &fs)) != HSA_STATUS_SUCCESS) return fail_add("iterate executables", st);
    for (int i = 0; i < n; ++i) {
        // arguments: (LiveFrame) | (LiveFrame, LiveGrid) | has_grid == 2: (LiveFrame, one pointer: LiveGrid.prebuf)
        const uint32_t need = (uint32_t)(sizeof(LiveFrame) + (k[i].has_grid == 1 ? sizeof(LiveGrid) : (k[i].has_grid == 2 ? sizeof(void*) : 0)));
        if (!fs.kobj[i]) return fail_add((std::string("kernel symbol not loaded: ") + k[i].name).c_str());
        // explicit arguments only: a kernel that grew hidden arguments (printf, dynamic LDS, blockDim) does not fit this path
        if (fs.karg[i] != need || fs.karg[i] > kKargStride) return fail_add((std::string("unexpected kernarg segment of ") + k[i].name).c_str());
    }
    // This path hands the packet processor kernel arguments in DEVICE memory that K1 / K4 rewrite while later packets of the frame are
    // already in the ring (LiveGrid.hot): that is only sound for kernels that read every argument word from memory when they run. A code
    // object built with kernarg preload (arguments copied into SGPRs when the packet is processed) would see stale words -- checked in the
    // kernel descriptors (bytes 58-59: preload length in the low 7 bits), not assumed.
    for (int i = 0; i < n; ++i) {
        unsigned char kd[64] = {0};
        if (hipMemcpy(kd, reinterpret_cast<const void*>(fs.kobj[i]), sizeof(kd), hipMemcpyDeviceToHost) != hipSuccess)
            return fail_add((std::string("cannot read the kernel descriptor of ") + k[i].name).c_str());
        const unsigned preload = (unsigned)kd[58] | ((unsigned)kd[59] << 8);
        uint32_t kd_karg = 0;
        std::memcpy(&kd_karg, kd + 8, 4);
        if ((preload & 0x7f) != 0) return fail_add((std::string("kernarg preload in ") + k[i].name + ": its arguments cannot be rewritten in place").c_str());
        if (kd_karg != fs.karg[i]) return fail_add((std::string("kernel descriptor / symbol mismatch of ") + k[i].name).c_str());
    }
    c->prog.emplace_back();
    AqlProgram& P = c->prog.back();
    auto drop = [&](const char* what) { if (P.kargs) (void)hipFree(P.kargs); c->prog.pop_back(); return fail_add(what); };
    P.n = n; P.frame = frame != 0;
    // kernel arguments: device memory, written once (a frame's inputs arrive at fixed pinned addresses)
    std::vector<char> host((size_t)n * kKargStride, 0);
    if (hipMalloc((void**)&P.kargs, host.size()) != hipSuccess) { P.kargs = nullptr; return drop("kernarg buffer"); }
    // The LSTM launches find their per-row words in their own argument block (LiveGrid.hot): the linear1 kernel in front of them
    // writes them there. Launch i's LiveGrid sits behind its LiveFrame; the plan is K1 | l0 l1 | K4 | l0 l1 | K7.
    LiveGrid* hot[4] = {nullptr, nullptr, nullptr, nullptr};
    if (P.frame && n == RC_LIVE_KERNELS) {
        const int lstm[4] = {1, 2, 4, 5};
        for (int q = 0; q < 4; ++q) hot[q] = reinterpret_cast<LiveGrid*>(P.kargs + (size_t)lstm[q] * kKargStride + sizeof(LiveFrame));
    }
    static const bool flag_env = [] { const char* e = std::getenv("RC_LIVE_DONE_FLAG"); return !e || std::atoi(e) != 0; }();
    if (P.frame && flag_env && n == RC_LIVE_KERNELS && k[0].F.B == 1 && !c->flag_h) {
        if (hipHostMalloc((void**)&c->flag_h, 64, hipHostMallocDefault) != hipSuccess) { c->flag_h = nullptr; return drop("completion word"); }
        *c->flag_h = 0;
        if (hipMalloc((void**)&c->seq_d, 64) != hipSuccess || hipMemset(c->seq_d, 0, 64) != hipSuccess) return drop("frame counter");
    }
    for (int i = 0; i < n; ++i) {
        LiveFrame F = k[i].F;
        for (int q = 0; q < 4; ++q) F.hot[q] = hot[q];
        F.done_flag = (P.frame && i == n - 1) ? c->flag_h : nullptr;
        F.done_seq = (P.frame && i == n - 1) ? c->seq_d : nullptr;
        std::memcpy(&host[(size_t)i * kKargStride], &F, sizeof(LiveFrame));
        if (k[i].has_grid == 1) {
            LiveGrid G = k[i].G;
            G.hot = hot[0] ? 1 : 0;
            std::memcpy(&host[(size_t)i * kKargStride + sizeof(LiveFrame)], &G, sizeof(LiveGrid));
        } else if (k[i].has_grid == 2) {
            const void* pbuf = k[i].G.prebuf;
            std::memcpy(&host[(size_t)i * kKargStride + sizeof(LiveFrame)], &pbuf, sizeof(void*));
        }
    }
    if (hipMemcpy(P.kargs, host.data(), host.size(), hipMemcpyHostToDevice) != hipSuccess) return drop("kernarg upload");
    for (int i = 0; i < n; ++i) {
        hsa_kernel_dispatch_packet_t& p = P.pkt[i];
        std::memset(&p, 0, sizeof(p));
        const unsigned wg = k[i].wg ? k[i].wg : 256u;
        p.setup = 1 << HSA_KERNEL_DISPATCH_PACKET_SETUP_DIMENSIONS;
        p.workgroup_size_x = (uint16_t)wg; p.workgroup_size_y = 1; p.workgroup_size_z = 1;
        p.grid_size_x = k[i].grid * wg; p.grid_size_y = 1; p.grid_size_z = 1;
        p.private_segment_size = fs.priv[i]; p.group_segment_size = fs.group[i];
        p.kernel_object = fs.kobj[i];
        p.kernarg_address = P.kargs + (size_t)i * kKargStride;
        p.completion_signal = i == n - 1 ? (P.frame ? c->done : c->bg_done) : hsa_signal_t{0};
        // (RC_AQL_EDGE_SCOPE=agent: probe of what the two system-scope fences cost; the frame's host-side I/O is fine-grained memory)
        static const bool edge_agent = [] { const char* e = std::getenv("RC_AQL_EDGE_SCOPE"); return e && !std::strcmp(e, "agent"); }();
        // a background program neither reads nor writes host memory: agent scope at both ends
        const int acq = (P.frame && i == 0 && !edge_agent) ? HSA_FENCE_SCOPE_SYSTEM : HSA_FENCE_SCOPE_AGENT;
        const int rel = (P.frame && i == n - 1 && !edge_agent) ? HSA_FENCE_SCOPE_SYSTEM : HSA_FENCE_SCOPE_AGENT;
        P.hdr[i] = (uint16_t)((HSA_PACKET_TYPE_KERNEL_DISPATCH << HSA_PACKET_HEADER_TYPE) | (1 << HSA_PACKET_HEADER_BARRIER) |
                              (acq << HSA_PACKET_HEADER_SCACQUIRE_FENCE_SCOPE) | (rel << HSA_PACKET_HEADER_SCRELEASE_FENCE_SCOPE));
    }
    return (int)c->prog.size() - 1;
}

static void aql_release(AqlChain* c) {
    if (c->armed >= 0) { hsa_signal_store_screlease(c->arm[c->armed], 0); c->armed = -1; }
}

// body first, header (which hands the packet to the packet processor) last; the barrier bit of every packet orders it behind everything
// in front of it in the ring -- a frame behind the pre-step that ran in the idle time, the pre-step behind the frame whose state it reads
static void aql_push(AqlChain* c, const AqlProgram& P, int i0 = 0, int i1 = -1, const bool first_beside = false) {
    hsa_queue_t* q = c->q;
    if (i1 < 0) i1 = P.n;
    const uint32_t mask = q->size - 1;
    const uint64_t base = hsa_queue_load_write_index_relaxed(q);           // single producer
    // The ring may hold an armed barrier, a pre-step (1-2 packets), a dismissed queued-ahead frame (7), the plain frame (7), a frame queued
    // ahead for a back-to-back caller (7) and a fence: ~25 of its 64 slots. Never write over a packet the processor has not consumed: wait for
    // the read index (a stalled packet processor then shows as a frame that does not complete -- rc_live_step's timeout -- not as corruption).
    while (base + (uint64_t)(i1 - i0) - hsa_queue_load_read_index_scacquire(q) > (uint64_t)q->size) { }
    for (int i = i0; i < i1; ++i) {
        hsa_kernel_dispatch_packet_t* p = (hsa_kernel_dispatch_packet_t*)q->base_address + ((base + (i - i0)) & mask);
        std::memcpy((char*)p + 4, (const char*)&P.pkt[i] + 4, sizeof(*p) - 4);
        // (first_beside: the first packet without the barrier bit -- it may start beside what is still running in front of it)
        const uint32_t hdr = (first_beside && i == i0) ? ((uint32_t)P.hdr[i] & ~(1u << HSA_PACKET_HEADER_BARRIER)) : (uint32_t)P.hdr[i];
        __atomic_store_n((uint32_t*)p, hdr | ((uint32_t)P.pkt[i].setup << 16), __ATOMIC_RELEASE);
    }
    hsa_queue_store_write_index_release(q, base + (i1 - i0));
    hsa_signal_store_screlease(q->doorbell_signal, (hsa_signal_value_t)(base + (i1 - i0) - 1));
    aql_release(c);
}

// Leave the packet processor waiting AT this queue while the caller is away (a paced caller: 16 ms between two frames): a barrier-AND
// packet on a signal the next push sets to zero once its packets are in the ring. Two signals in turn: the barrier that waited on the
// other one has retired by the time it is armed again (a frame has completed in between).
int rc_aql_arm(AqlChain* c) {
    if (!c || !c->q || c->dead || c->armed >= 0) return 0;
    const int a = c->arm_next;
    c->arm_next ^= 1;
    hsa_signal_store_screlease(c->arm[a], 1);
    hsa_queue_t* q = c->q;
    const uint32_t mask = q->size - 1;
    const uint64_t base = hsa_queue_load_write_index_relaxed(q);
    hsa_barrier_and_packet_t* p = (hsa_barrier_and_packet_t*)q->base_address + (base & mask);
    hsa_barrier_and_packet_t b;
    std::memset(&b, 0, sizeof(b));
    b.dep_signal[0] = c->arm[a];
    std::memcpy((char*)p + 4, (const char*)&b + 4, sizeof(b) - 4);
    const uint16_t hdr = (uint16_t)((HSA_PACKET_TYPE_BARRIER_AND << HSA_PACKET_HEADER_TYPE) | (1 << HSA_PACKET_HEADER_BARRIER));
    __atomic_store_n((uint32_t*)p, (uint32_t)hdr, __ATOMIC_RELEASE);
    hsa_queue_store_write_index_release(q, base + 1);
    hsa_signal_store_screlease(q->doorbell_signal, (hsa_signal_value_t)base);
    c->armed = a;
    return 0;
}

int rc_aql_submit(AqlChain* c, int prog) {
    if (!c || !c->q || prog < 0 || prog >= (int)c->prog.size() || c->prog[prog].frame) return -1;
    if (c->dead) return -3;
    aql_push(c, c->prog[prog]);
    c->bg_seq += 1;
    return 0;
}

// ---- a frame program queued AHEAD of its frame: its first kernel (rc_live_k1) waits on the device for the mailbox, the six behind it for
// the first; when the frame arrives the host only writes the inputs and the command word, then waits like for any frame
// beside != 0: the first kernel may start while the frame in front of it is still running (a back-to-back caller: it waits for ITS frame anyway,
// and then it is already there) -- every other packet keeps its barrier bit, i.e. waits for everything in front of it including that kernel
int rc_aql_submit_ahead(AqlChain* c, int prog, int beside) {
    if (!c || !c->q || prog < 0 || prog >= (int)c->prog.size() || !c->prog[prog].frame) return -1;
    if (c->dead) return -3;
    aql_push(c, c->prog[prog], 0, -1, beside != 0);
    c->seq += 1;
    return 0;
}
unsigned long long rc_aql_seq(const AqlChain* c) { return c ? c->seq : 0ull; }   // number of the frame submitted last

static int aql_wait_frame(AqlChain* c, unsigned long long seq);
int rc_aql_wait_seq(AqlChain* c, unsigned long long seq) {                  // frame number `seq` has retired (frames retire in order; none behind it has STARTED yet)
    if (!c || !c->q) return -1;
    if (c->dead) return -3;
    return aql_wait_frame(c, seq);
}
int rc_aql_wait_frame(AqlChain* c) { return rc_aql_wait_seq(c, c ? c->seq : 0ull); }

// everything in front of this packet has retired when the background signal has counted it (its barrier bit orders it)
int rc_aql_fence_background(AqlChain* c) {
    if (!c || !c->q) return -1;
    if (c->dead) return -3;
    hsa_queue_t* q = c->q;
    const uint32_t mask = q->size - 1;
    const uint64_t base = hsa_queue_load_write_index_relaxed(q);
    hsa_barrier_and_packet_t* p = (hsa_barrier_and_packet_t*)q->base_address + (base & mask);
    hsa_barrier_and_packet_t b;
    std::memset(&b, 0, sizeof(b));
    b.completion_signal = c->bg_done;
    std::memcpy((char*)p + 4, (const char*)&b + 4, sizeof(b) - 4);
    const uint16_t hdr = (uint16_t)((HSA_PACKET_TYPE_BARRIER_AND << HSA_PACKET_HEADER_TYPE) | (1 << HSA_PACKET_HEADER_BARRIER));
    __atomic_store_n((uint32_t*)p, (uint32_t)hdr, __ATOMIC_RELEASE);
    hsa_queue_store_write_index_release(q, base + 1);
    hsa_signal_store_screlease(q->doorbell_signal, (hsa_signal_value_t)base);
    aql_release(c);
    c->bg_seq += 1;
    return 0;
}

// Device memory the host writes directly (large BAR): the extended-scope fine-grained pool of the GPU (a kernel that is already running
// sees the host's stores: tools/spin_probe), made accessible to the CPU agent. Freed with the chain.
namespace {
struct PoolPick { hsa_amd_memory_pool_t pool{}; bool have = false; hsa_agent_t cpu; };
hsa_status_t pool_cb(hsa_amd_memory_pool_t p, void* d) {
    PoolPick* k = (PoolPick*)d;
    hsa_amd_segment_t seg;
    uint32_t flags = 0;
    bool alloc = false;
    if (hsa_amd_memory_pool_get_info(p, HSA_AMD_MEMORY_POOL_INFO_SEGMENT, &seg) != HSA_STATUS_SUCCESS || seg != HSA_AMD_SEGMENT_GLOBAL) return HSA_STATUS_SUCCESS;
    (void)hsa_amd_memory_pool_get_info(p, HSA_AMD_MEMORY_POOL_INFO_GLOBAL_FLAGS, &flags);
    (void)hsa_amd_memory_pool_get_info(p, HSA_AMD_MEMORY_POOL_INFO_RUNTIME_ALLOC_ALLOWED, &alloc);
    hsa_amd_memory_pool_access_t acc = HSA_AMD_MEMORY_POOL_ACCESS_NEVER_ALLOWED;
    (void)hsa_amd_agent_memory_pool_get_info(k->cpu, p, HSA_AMD_AGENT_MEMORY_POOL_INFO_ACCESS, &acc);
    if (!alloc || acc == HSA_AMD_MEMORY_POOL_ACCESS_NEVER_ALLOWED || !(flags & 8u)) return HSA_STATUS_SUCCESS;   // 8: extended-scope fine-grained
    if (!k->have) { k->pool = p; k->have = true; }
    return HSA_STATUS_SUCCESS;
}
}  // namespace

int rc_aql_alloc_shared(AqlChain* c, size_t bytes, void** ptr) {
    if (!c || !ptr || !c->have_cpu) return -1;
    *ptr = nullptr;
    PoolPick k;
    k.cpu = c->cpu;
    if (hsa_amd_agent_iterate_memory_pools(c->gpu, pool_cb, &k) != HSA_STATUS_SUCCESS || !k.have) return -2;
    void* q = nullptr;
    if (hsa_amd_memory_pool_allocate(k.pool, (bytes + 4095) & ~(size_t)4095, 0, &q) != HSA_STATUS_SUCCESS) return -3;
    hsa_agent_t both[2] = {c->cpu, c->gpu};
    if (hsa_amd_agents_allow_access(2, both, nullptr, q) != HSA_STATUS_SUCCESS) { (void)hsa_amd_memory_pool_free(q); return -4; }
    c->shared.push_back(q);
    *ptr = q;
    return 0;
}

void rc_aql_set_mailbox(AqlChain* c, volatile unsigned* mb) { if (c) c->mailbox = mb; }

int rc_aql_wait_background(AqlChain* c) {
    if (!c || !c->q || c->bg_seq == 0) return 0;
    const hsa_signal_value_t retired = (hsa_signal_value_t)(c->sig0 - (long long)c->bg_seq);
    if (hsa_signal_load_scacquire(c->bg_done) <= retired) return 0;
    return hsa_signal_wait_scacquire(c->bg_done, HSA_SIGNAL_CONDITION_LT, retired + 1, 2000000000ull, HSA_WAIT_STATE_BLOCKED) <= retired ? 0 : -2;
}

static int aql_wait_frame(AqlChain* c, const unsigned long long seq) {
    const unsigned seq32 = (unsigned)seq;                                   // what K7 stores: its device counter wraps the same way
    const hsa_signal_value_t retired = (hsa_signal_value_t)(c->sig0 - (long long)seq);
    const auto t0 = std::chrono::steady_clock::now();
    int spins = 0;
    for (;;) {
        if (c->flag_h ? (__atomic_load_n(c->flag_h, __ATOMIC_ACQUIRE) == seq32) : (hsa_signal_load_scacquire(c->done) == retired)) break;
        if ((++spins & 1023) == 0 && std::chrono::steady_clock::now() - t0 > std::chrono::milliseconds(2)) {
            // a frame is ~100 us: something is slow (profiler, contention) -- sleep on the signal, give up after 10 s
            // ("< retired + 1", not "== retired": the signal only counts down, and a wait that starts late must not miss the value)
            if (hsa_signal_wait_scacquire(c->done, HSA_SIGNAL_CONDITION_LT, retired + 1, 10000000000ull, HSA_WAIT_STATE_BLOCKED) > retired) {
                c->dead = true;                                             // the frame may still be running: nobody may reuse its buffers or the ring
                return -2;
            }
            break;
        }
    }
    return 0;
}

int rc_aql_run(AqlChain* c, int prog) {
    if (!c || !c->q || prog < 0 || prog >= (int)c->prog.size() || !c->prog[prog].frame) return -1;
    if (c->dead) return -3;
    // `done` is never re-armed: every retired frame decrements it once (sig0 - seq when frame seq has retired)
    aql_push(c, c->prog[prog]);
    c->seq += 1;
    return aql_wait_frame(c, c->seq);
}

// the last frame has retired (its dispatch packets are consumed, its release fence has run): before anything else touches the queue
static void aql_drain(AqlChain* c) {
    if (!c || !c->q) return;
    if (c->mailbox) { c->mailbox[0] = 2u; c->mailbox[32] = 2u; RC_STORE_FENCE(); }                  // a K1 still spinning (rc_live.hip) leaves: nothing is queued behind it
    aql_release(c);
    if (c->mailbox && !c->dead) (void)rc_aql_fence_background(c);           // ... and the wait below covers it
    if (c->seq) (void)hsa_signal_wait_scacquire(c->done, HSA_SIGNAL_CONDITION_LT, (hsa_signal_value_t)(c->sig0 - (long long)c->seq) + 1, 2000000000ull, HSA_WAIT_STATE_BLOCKED);
    (void)rc_aql_wait_background(c);
}

void rc_aql_destroy(AqlChain* c) {
    if (!c) return;
    aql_drain(c);
    if (c->flag_h) (void)hipHostFree(c->flag_h);
    if (c->seq_d) (void)hipFree(c->seq_d);
    if (c->q) (void)hsa_queue_destroy(c->q);
    for (void* q : c->shared) (void)hsa_amd_memory_pool_free(q);
    if (c->done.handle) (void)hsa_signal_destroy(c->done);
    if (c->bg_done.handle) (void)hsa_signal_destroy(c->bg_done);
    for (int a = 0; a < 2; ++a) if (c->arm[a].handle) (void)hsa_signal_destroy(c->arm[a]);
    for (AqlProgram& P : c->prog) if (P.kargs) (void)hipFree(P.kargs);
    if (c->hsa_up) (void)hsa_shut_down();
    delete c;
}
