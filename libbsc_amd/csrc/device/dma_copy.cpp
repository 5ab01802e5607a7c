// dma_copy.cpp — see dma_copy.h.  The HSA runtime is the one ALREADY in the process (the HIP runtime brought it): it is looked up with
// RTLD_NOLOAD and called through function pointers, so the library gains no link-time dependency and can never load a second copy.
#include "dma_copy.h"
#include <dlfcn.h>
#include <hsa/hsa.h>
#include <hsa/hsa_ext_amd.h>
#include <cstdio>
#include <cstdlib>
#include <atomic>
#include <mutex>

namespace {
struct Hsa {
    bool ok = false;
    int engines[4] = {0, 0, 0, 0}; int nengines = 0;   // BSC_D2H_ENGINE: empty = the runtime's device-to-host engine (hsa_amd_memory_async_copy); else up to
                                                       // four engine masks for ..._on_engine, taken in turn by successive copies ("2,4": two engines side by side)
    std::atomic<unsigned> turn{0};
    hsa_agent_t cpu{};
    decltype(&hsa_init) init = nullptr;
    decltype(&hsa_iterate_agents) iterate_agents = nullptr;
    decltype(&hsa_agent_get_info) agent_get_info = nullptr;
    decltype(&hsa_signal_create) signal_create = nullptr;
    decltype(&hsa_signal_destroy) signal_destroy = nullptr;
    decltype(&hsa_signal_store_relaxed) signal_store_relaxed = nullptr;
    decltype(&hsa_signal_store_screlease) signal_store_screlease = nullptr;
    decltype(&hsa_signal_wait_scacquire) signal_wait_scacquire = nullptr;
    decltype(&hsa_amd_pointer_info) pointer_info = nullptr;
    decltype(&hsa_amd_memory_async_copy) memory_async_copy = nullptr;
    decltype(&hsa_amd_memory_async_copy_on_engine) memory_async_copy_on_engine = nullptr;
};
Hsa g;
std::once_flag g_once;

hsa_status_t find_cpu(hsa_agent_t a, void* out)
{
    hsa_device_type_t t;
    if (g.agent_get_info(a, HSA_AGENT_INFO_DEVICE, &t) == HSA_STATUS_SUCCESS && t == HSA_DEVICE_TYPE_CPU) { *(hsa_agent_t*)out = a; return HSA_STATUS_INFO_BREAK; }
    return HSA_STATUS_SUCCESS;
}

void resolve()
{
    if (const char* e = getenv("BSC_D2H_DMA")) if (atoi(e) == 0) return;
    if (const char* e = getenv("BSC_D2H_ENGINE")) {
        while (*e && g.nengines < 4) { char* end = nullptr; const long v = strtol(e, &end, 0); if (end == e) break; if (v > 0) g.engines[g.nengines++] = (int)v; e = (*end == ',') ? end + 1 : end; }
    }
    void* h = dlopen("libhsa-runtime64.so.1", RTLD_NOW | RTLD_NOLOAD);
    if (!h) return;                                                     // no HSA runtime in the process: not a ROCm HIP runtime we know
#define SYM(field, name) g.field = (decltype(g.field))dlsym(h, name); if (!g.field) return
    SYM(init, "hsa_init"); SYM(iterate_agents, "hsa_iterate_agents"); SYM(agent_get_info, "hsa_agent_get_info");
    SYM(signal_create, "hsa_signal_create"); SYM(signal_destroy, "hsa_signal_destroy"); SYM(signal_store_relaxed, "hsa_signal_store_relaxed");
    SYM(signal_store_screlease, "hsa_signal_store_screlease"); SYM(signal_wait_scacquire, "hsa_signal_wait_scacquire");
    SYM(pointer_info, "hsa_amd_pointer_info"); SYM(memory_async_copy, "hsa_amd_memory_async_copy");
#undef SYM
    g.memory_async_copy_on_engine = (decltype(g.memory_async_copy_on_engine))dlsym(h, "hsa_amd_memory_async_copy_on_engine");
    if (g.nengines != 0 && !g.memory_async_copy_on_engine) g.nengines = 0;
    if (g.init() != HSA_STATUS_SUCCESS) return;                         // reference counted: the HIP runtime's own initialisation stands
    hsa_agent_t cpu{}; cpu.handle = 0;
    const hsa_status_t st = g.iterate_agents(find_cpu, &cpu);
    if ((st != HSA_STATUS_SUCCESS && st != HSA_STATUS_INFO_BREAK) || cpu.handle == 0) return;
    g.cpu = cpu;
    g.ok = true;
}
}  // namespace

int dma_available() { std::call_once(g_once, resolve); return g.ok ? 1 : 0; }

uint64_t dma_signal_create()
{
    if (!dma_available()) return 0;
    hsa_signal_t s{};
    if (g.signal_create(0, 0, nullptr, &s) != HSA_STATUS_SUCCESS) return 0;
    return s.handle;
}

void dma_signal_destroy(uint64_t sig)
{
    if (sig && g.ok) { hsa_signal_t s; s.handle = sig; g.signal_destroy(s); }
}

int dma_d2h(void* dst_dev, const void* src, size_t bytes, uint64_t sig)
{
    if (!g.ok || !sig) return -1;
    hsa_signal_t s; s.handle = sig;
    if (bytes == 0) { g.signal_store_screlease(s, 0); return 0; }
    hsa_amd_pointer_info_t info; info.size = sizeof info;
    if (g.pointer_info(const_cast<void*>(src), &info, nullptr, nullptr, nullptr) != HSA_STATUS_SUCCESS || info.type == HSA_EXT_POINTER_TYPE_UNKNOWN) return -1;
    const hsa_agent_t gpu = info.agentOwner;
    g.signal_store_relaxed(s, 1);
    hsa_status_t st;
    if (g.nengines != 0) {
        const int engine = g.engines[g.turn.fetch_add(1, std::memory_order_relaxed) % (unsigned)g.nengines];
        st = g.memory_async_copy_on_engine(dst_dev, g.cpu, src, gpu, bytes, 0, nullptr, s, (hsa_amd_sdma_engine_id_t)engine, true);
    } else st = g.memory_async_copy(dst_dev, g.cpu, src, gpu, bytes, 0, nullptr, s);
    if (st != HSA_STATUS_SUCCESS) { g.signal_store_screlease(s, 0); return -1; }
    return 0;
}

int dma_wait(uint64_t sig)
{
    if (!g.ok || !sig) return -1;
    hsa_signal_t s; s.handle = sig;
    // (the runtime sets a negative value on a failed copy)
    hsa_signal_value_t v;
    do { v = g.signal_wait_scacquire(s, HSA_SIGNAL_CONDITION_LT, 1, UINT64_MAX, HSA_WAIT_STATE_BLOCKED); } while (v >= 1);
    return v == 0 ? 0 : -1;
}
