// dma_copy.h — a block's probability stream leaves the device through a DMA engine, whatever the HIP runtime in the process would pick.
//
// Why this exists (round 6, profiles/r06/d2h_copy_path.txt): hipMemcpyAsync(device -> pinned host) is a DMA-engine copy on the system's
// HIP runtime (ROCm 7.2) but a copy KERNEL (__amd_rocclr_copyBuffer, 256 workgroups x 1024 threads) on the runtime a torch process
// carries (torch/lib/libamdhip64.so, whose ROCr reports no recommended engine: rocclr then takes engine 1, and that one is the shader) —
// sixteen wavefronts resident on every CU for the ~0.8 ms the PCIe link needs per 45 MB piece, eight pieces per block.  No environment
// variable of that runtime changes it.  The HSA call underneath is the same in both, so the eight pieces of a block are issued through
// it directly: hsa_amd_memory_async_copy from the device buffer to the (registered) landing zone's device address, one HSA signal per
// piece, the coder tasks wait on the signals.  Everything else (small copies, H2D of host-resident input) stays with the HIP runtime.
// Internal.  The path is optional at run time (BSC_D2H_DMA=0, or HSA not usable): then the copies are hipMemcpyAsync as before.
// BSC_D2H_ENGINE=<mask>[,<mask>..]: explicit engine(s) for hsa_amd_memory_async_copy_on_engine, taken in turn (experiments).
#pragma once
#include <cstddef>
#include <cstdint>

// 1 when the HSA path is usable in this process (resolved once, thread-safe)
int  dma_available();
// completion objects: an HSA signal's handle (0 = none)
uint64_t dma_signal_create();
void     dma_signal_destroy(uint64_t sig);
// dst_dev: the DEVICE address of pinned host memory (hipHostGetDevicePointer); src: device memory.  Arms the signal (value 1) and queues
// the copy; the signal reaches 0 when the bytes have landed.  bytes == 0: the signal is completed at once.  0 on success.
int  dma_d2h(void* dst_dev, const void* src, size_t bytes, uint64_t sig);
// blocks (sleeping, not spinning) until the signal's copy has landed; 0 on success, -1 when the signal reports an error value
int  dma_wait(uint64_t sig);
