// radix_dev.h — device helpers shared by the digit-pass kernels (radix_sort.hip, radix_onesweep.hip): the wave64 ballot-match
// ranking block and the per-digit exclusive sum.  Internal.
#pragma once
#include "dev_common.h"

// Volatile accesses must keep the LDS address space: through a generic `volatile u32*` the compiler emits system-scope
// FLAT loads/stores followed by s_waitcnt vmcnt(0) — every in-wave rank step would then drain all outstanding global
// loads and stores of the wave.
#ifndef RS_FLAT_VOLATILE
#define RS_FLAT_VOLATILE 0     // 1 = the old generic-pointer volatile (A/B builds only)
#endif
#if RS_FLAT_VOLATILE
typedef volatile u32 lds_vu32;
#else
typedef __attribute__((address_space(3))) volatile u32 lds_vu32;
#endif
constexpr int RS_WAVES_DEFAULT = 4;

// (the DPP form of the wave scan lives in dev_common.h since round 6: wave_incl_sum)
__device__ __forceinline__ u32 wave_incl_sum_dpp(u32 v) { return wave_incl_sum(v); }

// Exclusive sum over the first 256 threads' values (one per digit); every thread of the workgroup calls it
// (threads >= 256 pass 0).  scr: RS_WAVES u32.
// LEAD_BARRIER = false: the caller guarantees that nobody still reads scr from an earlier call.
template <int WAVES = RS_WAVES_DEFAULT, bool LEAD_BARRIER = true, bool DPP = false>
__device__ __forceinline__ u32 rs_digit_excl_sum(u32 v, u32* scr, u32* total) {
    const u32 incl = DPP ? wave_incl_sum_dpp(v) : wave_incl_sum(v);
    const u32 w = threadIdx.x >> 6, l = lane_id();
    if (LEAD_BARRIER) __syncthreads();
    if (l == 63) scr[w] = incl;
    __syncthreads();
    u32 base = 0, tot = 0;
#pragma unroll
    for (int i = 0; i < WAVES; ++i) { const u32 t = scr[i]; if ((u32)i < w) base += t; tot += t; }
    *total = tot;
    return base + incl - v;
}

// ---------------------------------------------------------------------------------------------
// In-wave match on an 8-bit digit: (mlo, mhi) = mask of the lanes whose digit equals this lane's.  Per digit bit: one
// sign-extended bit extract (t = all-ones if the bit is set), one compare that IS the ballot (it writes an SGPR pair), and
// one 3-input boolean op per 32-bit half, m &= ~(ballot ^ t) (v_bitop3 table 0x90) — 32 VALU instructions per record.
// Written as one asm block because the compiler's own lowering of the same expression takes ~8 instructions per bit
// (shift, compare, not, arithmetic shift, two xors, two 3-input ands), and the ranking loop is what the scatter kernels'
// VALU time goes into.  gfx950 needs two wait states between a VALU write of an SGPR and a VALU read of it, so two SGPR
// pairs (vcc and s[98:99]) alternate and every ballot is consumed three or more instructions after it was produced.
// All 64 lanes must be active.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ void rs_match8(const u32 d, u32& mlo, u32& mhi)
{
    u32 lo = ~0u, hi = ~0u, t0, t1, t2;
    asm volatile(
        "v_bfe_i32 %2, %5, 0, 1\n\t"
        "v_bfe_i32 %3, %5, 1, 1\n\t"
        "v_cmp_ne_u32_e32 vcc, 0, %2\n\t"
        "v_cmp_ne_u32_e64 s[98:99], 0, %3\n\t"
        "v_bfe_i32 %4, %5, 2, 1\n\t"
        "v_bitop3_b32 %0, %0, vcc_lo, %2 bitop3:0x90\n\t"
        "v_bitop3_b32 %1, %1, vcc_hi, %2 bitop3:0x90\n\t"
        "v_cmp_ne_u32_e32 vcc, 0, %4\n\t"
        "v_bfe_i32 %2, %5, 3, 1\n\t"
        "v_bitop3_b32 %0, %0, s98, %3 bitop3:0x90\n\t"
        "v_bitop3_b32 %1, %1, s99, %3 bitop3:0x90\n\t"
        "v_cmp_ne_u32_e64 s[98:99], 0, %2\n\t"
        "v_bfe_i32 %3, %5, 4, 1\n\t"
        "v_bitop3_b32 %0, %0, vcc_lo, %4 bitop3:0x90\n\t"
        "v_bitop3_b32 %1, %1, vcc_hi, %4 bitop3:0x90\n\t"
        "v_cmp_ne_u32_e32 vcc, 0, %3\n\t"
        "v_bfe_i32 %4, %5, 5, 1\n\t"
        "v_bitop3_b32 %0, %0, s98, %2 bitop3:0x90\n\t"
        "v_bitop3_b32 %1, %1, s99, %2 bitop3:0x90\n\t"
        "v_cmp_ne_u32_e64 s[98:99], 0, %4\n\t"
        "v_bfe_i32 %2, %5, 6, 1\n\t"
        "v_bitop3_b32 %0, %0, vcc_lo, %3 bitop3:0x90\n\t"
        "v_bitop3_b32 %1, %1, vcc_hi, %3 bitop3:0x90\n\t"
        "v_cmp_ne_u32_e32 vcc, 0, %2\n\t"
        "v_bfe_i32 %3, %5, 7, 1\n\t"
        "v_bitop3_b32 %0, %0, s98, %4 bitop3:0x90\n\t"
        "v_bitop3_b32 %1, %1, s99, %4 bitop3:0x90\n\t"
        "v_cmp_ne_u32_e64 s[98:99], 0, %3\n\t"
        "v_bitop3_b32 %0, %0, vcc_lo, %2 bitop3:0x90\n\t"
        "v_bitop3_b32 %1, %1, vcc_hi, %2 bitop3:0x90\n\t"
        "v_bitop3_b32 %0, %0, s98, %3 bitop3:0x90\n\t"
        "v_bitop3_b32 %1, %1, s99, %3 bitop3:0x90"
        : "+v"(lo), "+v"(hi), "=&v"(t0), "=&v"(t1), "=&v"(t2)
        : "v"(d)
        : "vcc", "s98", "s99");
    mlo = lo; mhi = hi;
}

// Stable rank of ITEMS records per lane inside the wave's 64 * ITEMS records (item-major order: item i of all lanes comes
// before item i + 1), by digit: rk[i] = number of earlier records of the wave with the same digit.  wh = the wave's 256
// digit counters in LDS (zeroed by the caller), left holding the wave's digit histogram.
template <int ITEMS>
__device__ __forceinline__ void rs_rank_wave(const u64 (&k)[ITEMS], const int shift, const u32 mask, lds_vu32* wh, u32 (&rk)[ITEMS])
{
#pragma unroll
    for (int i = 0; i < ITEMS; ++i) {
        const u32 d = (u32)(k[i] >> shift) & mask;
        u32 mlo, mhi;
        rs_match8(d, mlo, mhi);
        const u32 before = wh[d];                   // records of digit d seen by this wave so far
        const u32 r      = __builtin_amdgcn_mbcnt_hi(mhi, __builtin_amdgcn_mbcnt_lo(mlo, 0u));   // peers in lower lanes
        const u32 cnt    = (u32)(__popc(mlo) + __popc(mhi));
        rk[i] = before + r;
        if (r == cnt - 1) wh[d] = before + cnt;     // highest peer lane publishes
    }
}

