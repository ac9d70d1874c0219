// fft_device.h -- device-side vocabulary shared by every kernel of the engine.
//
// The kernels are written once.  hipcc compiles them for gfx950 (the product);
// tests/emu compiles the SAME source with g++ and -DFFT_EMU, where a workgroup
// is a set of host threads and __syncthreads() a barrier, so that the index
// algebra (Stockham digit order, four-step strides, LDS layouts) can be checked
// in the build container, which has no GPU.  The emulation is test
// infrastructure: nothing in the shipped library is built with FFT_EMU.
#pragma once

#include <stdint.h>

#if defined(FFT_EMU)
#include <cstring>
namespace emu {
struct dim3_ { unsigned x, y, z; };
extern thread_local dim3_ threadIdx_;
extern thread_local dim3_ blockIdx_;
extern dim3_ blockDim_;
extern dim3_ gridDim_;
extern thread_local unsigned char* smem_;  // per workgroup (workgroups of a team launch run concurrently)
void sync_threads();
long long clock_ticks();
extern int xcc_skew;  // test hook: workgroup 0 reports its neighbour's XCD, so no launch yields full teams
unsigned shfl_xor_u32(unsigned v, int mask);
}  // namespace emu
#define FFT_KERNEL
#define FFT_DEVICE inline
#define FFT_HOST_DEVICE inline
#define FFT_TID ((int)emu::threadIdx_.x)
#define FFT_BID ((long long)emu::blockIdx_.x)
#define FFT_NTHREADS ((int)emu::blockDim_.x)
#define FFT_NBLOCKS ((long long)emu::gridDim_.x)
#define FFT_DYN_SMEM(name) unsigned char* name = emu::smem_
#define FFT_SYNC() emu::sync_threads()
#define FFT_SYNC_LDS() emu::sync_threads()
#define FFT_WAIT_LOADED(v) (void)(v)
namespace emu {
template <typename T>
inline T shfl_xor_any(T v, int mask) {
    static_assert(sizeof(T) % 4 == 0, "32-bit granules");
    unsigned w[sizeof(T) / 4];
    memcpy(w, &v, sizeof(T));
    for (unsigned i = 0; i < sizeof(T) / 4; i++) w[i] = shfl_xor_u32(w[i], mask);
    memcpy(&v, w, sizeof(T));
    return v;
}
}  // namespace emu
#define FFT_SHFL_XOR(v, mask) emu::shfl_xor_any(v, mask)
#define FFT_XOR_EXCHANGE(v, mask, upper) emu::shfl_xor_any(v, mask)
#define FFT_OPAQUE(v) (void)(v)
#define FFT_LAUNCH_BOUNDS(n)
#define FFT_LAUNCH_BOUNDS2(n, w)
#define FFT_RESTRICT
#define FFT_UNROLL
#define FFT_NOUNROLL
#define FFT_UNROLL_N(n)
// ---- team-kernel vocabulary (fft_team.h): the emulation runs the workgroups of a launch concurrently, a "XCD"
// is blockIdx mod n_teams, every memory operation is sequentially consistent
#include <sched.h>
#define FFT_XCC_ID(nteams) ((unsigned)((emu::blockIdx_.x + ((emu::xcc_skew && emu::blockIdx_.x == 0) ? 1u : 0u)) % (unsigned)(nteams)))
#define FFT_ATOMIC_ADD_AGENT(p, v) __atomic_fetch_add((p), (v), __ATOMIC_SEQ_CST)
#define FFT_ATOMIC_ADD_AGENT_RELAXED(p, v) __atomic_fetch_add((p), (v), __ATOMIC_SEQ_CST)
#define FFT_ATOMIC_ADD_AGENT_RELEASE(p, v) __atomic_fetch_add((p), (v), __ATOMIC_SEQ_CST)
#define FFT_FENCE_ACQUIRE_AGENT() __atomic_thread_fence(__ATOMIC_SEQ_CST)
#define FFT_ATOMIC_LOAD_AGENT(p) __atomic_load_n((p), __ATOMIC_SEQ_CST)
#define FFT_ATOMIC_STORE_AGENT(p, v) __atomic_store_n((p), (v), __ATOMIC_SEQ_CST)
namespace emu {
inline unsigned cas_u32(unsigned* p, unsigned expected, unsigned desired) {  // returns the value found
    __atomic_compare_exchange_n(p, &expected, desired, false, __ATOMIC_SEQ_CST, __ATOMIC_SEQ_CST);
    return expected;
}
void test_delay();  // test hook: one chosen workgroup sleeps before it registers (FFT_EMU_LATE_BLOCK / _MS)
bool test_drop();   // test hook: this workgroup leaves right after the team has formed (FFT_EMU_DROP_BLOCK): a hung member
}  // namespace emu
#define FFT_ATOMIC_CAS_AGENT(p, expected, desired) emu::cas_u32((p), (expected), (desired))
#define FFT_TEST_DELAY() emu::test_delay()
#define FFT_TEST_DROP() emu::test_drop()
#define FFT_L2_FLAG_STORE(p, v) __atomic_store_n((p), (v), __ATOMIC_SEQ_CST)
#define FFT_L2_FLAG_LOAD(p) __atomic_load_n((p), __ATOMIC_SEQ_CST)
#define FFT_L2_COUNT_ADD(p) ((void)__atomic_fetch_add((p), 1u, __ATOMIC_SEQ_CST))
#define FFT_L2_COUNT_POLL(p) __atomic_load_n((p), __ATOMIC_SEQ_CST)
// the counter and the word behind it (read AFTER the counter): low half = counter, high half = that word
#define FFT_L2_COUNT_POLL2(p) ((unsigned long long)__atomic_load_n((p), __ATOMIC_SEQ_CST) | ((unsigned long long)__atomic_load_n((p) + 1, __ATOMIC_SEQ_CST) << 32))
// the pair protocol of team_quad_kernel (SLOTS = 3): 8 bytes stored / loaded as one; the unit that signals and guards for itself (a wave
// on the device, a thread here: host threads of a "wave" do not run in lockstep)
#define FFT_L2_STORE64(p, lo, hi) __atomic_store_n((unsigned long long*)(p), (unsigned long long)(lo) | ((unsigned long long)(hi) << 32), __ATOMIC_SEQ_CST)
#define FFT_L2_LOAD64(p) __atomic_load_n((const unsigned long long*)(p), __ATOMIC_SEQ_CST)
#define FFT_PAIR_UNIT 1
#define FFT_LDS_FRESH() __atomic_thread_fence(__ATOMIC_SEQ_CST)
#define FFT_SCHED_BARRIER() ((void)0)
#define FFT_WAIT_VM0() __atomic_thread_fence(__ATOMIC_SEQ_CST)
#define FFT_SLEEP() sched_yield()
#define FFT_CLOCK() emu::clock_ticks()
#define FFT_UNIFORM(v) (v)
#define FFT_LDS_ADDR(ptr) 0u
#define FFT_DMA16(gsrc, lds_base_ptr, lds_base_addr, off) memcpy((lds_base_ptr) + (off), (gsrc), 16)
#define FFT_DMA16_L2(gsrc, lds_base_ptr, lds_base_addr, off) memcpy((lds_base_ptr) + (off), (gsrc), 16)
#define FFT_DMA16_NT(gsrc, lds_base_ptr, lds_base_addr, off) memcpy((lds_base_ptr) + (off), (gsrc), 16)
#define FFT_DMA16_L2_NT(gsrc, lds_base_ptr, lds_base_addr, off) memcpy((lds_base_ptr) + (off), (gsrc), 16)
#define FFT_STORE16_NT(ptr, v) (*(ptr) = (v))
#define FFT_STORE16_SC1(ptr, v) (*(ptr) = (v))
template <int NT, class V16>
inline V16 fft_ld16(const V16* p) { return *p; }
template <int NT, class V16>
inline void fft_st16(V16* p, const V16& v) { *p = v; }
#define FFT_WAIT_VM_LE(n) __atomic_thread_fence(__ATOMIC_SEQ_CST)
// the lanes of a wave run in lock step on the device: LDS writes that follow LDS reads in program order can never overtake
// another lane's reads.  The emulation's "lanes" are free-running host threads: a workgroup barrier stands in.
#define FFT_WAVE_LOCKSTEP() emu::sync_threads()
#define FFT_LDS_LD64(p) (*(p))
#define FFT_LDS_ST64(p, v) (*(p) = (v))
#else
#include <hip/hip_runtime.h>
#define FFT_KERNEL __global__
#define FFT_DEVICE __device__ __forceinline__
#define FFT_HOST_DEVICE __host__ __device__ __forceinline__
#define FFT_TID ((int)threadIdx.x)
#define FFT_BID ((long long)blockIdx.x)
#define FFT_NTHREADS ((int)blockDim.x)
#define FFT_NBLOCKS ((long long)gridDim.x)
#define FFT_DYN_SMEM(name) extern __shared__ __attribute__((aligned(16))) unsigned char name[]
#define FFT_SYNC() __syncthreads()
// Workgroup barrier that orders LDS traffic only: waits for this wave's LDS operations (lgkmcnt(0)) and
// joins the barrier WITHOUT draining outstanding global loads (vmcnt untouched), so a prefetch of the next
// tile issued before the barrier stays in flight across it (__syncthreads() would add s_waitcnt vmcnt(0)).
#define FFT_SYNC_LDS() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")
// A fake use of a loaded register: makes the compiler place the s_waitcnt vmcnt for that load HERE.  vmcnt is
// an in-order counter shared by loads and stores, so waiting for the prefetched tile BEFORE this tile's
// stores are issued keeps the store latency out of the next iteration's critical path.
#define FFT_WAIT_LOADED(v) asm volatile("" ::"v"(v))
#define FFT_SHFL_XOR(v, mask) __shfl_xor(v, mask, 64)
// lane ^ mask exchange of one float; `upper` = this lane has the mask bit set.  Masks 1 and 8 stay inside a row of 16 lanes
// and are DPP moves (quad_perm [1,0,3,2], row_ror:8); 16 and 32 are the gfx950 row / half swaps (v_permlane16_swap,
// v_permlane32_swap: the odd rows / upper half of the first operand trade places with the even rows / lower half of the
// second -- with the same value in both, the lane finds its partner's value in the first result if `upper`, else in the
// second).  All VALU-rate: no trip through the LDS crossbar and no lgkmcnt wait like the ds_bpermute behind __shfl_xor.
#define FFT_XOR_EXCHANGE(v, mask, upper) fft_xor_exchange((v), (mask), (upper))
__device__ __forceinline__ float fft_xor_exchange(float v, int mask, bool upper) {
    const int iv = __builtin_bit_cast(int, v);
    if (mask == 1) return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, iv, 0xB1, 0xF, 0xF, true));
    if (mask == 8) return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, iv, 0x128, 0xF, 0xF, true));
    if (mask == 16) {
        const auto r = __builtin_amdgcn_permlane16_swap((unsigned)iv, (unsigned)iv, false, false);
        return __builtin_bit_cast(float, upper ? r[0] : r[1]);
    }
    if (mask == 32) {
        const auto r = __builtin_amdgcn_permlane32_swap((unsigned)iv, (unsigned)iv, false, false);
        return __builtin_bit_cast(float, upper ? r[0] : r[1]);
    }
    return __shfl_xor(v, mask, 64);
}
__device__ __forceinline__ double fft_xor_exchange(double v, int mask, bool) { return __shfl_xor(v, mask, 64); }
// Hide a loop-invariant value from the optimizer: without this, LICM hoists every per-stage LDS address and
// twiddle index out of the persistent tile loop and keeps hundreds of them live in VGPRs across it.
#ifdef FFT_NO_OPAQUE
#define FFT_OPAQUE(v) (void)(v)
#else
#define FFT_OPAQUE(v) asm volatile("" : "+v"(v))
#endif
#define FFT_LAUNCH_BOUNDS(n) __launch_bounds__(n)
#define FFT_LAUNCH_BOUNDS2(n, w) __launch_bounds__(n, w)
#define FFT_RESTRICT __restrict__
#define FFT_UNROLL _Pragma("unroll")
#define FFT_NOUNROLL _Pragma("nounroll")
#define FFT_PRAGMA_(x) _Pragma(#x)
#define FFT_UNROLL_N(n) FFT_PRAGMA_(unroll n)  // n: an integral constant expression (1 = keep the loop)
// ---- team-kernel vocabulary (fft_team.h)
// The XCD this wave runs on (HW_REG_XCC_ID, bits 3:0).  Workgroups that read the same id share one L2.
#define FFT_XCC_ID(nteams) ((unsigned)__builtin_amdgcn_s_getreg((3 << 11) | (0 << 6) | 20) & 15u)
// device-coherent (agent-scope) atomics: team formation and the status word only
#define FFT_ATOMIC_ADD_AGENT(p, v) __hip_atomic_fetch_add((p), (v), __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT)
// ... without ordering (no L2 write-back / invalidate around it): a counter that guards no data (team_quad_kernel's work claims)
#define FFT_ATOMIC_ADD_AGENT_RELAXED(p, v) __hip_atomic_fetch_add((p), (v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
#define FFT_ATOMIC_ADD_AGENT_RELEASE(p, v) __hip_atomic_fetch_add((p), (v), __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT)
#define FFT_FENCE_ACQUIRE_AGENT() __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent")
#define FFT_ATOMIC_LOAD_AGENT(p) __hip_atomic_load((p), __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT)
#define FFT_ATOMIC_STORE_AGENT(p, v) __hip_atomic_store((p), (v), __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT)
__device__ __forceinline__ unsigned fft_cas_agent(unsigned* p, unsigned expected, unsigned desired) {  // returns the value found
    __hip_atomic_compare_exchange_strong(p, &expected, desired, __ATOMIC_ACQ_REL, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT);
    return expected;
}
#define FFT_ATOMIC_CAS_AGENT(p, expected, desired) fft_cas_agent((p), (expected), (desired))
#define FFT_TEST_DELAY() ((void)0)
#define FFT_TEST_DROP() false
// Same-XCD signalling through the XCD's own L2: a PLAIN dword store stays in L2 (write-through L1), an sc1 load
// bypasses the reader's L1 and is served by that L2.  Valid ONLY between workgroups that read the same XCC id.
// (hand-written: a volatile C store comes out as flat_store_dword sc0 sc1, i.e. written through to memory)
#define FFT_L2_FLAG_STORE(p, v) asm volatile("global_store_dword %0, %1, off" ::"v"((unsigned*)(p)), "v"((unsigned)(v)) : "memory")
#define FFT_L2_FLAG_LOAD(p) __hip_atomic_load((p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
// Arrival counter of a team in the XCD's L2: the add is performed by the L2 (no return value, nothing to wait for);
// the poll is a SCALAR load (glc: past the scalar cache, served by that L2).  A vector load would do, but its result
// could only be looked at behind `s_waitcnt vmcnt(0)`, i.e. behind every store and LDS-DMA the wave still has in
// flight -- a poll that drains the memory pipeline it is supposed to let run.  Scalar loads count on lgkmcnt.
#define FFT_L2_COUNT_ADD(p) asm volatile("global_atomic_add %0, %1, off" ::"v"((unsigned*)(p)), "v"(1u) : "memory")
#define FFT_L2_COUNT_POLL(p) fft_scalar_load_glc((const unsigned*)(p))
__device__ __forceinline__ unsigned fft_scalar_load_glc(const unsigned* p) {
    unsigned v;
    asm volatile("s_load_dword %0, %1, 0x0 glc\n\ts_waitcnt lgkmcnt(0)" : "=s"(v) : "s"(p) : "memory");
    return v;
}
// the counter and the word behind it in ONE 8-byte access (p 8-byte aligned): a value stored to that word before an arrival is seen by
// every poll that sees the arrival -- a second load would cost a second L2 round trip on the critical path (scalar loads wait 1 - 4 us
// behind a busy L2: profiles/r3_ab_dynamic.txt)
#define FFT_L2_COUNT_POLL2(p) fft_scalar_load2_glc((const unsigned*)(p))
__device__ __forceinline__ unsigned long long fft_scalar_load2_glc(const unsigned* p) {
    unsigned long long v;
    asm volatile("s_load_dwordx2 %0, %1, 0x0 glc\n\ts_waitcnt lgkmcnt(0)" : "=s"(v) : "s"(p) : "memory");
    return v;
}
// the pair protocol of team_quad_kernel (SLOTS = 3): 8 bytes stored / loaded as one (p 8-byte aligned), eight adjacent words in one scalar
// load (p 32-byte aligned); the unit that signals and guards for itself is a wave
#define FFT_L2_STORE64(p, lo, hi) fft_l2_store64((unsigned*)(p), (unsigned)(lo), (unsigned)(hi))
__device__ __forceinline__ void fft_l2_store64(unsigned* p, unsigned lo, unsigned hi) {
    typedef unsigned fft_u32x2_ __attribute__((ext_vector_type(2)));
    const fft_u32x2_ v = {lo, hi};
    asm volatile("global_store_dwordx2 %0, %1, off" ::"v"(p), "v"(v) : "memory");
}
#define FFT_L2_LOAD64(p) fft_scalar_load2_glc((const unsigned*)(p))
typedef unsigned fft_u32x8 __attribute__((ext_vector_type(8)));
__device__ __forceinline__ fft_u32x8 fft_scalar_load8_glc(const unsigned* p) {
    fft_u32x8 v;
    asm volatile("s_load_dwordx8 %0, %1, 0x0 glc\n\ts_waitcnt lgkmcnt(0)" : "=s"(v) : "s"(p) : "memory");
    return v;
}
typedef unsigned fft_u32x4s __attribute__((ext_vector_type(4)));
__device__ __forceinline__ fft_u32x4s fft_scalar_load4_glc(const unsigned* p) {
    fft_u32x4s v;
    asm volatile("s_load_dwordx4 %0, %1, 0x0 glc\n\ts_waitcnt lgkmcnt(0)" : "=s"(v) : "s"(p) : "memory");
    return v;
}
#define FFT_PAIR_UNIT 64
// the next read of a (non-volatile) LDS word is a real ds_read: volatile accesses through a pointer whose address
// space the compiler has to infer come out as FLAT loads, which wait on vmcnt as well
#define FFT_LDS_FRESH() asm volatile("" ::: "memory")
#define FFT_SCHED_BARRIER() __builtin_amdgcn_sched_barrier(0)  // the instruction scheduler moves nothing across this point
#define FFT_WAIT_VM0() asm volatile("s_waitcnt vmcnt(0)" ::: "memory")
#ifndef FFT_SLEEP_N
#define FFT_SLEEP_N 2  // (64-clock units between two polls of a team wait; 0, 1, 4 measured: tools/ab_quad.sh)
#endif
#define FFT_SLEEP() __builtin_amdgcn_s_sleep(FFT_SLEEP_N)
#define FFT_CLOCK() ((long long)wall_clock64())
#define FFT_UNIFORM(v) __builtin_amdgcn_readfirstlane(v)
// LDS-DMA: 16 bytes per lane from global memory straight into LDS (global_load_lds_dwordx4), no VGPR destination.
// The 64 lanes of a wave land in 1 KiB CONTIGUOUS LDS bytes starting at the first lane's LDS address: every call
// site passes the image base + 16 * lane-linear index.  Counts in vmcnt.
// Hand-issued (inline asm; the builtin also silently drops the kernel's host stub in hipcc's host pass): hipcc does
// not see it, so it neither counts it in its own s_waitcnt bookkeeping
// (its waits only get stricter: vmcnt is in order) nor drains it in front of LDS reads that "may alias" -- the kernel
// waits for its DMA itself (FFT_WAIT_VM0 / FFT_WAIT_VM_LE + barrier) before anybody reads the landing image.
//   FFT_DMA16     from HBM, default cache policy        FFT_DMA16_L2  sc1: served by the XCD's L2, never the vector L1
//   FFT_LDS_ADDR  the 32-bit LDS byte address of a __shared__ pointer (taken once per kernel)
// `off` is the byte offset of THIS lane's 16 bytes from the image base; the wave's first lane gives M0.
#define FFT_LDS_ADDR(ptr) ((unsigned)(uintptr_t)(__attribute__((address_space(3))) void*)(ptr))
#define FFT_DMA16(gsrc, lds_base_ptr, lds_base_addr, off) fft_dma16<0>((gsrc), (lds_base_addr) + (unsigned)(off))
#define FFT_DMA16_L2(gsrc, lds_base_ptr, lds_base_addr, off) fft_dma16<1>((gsrc), (lds_base_addr) + (unsigned)(off))
#define FFT_DMA16_NT(gsrc, lds_base_ptr, lds_base_addr, off) fft_dma16<2>((gsrc), (lds_base_addr) + (unsigned)(off))
#define FFT_DMA16_L2_NT(gsrc, lds_base_ptr, lds_base_addr, off) fft_dma16<3>((gsrc), (lds_base_addr) + (unsigned)(off))
// 16-byte store with the non-temporal (streaming) hint: the line is the first to leave the L2
#define FFT_STORE16_NT(ptr, v) fft_store16_nt((ptr), (v))
typedef unsigned fft_u32x4 __attribute__((ext_vector_type(4)));
template <class V16>
__device__ __forceinline__ void fft_store16_nt(V16* ptr, const V16& v) {
    static_assert(sizeof(V16) == 16, "one 16-byte lane access");
    fft_u32x4 raw;
    __builtin_memcpy(&raw, &v, 16);
    // hand-written: __builtin_nontemporal_store of a 16-byte vector comes out as a plain global_store_dwordx4 here
    asm volatile("global_store_dwordx4 %0, %1, off nt" ::"v"(ptr), "v"(raw) : "memory");
}
// 16-byte accesses of the tile kernels' HBM streams, optionally with the non-temporal hint (NT != 0): through the clang
// builtins on a 4 x 32-bit vector, which hipcc tracks in its own s_waitcnt bookkeeping (global_load_dwordx4 ... nt)
template <int NT, class V16>
__device__ __forceinline__ V16 fft_ld16(const V16* p) {
    static_assert(sizeof(V16) == 16, "one 16-byte lane access");
    if (NT) {
        const fft_u32x4 raw = __builtin_nontemporal_load(reinterpret_cast<const fft_u32x4*>(p));
        V16 v;
        __builtin_memcpy(&v, &raw, 16);
        return v;
    }
    return *p;
}
template <int NT, class V16>
__device__ __forceinline__ void fft_st16(V16* p, const V16& v) {
    static_assert(sizeof(V16) == 16, "one 16-byte lane access");
    if (NT) {
        fft_u32x4 raw;
        __builtin_memcpy(&raw, &v, 16);
        __builtin_nontemporal_store(raw, reinterpret_cast<fft_u32x4*>(p));
    } else {
        *p = v;
    }
}
// 16-byte store written through and NOT kept in the L2 (sc1: MI355X_MICROARCH.md, "stores of each flavour")
#define FFT_STORE16_SC1(ptr, v) fft_store16_sc1((ptr), (v))
template <class V16>
__device__ __forceinline__ void fft_store16_sc1(V16* ptr, const V16& v) {
    static_assert(sizeof(V16) == 16, "one 16-byte lane access");
    fft_u32x4 raw;
    __builtin_memcpy(&raw, &v, 16);
    asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(ptr), "v"(raw) : "memory");
}
// experiments: 16-byte store with arbitrary cache-policy bits (POL: 1 nt, 2 sc0 sc1 nt, 3 sc1 nt, 4 sc0 sc1, 5 sc0 nt)
template <int POL, class V16>
__device__ __forceinline__ void fft_store16_pol(V16* ptr, const V16& v) {
    fft_u32x4 raw;
    __builtin_memcpy(&raw, &v, 16);
    if (POL == 2) asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1 nt" ::"v"(ptr), "v"(raw) : "memory");
    else if (POL == 3) asm volatile("global_store_dwordx4 %0, %1, off sc1 nt" ::"v"(ptr), "v"(raw) : "memory");
    else if (POL == 4) asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" ::"v"(ptr), "v"(raw) : "memory");
    else if (POL == 5) asm volatile("global_store_dwordx4 %0, %1, off sc0 nt" ::"v"(ptr), "v"(raw) : "memory");
    else asm volatile("global_store_dwordx4 %0, %1, off nt" ::"v"(ptr), "v"(raw) : "memory");
}
// at most n of my memory operations still in flight (the counter has 6 bits: a larger n is clamped, which only waits for more)
#define FFT_WAIT_VM_LE(n) asm volatile("s_waitcnt vmcnt(%0)" ::"n"((n) > 63 ? 63 : (n)) : "memory")
#define FFT_WAVE_LOCKSTEP() ((void)0)
// One 8-byte LDS access that STAYS one ds_read_b64 / ds_write_b64: hipcc otherwise fuses pairs of them into ds_read2_b64 /
// ds_read2st64_b64 / ds_write2st64_b64, which move half the bytes per clock of the single forms and bank on 32 dwords in
// 16-lane groups instead of 64 dwords in 32-lane groups (MI355X_MICROARCH.md, LDS table) -- layouts made conflict-free for
// the single form are 2-way conflicted in the fused one.  Volatile accesses through an explicit LDS pointer are never fused
// (and, the address space being explicit, never come out as FLAT); hipcc still tracks them in its lgkmcnt bookkeeping.
#define FFT_LDS_LD64(p) fft_lds_ld64(p)
#define FFT_LDS_ST64(p, v) fft_lds_st64((p), (v))
template <int SC1>
__device__ __forceinline__ void fft_dma16(const void* gsrc, unsigned lane_lds_addr) {
#if defined(__HIP_DEVICE_COMPILE__)
    // M0 = LDS byte address of the wave's first lane; the hardware adds 16 * lane
    const unsigned lds_addr = __builtin_amdgcn_readfirstlane(lane_lds_addr);
    unsigned saved_m0;
    if (SC1 == 1)
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off sc1\n\ts_mov_b32 m0, %0"
                     : "=&s"(saved_m0) : "v"(gsrc), "s"(lds_addr) : "memory");
    else if (SC1 == 2)
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off nt\n\ts_mov_b32 m0, %0"
                     : "=&s"(saved_m0) : "v"(gsrc), "s"(lds_addr) : "memory");
    else if (SC1 == 4)  // system scope
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off sc0 sc1\n\ts_mov_b32 m0, %0"
                     : "=&s"(saved_m0) : "v"(gsrc), "s"(lds_addr) : "memory");
    else if (SC1 == 5)
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off sc0 sc1 nt\n\ts_mov_b32 m0, %0"
                     : "=&s"(saved_m0) : "v"(gsrc), "s"(lds_addr) : "memory");
    else if (SC1 == 6)
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off sc0 nt\n\ts_mov_b32 m0, %0"
                     : "=&s"(saved_m0) : "v"(gsrc), "s"(lds_addr) : "memory");
    else if (SC1 == 3)  // served by the shared L2, and the line is the first to leave it afterwards (read once)
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off sc1 nt\n\ts_mov_b32 m0, %0"
                     : "=&s"(saved_m0) : "v"(gsrc), "s"(lds_addr) : "memory");
    else
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                     : "=&s"(saved_m0) : "v"(gsrc), "s"(lds_addr) : "memory");
#else
    (void)gsrc;
    (void)lane_lds_addr;
#endif
}
#endif

// Experiment / ablation switches (environment variables read by the planner, ablation bits read by the kernels) exist
// only in builds made with -DFFT_EXPERIMENTS (libfft_mi355x_exp.so: tools/, the variant tests).  In the shipped
// library FFT_EXP_ENV() is a null constant and FFT_ABLATE() a zero constant, so every such branch folds away: the
// product cannot be told to skip work, and reads no environment variable per execute.
#if defined(FFT_EXPERIMENTS)
#include <stdlib.h>
#define FFT_EXP_ENV(name) getenv(name)
#define FFT_ABLATE(bits) (bits)
#else
#define FFT_EXP_ENV(name) ((const char*)0)
#define FFT_ABLATE(bits) 0
#endif

namespace fftk {

// Interleaved complex value: the device image of complex_t / complex32_t
// (reference include/fft_common.h:28 -- C99 `double complex` == double[2]).
template <typename T>
struct alignas(2 * sizeof(T)) cpx {  // 8 / 16-byte aligned: one ds_read_b64 / b128, one global_load_dwordx2 / x4
    T re, im;
};

template <typename T>
FFT_DEVICE cpx<T> mk(T a, T b) {
    cpx<T> r;
    r.re = a;
    r.im = b;
    return r;
}
template <typename T>
FFT_DEVICE cpx<T> cadd(cpx<T> a, cpx<T> b) { return mk<T>(a.re + b.re, a.im + b.im); }
template <typename T>
FFT_DEVICE cpx<T> csub(cpx<T> a, cpx<T> b) { return mk<T>(a.re - b.re, a.im - b.im); }
template <typename T>
FFT_DEVICE cpx<T> cmul(cpx<T> a, cpx<T> b) {
    return mk<T>(a.re * b.re - a.im * b.im, a.re * b.im + a.im * b.re);
}
template <typename T>
FFT_DEVICE cpx<T> cmul_conj(cpx<T> a, cpx<T> b) {  // a * conj(b)
    return mk<T>(a.re * b.re + a.im * b.im, a.im * b.re - a.re * b.im);
}
template <typename T>
FFT_DEVICE cpx<T> cscale(cpx<T> a, T s) { return mk<T>(a.re * s, a.im * s); }
template <typename T>
FFT_DEVICE cpx<T> mul_neg_i(cpx<T> a) { return mk<T>(a.im, -a.re); }  // a * (-i)
template <typename T>
FFT_DEVICE cpx<T> mul_pos_i(cpx<T> a) { return mk<T>(-a.im, a.re); }  // a * (+i)
template <typename T>
FFT_DEVICE cpx<T> cswap(cpx<T> a) { return mk<T>(a.im, a.re); }

// a + (-i) b  and  a - (-i) b: the L-shaped butterfly's odd outputs (fft_codelets.h) without a separate rotation
template <typename T>
FFT_DEVICE cpx<T> cadd_mni(cpx<T> a, cpx<T> b) { return mk<T>(a.re + b.im, a.im - b.re); }
template <typename T>
FFT_DEVICE cpx<T> csub_mni(cpx<T> a, cpx<T> b) { return mk<T>(a.re - b.im, a.im + b.re); }

#if !defined(FFT_EMU)
// fp32 on gfx950: a complex value is one 64-bit VGPR pair and the arithmetic below is PACKED math on that pair
// (v_pk_add_f32 / v_pk_mul_f32 / v_pk_fma_f32 with op_sel / neg modifiers doing the re<->im swizzles), written out
// because hipcc otherwise re-packs pairs of real parts and pairs of imaginary parts of DIFFERENT values and pays
// for it in v_mov shuffles (a quarter of the vector instructions of a radix-16 stage, measured).  The kernels are
// vector-ALU bound (2 waves per SIMD), so instruction count is time.  Non-template overloads: chosen over the
// templates above for cpx<float>.
typedef float fft_v2f __attribute__((ext_vector_type(2)));
// (built from / split into the two members: a bit cast goes through memory and keeps the kernels' register arrays
// from being promoted)
FFT_DEVICE fft_v2f as_v2(cpx<float> a) {
    fft_v2f v = {a.re, a.im};
    return v;
}
FFT_DEVICE cpx<float> as_cpx(fft_v2f v) {
    cpx<float> r;
    r.re = v.x;
    r.im = v.y;
    return r;
}
FFT_DEVICE cpx<float> cadd(cpx<float> a, cpx<float> b) { return as_cpx(as_v2(a) + as_v2(b)); }
FFT_DEVICE cpx<float> csub(cpx<float> a, cpx<float> b) { return as_cpx(as_v2(a) - as_v2(b)); }
FFT_DEVICE cpx<float> cscale(cpx<float> a, float s) { return as_cpx(as_v2(a) * s); }
FFT_DEVICE cpx<float> cmul(cpx<float> a, cpx<float> b) {
    fft_v2f t, r;  // t = (a.re b.re, a.re b.im);  r = (-a.im b.im + t.lo, a.im b.re + t.hi)
    asm("v_pk_mul_f32 %0, %1, %2 op_sel_hi:[0,1]" : "=v"(t) : "v"(as_v2(a)), "v"(as_v2(b)));
    asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_lo:[1,0,0]" : "=v"(r) : "v"(as_v2(a)), "v"(as_v2(b)), "v"(t));
    return as_cpx(r);
}
FFT_DEVICE cpx<float> cmul_conj(cpx<float> a, cpx<float> b) {  // a * conj(b)
    fft_v2f t, r;  // t = (a.re b.re, -a.re b.im);  r = (a.im b.im + t.lo, a.im b.re + t.hi)
    asm("v_pk_mul_f32 %0, %1, %2 op_sel_hi:[0,1] neg_hi:[0,1]" : "=v"(t) : "v"(as_v2(a)), "v"(as_v2(b)));
    asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,1,0] op_sel_hi:[1,0,1]" : "=v"(r) : "v"(as_v2(a)), "v"(as_v2(b)), "v"(t));
    return as_cpx(r);
}
FFT_DEVICE cpx<float> cadd_mni(cpx<float> a, cpx<float> b) {  // (a.re + b.im, a.im - b.re)
    fft_v2f r;
    asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0] neg_hi:[0,1]" : "=v"(r) : "v"(as_v2(a)), "v"(as_v2(b)));
    return as_cpx(r);
}
FFT_DEVICE cpx<float> csub_mni(cpx<float> a, cpx<float> b) {  // (a.re - b.im, a.im + b.re)
    fft_v2f r;
    asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0] neg_lo:[0,1]" : "=v"(r) : "v"(as_v2(a)), "v"(as_v2(b)));
    return as_cpx(r);
}
#endif

#if !defined(FFT_EMU)
typedef float fft_lds_v2f __attribute__((ext_vector_type(2)));
__device__ __forceinline__ cpx<float> fft_lds_ld64(const cpx<float>* p) {
    const fft_lds_v2f v = *(const volatile __attribute__((address_space(3))) fft_lds_v2f*)(const __attribute__((address_space(3))) void*)p;
    cpx<float> r;
    r.re = v.x;
    r.im = v.y;
    return r;
}
__device__ __forceinline__ void fft_lds_st64(cpx<float>* p, cpx<float> a) {
    fft_lds_v2f v = {a.re, a.im};
    *(volatile __attribute__((address_space(3))) fft_lds_v2f*)(__attribute__((address_space(3))) void*)p = v;
}
#endif

// One 16-byte lane access: 2 adjacent complex32 or 1 complex128.  Every HBM
// and LDS data access of the tile kernels moves one of these per lane
// (global_load_dwordx4 / ds_read_b128).
template <typename T>
struct alignas(16) vec16 {
    static constexpr int V = 16 / (int)sizeof(cpx<T>);
    cpx<T> c[V];
};

// V adjacent complex values moved by ONE LDS access of a Stockham stage (V * sizeof(cpx<T>) bytes: 8 or 16).
template <typename T, int V>
struct alignas(V * sizeof(cpx<T>)) lvec {
    cpx<T> c[V];
};

FFT_DEVICE unsigned bitrev32(unsigned v, int log2n) {
    // reverse the low log2n bits (reference include/fft_common.h:59-77, all log2n)
    v = ((v >> 1) & 0x55555555u) | ((v & 0x55555555u) << 1);
    v = ((v >> 2) & 0x33333333u) | ((v & 0x33333333u) << 2);
    v = ((v >> 4) & 0x0F0F0F0Fu) | ((v & 0x0F0F0F0Fu) << 4);
    v = ((v >> 8) & 0x00FF00FFu) | ((v & 0x00FF00FFu) << 8);
    v = (v >> 16) | (v << 16);
    return log2n > 0 ? (v >> (32 - log2n)) : 0u;
}

}  // namespace fftk
