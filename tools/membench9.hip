// membench9.hip -- the hand-off chain of the team exchange, priced by itself: how long is one round of "32 workgroups of one XCD write
// 64 KiB each into the other 31's images in the XCD's L2, tell them, wait, pull the own image in by LDS-DMA, say so" when the telling is
//   mode 0: the shipped protocol's -- ONE arrival counter per team, polled by every seat (all-to-all), whole 2 MiB rounds, K slots;
//   mode 2: per-receiver counters -- every WAVE signals the 8 receivers it feeds as soon as ITS stores are complete (no workgroup barrier, one
//           atomic instruction of 8 lanes), a receiver waits for its own 64 senders only and acknowledges to the 64 sender waves (one atomic
//           instruction of 64 lanes); the window is a ring of K HALF rounds (1 MiB: the images of 16 receivers), so K = 3 is the 3 MiB
//           window that profiles/r4_membench8_window_sizes.txt prices at 1.9 ms of memory pace instead of 2.4.  Senders may run up to K / 2
//           rounds ahead of the slowest one, so every counter exists four times, by the round mod 4 (K <= 6), in one 128-byte line.
// (measurement tool, round 4: DESIGN 4.3c.9 -- the 4 MiB window is what the memory side charges for, and a smaller one needs a hand-off
// chain below 4.4 us.  This tool measures the chains.)  Every landed image is CHECKED against the round's tag, so a protocol that lets a
// sender overwrite an image before its receiver has read it shows as errors, not as a good time.  hbm = 1 adds the transform's HBM streams
// (64 KiB nt LDS-DMA in, 64 KiB nt stores out per workgroup and round): the traffic of team_quad_kernel at n = 2^20 with nothing computed.
// Every wait is bounded (timeout -> status bit 2, the launch drains).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e_), __LINE__); exit(1);} } while (0)
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

#define PADW 32  // words between two counters: a 128-byte line each
enum { C_REG = 0, C_STATUS = 1, C_ERR = 2, C_XCNT = 32, C_TEAM = 1024, TEAM_WORDS = 32768,
       T_ARR = 0, T_LAND = PADW, T_RCV = 2 * PADW, T_SND = 2 * PADW + 32 * PADW };
#define HALF_PIECES 65536u  // 1 MiB of 16-byte pieces
#define TEAM_PIECES (8u * HALF_PIECES)

__device__ __forceinline__ void st16(u32x4* p, u32x4 v) { asm volatile("global_store_dwordx4 %0, %1, off" ::"v"(p), "v"(v) : "memory"); }
__device__ __forceinline__ void st16_nt(u32x4* p, u32x4 v) { asm volatile("global_store_dwordx4 %0, %1, off nt" ::"v"(p), "v"(v) : "memory"); }
__device__ __forceinline__ void dma16(const u32x4* p, unsigned lds, bool nt) {
    const unsigned a = __builtin_amdgcn_readfirstlane(lds);
    unsigned saved;
    if (nt) asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off nt\n\ts_mov_b32 m0, %0" : "=&s"(saved) : "v"(p), "s"(a) : "memory");
    else asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off sc1\n\ts_mov_b32 m0, %0" : "=&s"(saved) : "v"(p), "s"(a) : "memory");
}
__device__ __forceinline__ unsigned sload_glc(const unsigned* pv) {
    const unsigned long long a = (unsigned long long)pv;  // wave-uniform by construction: tell the compiler
    const unsigned hi = (unsigned)__builtin_amdgcn_readfirstlane((int)(a >> 32)), lo = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)a);
    const unsigned* p = (const unsigned*)(((unsigned long long)hi << 32) | (unsigned long long)lo);
    unsigned v;
    asm volatile("s_load_dword %0, %1, 0x0 glc\n\ts_waitcnt lgkmcnt(0)" : "=s"(v) : "s"(p) : "memory");
    return v;
}
__device__ __forceinline__ void count_add(unsigned* p) { asm volatile("global_atomic_add %0, %1, off" ::"v"(p), "v"(1u) : "memory"); }
__device__ __forceinline__ void vm0() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
__device__ __forceinline__ long long now() { return (long long)wall_clock64(); }

template <int MODE>
__global__ __launch_bounds__(512) void k_chain(unsigned* ctl, u32x4* win, const u32x4* hin, u32x4* hout, int rounds, int K, int hbm,
                                               int work_ticks, long long timeout, long long* clocks) {
    extern __shared__ u32x4 lds[];  // image 0: the window image, image 1: the HBM chunk
    __shared__ unsigned sh[4];
    const unsigned tid = threadIdx.x, w = tid >> 6, lane = tid & 63;
    unsigned x;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(x));
    x &= 15u;
    if (tid == 0) {
        const unsigned seat = __hip_atomic_fetch_add(&ctl[C_XCNT + 32 * x], 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_fetch_add(&ctl[C_REG], 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
        const long long t0 = now();
        unsigned ok = 1;
        while (__hip_atomic_load(&ctl[C_REG], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) != gridDim.x) {
            if (now() - t0 > timeout) { ok = 0; break; }
            __builtin_amdgcn_s_sleep(8);
        }
        if (ok && __hip_atomic_load(&ctl[C_XCNT + 32 * x], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) != 32u) ok = 0;
        if (!ok) atomicOr(&ctl[C_STATUS], 1u);
        sh[0] = seat; sh[1] = ok; sh[3] = 0;
    }
    __syncthreads();
    if (!sh[1]) return;
    const unsigned s = sh[0];
    unsigned* const T = ctl + C_TEAM + x * TEAM_WORDS;
    u32x4* const W = win + (size_t)x * TEAM_PIECES;
    const unsigned lds0 = (unsigned)(size_t)lds;
    bool dead = false;
    unsigned errs = 0;

    auto wait_team = [&](unsigned* ptr, unsigned target) {  // first thread polls, the workgroup waits at the barrier
        if (tid == 0 && !*(volatile unsigned*)&sh[3]) {
            const long long t0 = now();
            for (;;) {
                if ((int)(sload_glc(ptr) - target) >= 0) break;
                if (now() - t0 > timeout) { atomicOr(&ctl[C_STATUS], 2u); sh[3] = 1; break; }
                __builtin_amdgcn_s_sleep(2);
            }
        }
        __syncthreads();
    };
    auto wait_wave = [&](unsigned* ptr, unsigned target) {  // every wave for itself
        if (dead) return;
        const long long t0 = now();
        for (;;) {
            if ((int)(sload_glc(ptr) - target) >= 0) break;
            if (now() - t0 > timeout) { atomicOr(&ctl[C_STATUS], 2u); dead = true; break; }
            __builtin_amdgcn_s_sleep(2);
        }
    };
    const unsigned my_h = (s >> 2) & 1u, my_j = (s >> 3) * 4u + (s & 3u);  // mode 2: my image's half and place in the half round

    if (tid == 0) clocks[blockIdx.x * 2] = now();
    for (int r = 0; r < rounds; r++) {
        const u32x4 val = {(unsigned)r + 1u, s, w, lane};
        // ---- send: my 64 KiB of the round, 2 KiB to every seat of the team
        if (MODE == 0) {
            if (r >= K) wait_team(T + T_LAND, 32u * (unsigned)(r - K + 1));  // the slot's previous round has landed everywhere
#pragma unroll
            for (unsigned i = 0; i < 8; i++) {
                const unsigned d = (w & 3u) * 8u + i;
                st16(W + (size_t)(r % K) * (2u * HALF_PIECES) + d * 4096u + (s * 2u + (w >> 2)) * 64u + lane, val);
            }
            vm0();
            __syncthreads();
            if (tid == 0) count_add(T + T_ARR);
        } else {
#pragma unroll
            for (unsigned h = 0; h < 2; h++) {
                const int u = 2 * r + (int)h;
                if (u >= K) {  // the half round that had this piece of the ring: its four receivers of mine have read it
                    const int up = u - K;
                    const int rp = up >> 1;
                    wait_wave(T + T_SND + PADW * ((s * 8u + w) * 2u + (unsigned)(up & 1)) + (unsigned)(rp & 3), 4u * (unsigned)((rp >> 2) + 1));
                }
#pragma unroll
                for (unsigned i2 = 0; i2 < 4; i2++)
                    st16(W + (size_t)(u % K) * HALF_PIECES + ((w & 3u) * 4u + i2) * 4096u + (s * 2u + (w >> 2)) * 64u + lane, val);
            }
            vm0();
            if (lane < 8) {  // receiver d = 8 (w & 3) + i, i = 4 h + i2
                count_add(T + T_RCV + PADW * ((w & 3u) * 8u + lane) + (unsigned)(r & 3));
            }
        }
        // ---- the transform's HBM streams, free-running: results of the previous round out, the next chunk in
        if (hbm) {
            u32x4* dst = hout + ((size_t)blockIdx.x * rounds + r) * 4096u + tid;
#pragma unroll
            for (int i = 0; i < 8; i++) st16_nt(dst + i * 512, val);
            const u32x4* src = hin + ((size_t)blockIdx.x * rounds + r) * 4096u + tid;
#pragma unroll
            for (int i = 0; i < 8; i++) dma16(src + i * 512, lds0 + 65536u + (unsigned)(i * 512 + (tid & ~63u)) * 16u, true);
        }
        if (work_ticks) {
            const long long t0 = now();
            while (now() - t0 < work_ticks) __builtin_amdgcn_s_sleep(4);
        }
        // ---- receive: my image of the round
        const u32x4* img;
        if (MODE == 0) {
            wait_team(T + T_ARR, 32u * (unsigned)(r + 1));
            img = W + (size_t)(r % K) * (2u * HALF_PIECES) + s * 4096u;
        } else {
            if (w == 0) wait_wave(T + T_RCV + PADW * s + (unsigned)(r & 3), 64u * (unsigned)((r >> 2) + 1));
            __syncthreads();
            img = W + (size_t)((2 * r + (int)my_h) % K) * HALF_PIECES + my_j * 4096u;
        }
#pragma unroll
        for (int i = 0; i < 8; i++) dma16(img + i * 512 + tid, lds0 + (unsigned)(i * 512 + (tid & ~63u)) * 16u, false);
        vm0();
        __syncthreads();
        if (MODE == 0) {
            if (tid == 0) count_add(T + T_LAND);
        } else if (w == 0) {  // to the 64 sender waves of my image: seat lane / 2, wave (s >> 3) + 4 (lane & 1)
            count_add(T + T_SND + PADW * (((lane >> 1) * 8u + (s >> 3) + 4u * (lane & 1u)) * 2u + my_h) + (unsigned)(r & 3));
        }
#pragma unroll
        for (int i = 0; i < 8; i++) errs += (lds[i * 512 + tid].x != (unsigned)r + 1u);
    }
    vm0();
    __syncthreads();
    if (tid == 0) clocks[blockIdx.x * 2 + 1] = now();
    if (errs && !dead && !sh[3]) atomicAdd(&ctl[C_ERR], errs);
}

// The same hand-offs, PIPELINED as a transform kernel would run them: the sends of round r + 1 go out in front of the receive of round r
// wherever the ring has room (half h of round r + 1 reuses the ring piece of unit 2 (r + 1) + h - K: free before round r is read iff
// K > 2 + h; otherwise the half is sent behind the own receive, when its four receivers' acknowledgements of round r are in), signals
// ride on waits that are there anyway, and the HBM streams are issued first in every round so that they fly under the hand-off.
//   MODE 0: team counters, 2 slots of a whole round (4 MiB) -- the shipped protocol
//   MODE 2: per-receiver counters, ring of K half rounds, a signal per half
template <int MODE>
__global__ __launch_bounds__(512) void k_pipe(unsigned* ctl, u32x4* win, const u32x4* hin, u32x4* hout, int rounds, int K, int hbm,
                                              int work_ticks, long long timeout, long long* clocks) {
    extern __shared__ u32x4 lds[];
    __shared__ unsigned sh[4];
    const unsigned tid = threadIdx.x, w = tid >> 6, lane = tid & 63;
    unsigned x;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(x));
    x &= 15u;
    if (tid == 0) {
        const unsigned seat = __hip_atomic_fetch_add(&ctl[C_XCNT + 32 * x], 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_fetch_add(&ctl[C_REG], 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
        const long long t0 = now();
        unsigned ok = 1;
        while (__hip_atomic_load(&ctl[C_REG], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) != gridDim.x) {
            if (now() - t0 > timeout) { ok = 0; break; }
            __builtin_amdgcn_s_sleep(8);
        }
        if (ok && __hip_atomic_load(&ctl[C_XCNT + 32 * x], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) != 32u) ok = 0;
        if (!ok) atomicOr(&ctl[C_STATUS], 1u);
        sh[0] = seat; sh[1] = ok; sh[3] = 0;
    }
    __syncthreads();
    if (!sh[1]) return;
    const unsigned s = sh[0];
    unsigned* const T = ctl + C_TEAM + x * TEAM_WORDS;
    u32x4* const W = win + (size_t)x * TEAM_PIECES;
    const unsigned lds0 = (unsigned)(size_t)lds;
    bool dead = false;
    unsigned errs = 0;
    auto wait_team = [&](unsigned* ptr, unsigned target) {
        if (tid == 0 && !*(volatile unsigned*)&sh[3]) {
            const long long t0 = now();
            for (;;) {
                if ((int)(sload_glc(ptr) - target) >= 0) break;
                if (now() - t0 > timeout) { atomicOr(&ctl[C_STATUS], 2u); sh[3] = 1; break; }
                __builtin_amdgcn_s_sleep(2);
            }
        }
        __syncthreads();
    };
    auto wait_wave = [&](unsigned* ptr, unsigned target) {
        if (dead) return;
        const long long t0 = now();
        for (;;) {
            if ((int)(sload_glc(ptr) - target) >= 0) break;
            if (now() - t0 > timeout) { atomicOr(&ctl[C_STATUS], 2u); dead = true; break; }
            __builtin_amdgcn_s_sleep(2);
        }
    };
    const unsigned my_h = (s >> 2) & 1u, my_j = (s >> 3) * 4u + (s & 3u);
    auto send_half = [&](int r1, unsigned h) {  // mode 2: the four stores of half h of round r1, behind the ring's guard
        const int u = 2 * r1 + (int)h;
        if (u >= K) {
            const int up = u - K, rp = up >> 1;
            wait_wave(T + T_SND + PADW * ((s * 8u + w) * 2u + (unsigned)(up & 1)) + (unsigned)(rp & 3), 4u * (unsigned)((rp >> 2) + 1));
        }
        const u32x4 val = {(unsigned)r1 + 1u, s, w, lane};
#pragma unroll
        for (unsigned i2 = 0; i2 < 4; i2++)
            st16(W + (size_t)(u % K) * HALF_PIECES + ((w & 3u) * 4u + i2) * 4096u + (s * 2u + (w >> 2)) * 64u + lane, val);
    };
    auto signal_half = [&](int r1, unsigned h) {  // the half's stores are complete
        if (lane < 4) count_add(T + T_RCV + PADW * ((w & 3u) * 8u + 4u * h + lane) + (unsigned)(r1 & 3));
    };
    auto send_round = [&](int r1) {  // mode 0
        const u32x4 val = {(unsigned)r1 + 1u, s, w, lane};
#pragma unroll
        for (unsigned i = 0; i < 8; i++)
            st16(W + (size_t)(r1 & 1) * (2u * HALF_PIECES) + ((w & 3u) * 8u + i) * 4096u + (s * 2u + (w >> 2)) * 64u + lane, val);
    };
    const bool pre0 = K > 2, pre1 = K > 3;  // half h of the next round goes out in front of this round's receive

    if (tid == 0) clocks[blockIdx.x * 2] = now();
    if (MODE == 0) {
        send_round(0);
        vm0();
        __syncthreads();
        if (tid == 0) count_add(T + T_ARR);
    } else {
        send_half(0, 0);
        send_half(0, 1);
        vm0();
        signal_half(0, 0);
        signal_half(0, 1);
    }
    for (int r = 0; r < rounds; r++) {
        const bool more = r + 1 < rounds;
        if (hbm) {
            const u32x4 val = {(unsigned)r + 1u, s, w, lane};
            u32x4* dst = hout + ((size_t)blockIdx.x * rounds + r) * 4096u + tid;
#pragma unroll
            for (int i = 0; i < 8; i++) st16_nt(dst + i * 512, val);
            const u32x4* src = hin + ((size_t)blockIdx.x * rounds + r) * 4096u + tid;
#pragma unroll
            for (int i = 0; i < 8; i++) dma16(src + i * 512, lds0 + 65536u + (unsigned)(i * 512 + (tid & ~63u)) * 16u, true);
        }
        if (MODE == 2 && r > 0) {  // the halves sent behind the previous receive: complete once only this round's 16 HBM requests fly
            if (!pre0 || !pre1) {
                if (hbm) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
                else vm0();
                if (!pre0) signal_half(r, 0);
                if (!pre1) signal_half(r, 1);
            }
        }
        if (work_ticks) {
            const long long t0 = now();
            while (now() - t0 < work_ticks) __builtin_amdgcn_s_sleep(4);
        }
        const u32x4* img;
        if (MODE == 0) {
            if (more) {
                if (r >= 1) wait_team(T + T_LAND, 32u * (unsigned)r);  // round r - 1 has landed everywhere: its slot is free
                send_round(r + 1);
            }
            wait_team(T + T_ARR, 32u * (unsigned)(r + 1));
            img = W + (size_t)(r & 1) * (2u * HALF_PIECES) + s * 4096u;
        } else {
            if (more && pre0) send_half(r + 1, 0);
            if (more && pre1) send_half(r + 1, 1);
            if (w == 0) wait_wave(T + T_RCV + PADW * s + (unsigned)(r & 3), 64u * (unsigned)((r >> 2) + 1));
            __syncthreads();
            img = W + (size_t)((2 * r + (int)my_h) % K) * HALF_PIECES + my_j * 4096u;
        }
#pragma unroll
        for (int i = 0; i < 8; i++) dma16(img + i * 512 + tid, lds0 + (unsigned)(i * 512 + (tid & ~63u)) * 16u, false);
        vm0();
        if (MODE == 2 && more) {
            if (pre0) signal_half(r + 1, 0);
            if (pre1) signal_half(r + 1, 1);
        }
        __syncthreads();
        if (MODE == 0) {
            if (tid == 0) {
                if (more) count_add(T + T_ARR);
                count_add(T + T_LAND);
            }
        } else {
            if (w == 0) count_add(T + T_SND + PADW * (((lane >> 1) * 8u + (s >> 3) + 4u * (lane & 1u)) * 2u + my_h) + (unsigned)(r & 3));
            if (more && !pre0) send_half(r + 1, 0);
            if (more && !pre1) send_half(r + 1, 1);
        }
#pragma unroll
        for (int i = 0; i < 8; i++) errs += (lds[i * 512 + tid].x != (unsigned)r + 1u);
    }
    vm0();
    __syncthreads();
    if (tid == 0) clocks[blockIdx.x * 2 + 1] = now();
    if (errs && !dead && !sh[3]) atomicAdd(&ctl[C_ERR], errs);
}

int main(int argc, char** argv) {
    const int nwg = 256;
    const int rounds = argc > 1 ? atoi(argv[1]) : 256;  // 4 rounds x 64 transforms per team = n 2^20 x 512
    unsigned* ctl; u32x4 *win, *hin, *hout; long long* clocks;
    const size_t ctl_bytes = (size_t)(C_TEAM + 16 * TEAM_WORDS) * 4, hbytes = (size_t)nwg * rounds * 65536;
    CK(hipMalloc(&ctl, ctl_bytes));
    CK(hipMalloc(&win, (size_t)16 * TEAM_PIECES * 16));
    CK(hipMalloc(&hin, hbytes));
    CK(hipMalloc(&hout, hbytes));
    CK(hipMemset(hin, 1, hbytes));
    CK(hipMemset(hout, 0, hbytes));
    CK(hipMemset(win, 0, (size_t)16 * TEAM_PIECES * 16));
    CK(hipMalloc(&clocks, nwg * 2 * sizeof(long long)));
    CK(hipFuncSetAttribute((const void*)k_chain<0>, hipFuncAttributeMaxDynamicSharedMemorySize, 131072));
    CK(hipFuncSetAttribute((const void*)k_chain<2>, hipFuncAttributeMaxDynamicSharedMemorySize, 131072));
    CK(hipFuncSetAttribute((const void*)k_pipe<0>, hipFuncAttributeMaxDynamicSharedMemorySize, 131072));
    CK(hipFuncSetAttribute((const void*)k_pipe<2>, hipFuncAttributeMaxDynamicSharedMemorySize, 131072));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    struct V { int mode, K; const char* name; };
    const V vs[] = {{0, 1, "team counter, 1 slot  (2 MiB)"}, {0, 2, "team counter, 2 slots (4 MiB)"},
                    {2, 2, "per receiver, ring 2  (2 MiB)"}, {2, 3, "per receiver, ring 3  (3 MiB)"},
                    {2, 4, "per receiver, ring 4  (4 MiB)"}, {2, 6, "per receiver, ring 6  (6 MiB)"},
                    {10, 2, "PIPELINED team counter, 2 slots (4 MiB)"}, {12, 2, "PIPELINED per receiver, ring 2 (2 MiB)"},
                    {12, 3, "PIPELINED per receiver, ring 3 (3 MiB)"}, {12, 4, "PIPELINED per receiver, ring 4 (4 MiB)"}};
    static long long h[nwg * 2];
    printf("rounds %d (64 KiB per workgroup and round into the window and back); period = in-kernel time / rounds\n", rounds);
    fflush(stdout);
    for (int hbm = 0; hbm < 2; hbm++)
        for (int work : {0, 100, 200})
            for (const V& v : vs) {
                float best = 1e30f;
                double period = 0;
                unsigned st[4] = {0, 0, 0, 0}, worst_st = 0, worst_err = 0;
                for (int rep = 0; rep < 3; rep++) {
                    CK(hipMemset(ctl, 0, ctl_bytes));
                    CK(hipMemset(clocks, 0, nwg * 2 * sizeof(long long)));
                    CK(hipEventRecord(e0));
                    if (v.mode == 10) hipLaunchKernelGGL(k_pipe<0>, dim3(nwg), dim3(512), 131072, 0, ctl, win, hin, hout, rounds, v.K, hbm, work, 500000LL, clocks);
                    else if (v.mode == 12) hipLaunchKernelGGL(k_pipe<2>, dim3(nwg), dim3(512), 131072, 0, ctl, win, hin, hout, rounds, v.K, hbm, work, 500000LL, clocks);
                    else if (v.mode == 0) hipLaunchKernelGGL(k_chain<0>, dim3(nwg), dim3(512), 131072, 0, ctl, win, hin, hout, rounds, v.K, hbm, work, 500000LL, clocks);
                    else hipLaunchKernelGGL(k_chain<2>, dim3(nwg), dim3(512), 131072, 0, ctl, win, hin, hout, rounds, v.K, hbm, work, 500000LL, clocks);
                    CK(hipEventRecord(e1));
                    CK(hipEventSynchronize(e1));
                    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
                    CK(hipMemcpy(st, ctl, sizeof(st), hipMemcpyDeviceToHost));
                    worst_st |= st[C_STATUS];
                    worst_err += st[C_ERR];
                    if (rep > 0 && ms < best) {
                        best = ms;
                        CK(hipMemcpy(h, clocks, sizeof(h), hipMemcpyDeviceToHost));
                        long long first = 0, last = 0; int live = 0;
                        for (int b = 0; b < nwg; b++) if (h[2 * b + 1]) { if (!live || h[2 * b] < first) first = h[2 * b]; if (h[2 * b + 1] > last) last = h[2 * b + 1]; live++; }
                        period = (double)(last - first) / 100.0 / rounds;
                    }
                }
                printf("hbm %d work %3.1f us  %-40s  period %6.3f us  launch %7.3f ms  status %u errors %u\n", hbm, work / 100.0, v.name, period, best, worst_st, worst_err);
                fflush(stdout);
            }
    return 0;
}
