// fft_wide_row.h -- wide_row_kernel: single-pass transforms of n = 8192 and 16384 fp32 and n = 8192 fp64 (one HBM round trip), the sizes just above
// what the tile kernels' 512 threads x 8 values reach (fft_kernels.h: n <= 4096).
//
// One 512-thread workgroup per CU walks the batch; a transform is ONE 64 KiB row: it lands in LDS by LDS-DMA (nt; the next row
// flies while this one is transformed), runs radix-16 x 16 x 16 x 2 Stockham stages with 16 values per thread (team_all_stages of
// fft_team.h: three LDS exchanges, twiddles by powers) and leaves as 16-byte non-temporal stores (lanes of adjacent frequencies pair
// up).  n = 16384: 1024 threads, radix-16 x 16 x 16 x 4, and ONE 128 KiB image that the stages run in place -- 160 KiB of LDS hold no
// second one, so the next row is requested only when the last stage has read the image (it lands under the result stores).
// Reference shape: the stage loop of algorithms/core/radix2_dit.c:84-112 with its intermediate stages held in LDS.
#pragma once

#include "fft_team.h"

namespace fftk {

template <typename T>
struct WideParams {
    const cpx<T>* in;
    cpx<T>* out;
    const cpx<T>* tables;  // [sa | sb]: W_L^m, m < 2^sa_bits; W_L^(m 2^sa_bits)
    int tables_bytes;
    int o_sb, sa_bits;
    int nb;
    int inverse;
    int nt;  // bit 0 loads, bit 1 stores non-temporal
    T scale;
};

#if defined(FFT_EMU)
#define FFT_WIDE_BOUNDS(LOG2L, E)
#else
#define FFT_WIDE_BOUNDS(LOG2L, E) __launch_bounds__((1 << (LOG2L)) / (E), ((1 << (LOG2L)) / (E)) / 256)
#endif

// E = values per thread = the radix of the stages (plus one stage of the remaining power of two)
template <typename T, int LOG2L, int E = 16>
FFT_KERNEL void FFT_WIDE_BOUNDS(LOG2L, E) wide_row_kernel(WideParams<T> p);

#if !defined(FFT_WIDE_DECL_ONLY)
template <typename T, int LOG2L, int E>
FFT_KERNEL void FFT_WIDE_BOUNDS(LOG2L, E) wide_row_kernel(WideParams<T> p) {
    constexpr int V16 = vec16<T>::V;  // values per 16-byte access: 2 (fp32), 1 (fp64)
    constexpr int L = 1 << LOG2L, NTHR = L / E, NCH = E / V16, log2TPC = LOG2L - Log2<E>::value;
    constexpr unsigned IMG = (unsigned)L * (unsigned)sizeof(cpx<T>);
    FFT_DYN_SMEM(smem);
    const int tid0 = FFT_TID;
    unsigned char* const land = smem;
    constexpr bool INPLACE = 2u * IMG > 140u * 1024u;  // two images do not fit next to the tables
    unsigned char* const work = INPLACE ? smem : smem + IMG;
    unsigned char* const tab_bytes = smem + (INPLACE ? 1 : 2) * (size_t)IMG;
    const unsigned land_lds = FFT_LDS_ADDR(land);
    {
        const vec16<T>* src = reinterpret_cast<const vec16<T>*>(p.tables);
        vec16<T>* dst = reinterpret_cast<vec16<T>*>(tab_bytes);
        for (int i = tid0; i < (p.tables_bytes >> 4); i += NTHR) dst[i] = src[i];
    }
    StageTw<T> tw;
    tw.sa = reinterpret_cast<const cpx<T>*>(tab_bytes);
    tw.sb = tw.sa + p.o_sb;
    tw.sa_bits = p.sa_bits;
    tw.log2L = LOG2L;
    auto dma_row = [&](long long b) __attribute__((always_inline)) {
        int tid = tid0;
        FFT_OPAQUE(tid);
        const cpx<T>* src = p.in + b * L + V16 * tid;
        if (p.nt & 1) {
            FFT_UNROLL
            for (int i = 0; i < NCH; i++) FFT_DMA16_NT(src + i * V16 * NTHR, land, land_lds, (unsigned)(i * NTHR + tid) * 16u);
        } else {
            FFT_UNROLL
            for (int i = 0; i < NCH; i++) FFT_DMA16(src + i * V16 * NTHR, land, land_lds, (unsigned)(i * NTHR + tid) * 16u);
        }
    };
    const long long stride = FFT_NBLOCKS;
    long long b = FFT_BID;
    if (b < p.nb) dma_row(b);
    bool first = true;
    FFT_SYNC();  // the tables are in LDS
    for (; b < p.nb; b += stride, first = false) {
        // the row has landed (what may still fly are the previous row's NCH result stores, issued behind this row's DMA)
        if (first) FFT_WAIT_VM0();
        else FFT_WAIT_VM_LE(NCH);
        FFT_SYNC_LDS();
        int tid = tid0;
        FFT_OPAQUE(tid);
        cpx<T> x[1][E][1];
        team_all_stages<T, E, (sizeof(T) == 4)>(x, land, work, tw, tid, 0, 0, log2TPC, LOG2L, [&](int s, int) {
            if (!INPLACE && s == 0 && b + stride < p.nb) dma_row(b + stride);  // the landing image is free: the next row flies under the stages
        }, p.inverse != 0);
        if (INPLACE && b + stride < p.nb) {
            FFT_SYNC_LDS();  // everybody has read the last stage's inputs
            dma_row(b + stride);
        }
        if (p.inverse) {
            FFT_UNROLL
            for (int e = 0; e < E; e++) x[0][e][0] = cswap(x[0][e][0]);
        }
        if (p.scale != (T)1) {
            FFT_UNROLL
            for (int e = 0; e < E; e++) x[0][e][0] = cscale(x[0][e][0], p.scale);
        }
        // slot e holds X[tid + (L / E) e].  fp32: the lanes of frequencies tid, tid ^ 1 pair up for 16-byte stores
        if constexpr (V16 == 2) {
            const bool odd = (tid & 1) != 0;
            cpx<T>* const dst0 = p.out + b * L + (tid & ~1);
            FFT_UNROLL
            for (int q = 0; q < E / 2; q++) {
                vec16<T> v;
                pair_rows<T>(x[0][2 * q][0], x[0][2 * q + 1][0], odd, 1, v);
                vec16<T>* const dst = reinterpret_cast<vec16<T>*>(dst0 + ((2 * q + (odd ? 1 : 0)) << log2TPC));
                if (p.nt & 2) FFT_STORE16_NT(dst, v);
                else *dst = v;
            }
        } else {
            cpx<T>* const dst0 = p.out + b * L + tid;
            FFT_UNROLL
            for (int e = 0; e < E; e++) {
                vec16<T> v;
                v.c[0] = x[0][e][0];
                vec16<T>* const dst = reinterpret_cast<vec16<T>*>(dst0 + (e << log2TPC));
                if (p.nt & 2) FFT_STORE16_NT(dst, v);
                else *dst = v;
            }
        }
    }
}
#endif

}  // namespace fftk
