// fft_team.h -- team_fft_kernel: a whole transform per XCD, ONE HBM round trip.
//
// The two-pass (four-step) plan of fft_engine.h moves every element through HBM
// twice: pass A writes the n intermediate values to a scratch image, pass B reads
// them back.  That caps the algorithmic bandwidth at half of what the memory
// system moves.  This kernel keeps the intermediate ON the XCD:
//
//   * A "team" is the TS workgroups (one per CU, TS = 32 on MI355X) that run on
//     ONE XCD and therefore share one 4 MiB L2.  Teams are formed at run time
//     from HW_REG_XCC_ID -- never from blockIdx -- so the placement is verified,
//     not assumed; if the launch does not yield n_teams full teams the kernel
//     writes a status word and exits, and the two-pass plan queued behind it (it
//     reads the same status word) does the work instead.
//   * A team transforms one length-n signal at a time, n = L1 * L2 (the four-step
//     split of optimizations/parallel_fft.c:213-272 in the reference):
//       step A  each workgroup loads NT column tiles (L1 rows x CA columns,
//               L2-strided) from HBM straight into LDS (LDS-DMA, no staging
//               registers: the next tile lands while this one is transformed),
//               runs the length-L1 Stockham FFTs in LDS, applies W_n^(k1 n2) and
//               KEEPS the results in registers
//               (NT * 16 complex values per thread: the team's register files
//               hold the whole intermediate, 8 MiB for n = 2^20 fp32);
//       step B  in NT phases the team transposes the intermediate through its
//               L2: every workgroup stores 1/NT of its registers into a scratch
//               window (2 x 2 MiB per team, rewritten every phase, so it stays
//               dirty in L2 and never needs to reach HBM), a team barrier, then
//               every workgroup pulls the CB rows it owns into LDS (L1-bypassing
//               sc1 LDS-DMA, served by the shared L2), runs the length-L2 FFTs and
//               stores the result transposed => natural order in HBM.
//   * Same-XCD visibility needs no cache maintenance: a plain store is in the L2
//     once vmcnt says so (the vector L1 is write-through), an sc1 load bypasses the
//     reader's L1.  The team barrier is a line of 32 generation words, one per
//     member, stored plain and polled with sc1 loads.  Every spin is bounded.
//
// HBM traffic per transform: n elements in, n elements out -- the algorithmic
// minimum; the transposition traffic stays inside the XCD.
#pragma once

#include "fft_kernels.h"

namespace fftk {

// control block (32-bit words), zeroed before every launch
enum {
    TEAM_CTL_REGISTERED = 0,          // workgroups that have registered
    TEAM_CTL_STATUS = 1,              // 0 = done by this kernel; 1 = teams could not be formed (nothing touched); 2 = barrier timeout
    TEAM_CTL_COUNT = 32,              // + 32 * xcc : workgroups registered on that XCD (own 128-byte line each)
    TEAM_CTL_FLAGS = 32 + 32 * 16,    // + 32 * xcc : the team's barrier line, one generation word per member
    TEAM_CTL_WORDS = 32 + 32 * 16 + 32 * 16
};
enum { TEAM_STATUS_OK = 0, TEAM_STATUS_NO_TEAMS = 1, TEAM_STATUS_TIMEOUT = 2 };

template <typename T>
struct TeamParams {
    const cpx<T>* in;
    cpx<T>* out;
    const cpx<T>* tables;    // blob [sa1 | sb1 | sa2 | sb2 | t0 | t1]
    unsigned char* scratch;  // n_teams windows of 2 * (tile_bytes << log2TS) bytes
    unsigned* ctl;
    int tables_bytes;
    int data_bytes;          // LDS bytes of the data region (stage exchange / row staging image); tables follow
    int log2L1, log2L2, log2CA, log2CB, log2TS;
    int n_teams;
    int nb;
    int inverse;
    int o_sb1, o_sa2, o_sb2, o_t0, o_t1;
    int sa1_bits, sa2_bits, t0_bits;
    long long timeout_ticks;  // bound of every spin, in FFT_CLOCK ticks
    int ablate;               // experiments: 1 skip the inter-pass twiddle, 2 skip the stages
    long long* trace;         // profiling: not NULL = every workgroup logs FFT_CLOCK at its first trace_events events
    int trace_events;
    T scale;
};

// Has every member of the team stored generation `gen`?  On the device the first wavefront polls the team's
// 128-byte flag line with ONE sc1 load (lane m reads member m's word); the emulation polls from one thread.
#if defined(FFT_EMU)
#define FFT_TEAM_POLL_LANES 1
FFT_DEVICE bool team_all_arrived(unsigned* flags, int TS, unsigned gen, int /*lane*/) {
    for (int m = 0; m < TS; m++)
        if ((int)(FFT_L2_FLAG_LOAD(&flags[m]) - gen) < 0) return false;
    return true;
}
#else
#define FFT_TEAM_POLL_LANES 64
FFT_DEVICE bool team_all_arrived(unsigned* flags, int TS, unsigned gen, int lane) {
    unsigned f = gen;
    if (lane < TS) f = FFT_L2_FLAG_LOAD(&flags[lane]);
    return __all((int)(f - gen) >= 0) != 0;
}
#endif

#define FFT_TEAM_GEO(l1, l2, ca, cb, ts) ((l1) | ((l2) << 5) | ((ca) << 10) | ((cb) << 15) | ((ts) << 20))

// All Stockham stages (radix 4, plus one radix-2 stage when log2L is odd) of one tile whose samples sit in the
// LDS-DMA landing image `land` ([element][column], the stage layout): the first stage reads `land` and writes the
// work image `work`, the others run in `work`.  `land_is_free` runs as soon as every wave has read `land`.
template <typename T, int E, int V, class Hook>
FFT_DEVICE void team_all_stages(cpx<T> (&x)[1][E][V], const unsigned char* land, unsigned char* work, const StageTw<T>& tw,
                                int r, int j, int log2J, int log2TPC, int log2L, Hook&& land_is_free, bool swap_in) {
    int log2Lprev = log2L, log2P = 0;
    const int n_full = log2L >> 1;
    const int rem = log2L & 1;
    FFT_UNROLL
    for (int s = 0; s < n_full; s++) {
        if (s == 0)
            stockham_stage_rw<T, E, 4, V, 1>(x, land, work, 0, tw, r, j, log2J, log2TPC, log2Lprev, log2P, false,
                                             n_full + rem == 1, land_is_free, swap_in);
        else
            stockham_stage_rw<T, E, 4, V, 1>(x, work, work, 0, tw, r, j, log2J, log2TPC, log2Lprev, log2P, false,
                                             s == n_full + rem - 1, StageNoHook());
    }
    if (rem) stockham_stage_rw<T, E, 2, V, 1>(x, work, work, 0, tw, r, j, log2J, log2TPC, log2Lprev, log2P, false, true, StageNoHook());
}

// NT = tiles per workgroup per step = n / (TS * tile elements); GEO != 0 bakes the geometry into the instantiation.
template <typename T, int NT, int GEO>
FFT_KERNEL void FFT_LAUNCH_BOUNDS2(512, 2) team_fft_kernel(TeamParams<T> p) {
    constexpr int E = 8;
    constexpr int V = vec16<T>::V;
    constexpr int log2V = Log2<V>::value;
    constexpr int SZ = (int)sizeof(cpx<T>);
    constexpr int EP = E / NT;  // register slots (rows r + TPC*e) that one phase hands over
    constexpr int log2NT = Log2<NT>::value;
    static_assert(NT == 1 || NT == 2 || NT == 4, "the team's register files hold at most 4 tiles per workgroup");
    FFT_DYN_SMEM(smem);

    const int tid_invariant = FFT_TID;
    const int nthreads = FFT_NTHREADS;
    const int log2L1 = GEO ? (GEO & 31) : p.log2L1;
    const int log2L2 = GEO ? ((GEO >> 5) & 31) : p.log2L2;
    const int log2CA = GEO ? ((GEO >> 10) & 31) : p.log2CA;
    const int log2CB = GEO ? ((GEO >> 15) & 31) : p.log2CB;
    const int log2TS = GEO ? ((GEO >> 20) & 31) : p.log2TS;
    const int TS = 1 << log2TS;
    // step A: thread (jA, rA) owns rows n1 = rA + TPCA*e of columns V*jA .. V*jA + V-1 of its tile
    const int log2TPCA = log2L1 - 3, log2JA = log2CA - log2V;
    // step B: thread (jB, rB) owns samples n2 = rB + TPCB*e of rows V*jB .. V*jB + V-1 of its tile
    const int log2TPCB = log2L2 - 3, log2JB = log2CB - log2V;
    const long long n = 1ll << (log2L1 + log2L2);
    const unsigned tile_bytes = (unsigned)SZ << (log2L1 + log2CA);
    const unsigned phase_bytes = tile_bytes << log2TS;

    // LDS: [ landing image | work image | tables | 4 words ]
    unsigned char* const land = smem;
    unsigned char* const work = smem + tile_bytes;
    unsigned char* const tab_bytes = smem + 2 * tile_bytes;
    {
        const vec16<T>* src = reinterpret_cast<const vec16<T>*>(p.tables);
        vec16<T>* dst = reinterpret_cast<vec16<T>*>(tab_bytes);
        for (int i = tid_invariant; i < (p.tables_bytes >> 4); i += nthreads) dst[i] = src[i];
    }
    const cpx<T>* tab = reinterpret_cast<const cpx<T>*>(tab_bytes);
    volatile unsigned* sh = reinterpret_cast<volatile unsigned*>(tab_bytes + p.tables_bytes);
    StageTw<T> twA, twB;
    twA.sa = tab;
    twA.sb = tab + p.o_sb1;
    twA.sa_bits = p.sa1_bits;
    twA.log2L = log2L1;
    twB.sa = tab + p.o_sa2;
    twB.sb = tab + p.o_sb2;
    twB.sa_bits = p.sa2_bits;
    twB.log2L = log2L2;

    // ---- team formation: who shares my L2?
    if (tid_invariant == 0) {
        const unsigned xcc = FFT_XCC_ID(p.n_teams);
        const unsigned slot = FFT_ATOMIC_ADD_AGENT(&p.ctl[TEAM_CTL_COUNT + 32 * xcc], 1u);
        FFT_ATOMIC_ADD_AGENT(&p.ctl[TEAM_CTL_REGISTERED], 1u);
        unsigned ok = 1;
        const long long t0 = FFT_CLOCK();
        while (FFT_ATOMIC_LOAD_AGENT(&p.ctl[TEAM_CTL_REGISTERED]) < (unsigned)FFT_NBLOCKS) {
            if (FFT_ATOMIC_LOAD_AGENT(&p.ctl[TEAM_CTL_STATUS]) != 0 || FFT_CLOCK() - t0 > p.timeout_ticks) {
                ok = 0;  // not every workgroup is resident (the device is shared): give up before touching anything
                break;
            }
            FFT_SLEEP();
        }
        if (ok) {
            for (int x = 0; x < 16; x++) {
                const unsigned cnt = FFT_ATOMIC_LOAD_AGENT(&p.ctl[TEAM_CTL_COUNT + 32 * x]);
                if (cnt != (x < p.n_teams ? (unsigned)TS : 0u)) ok = 0;
            }
        }
        if (!ok) FFT_ATOMIC_STORE_AGENT(&p.ctl[TEAM_CTL_STATUS], (unsigned)TEAM_STATUS_NO_TEAMS);
        sh[0] = slot;
        sh[1] = xcc;
        sh[2] = ok;
        sh[3] = 0;  // abort word of the barrier
    }
    FFT_SYNC();
    if (!sh[2]) return;
    const int c = (int)FFT_UNIFORM(sh[0]);  // my seat in the team
    const int team = (int)FFT_UNIFORM(sh[1]);

    unsigned char* const sbase = p.scratch + (size_t)team * 2 * phase_bytes;
    unsigned* const flags = p.ctl + TEAM_CTL_FLAGS + 32 * team;
    unsigned gen = 0;  // phases completed by this team

    cpx<T> keep[NT][E][V];
    int n_ev = 0;
    auto ev = [&]() __attribute__((always_inline)) {  // profiling timeline (tools/team_trace.py); one scalar branch when off
        if (p.trace && tid_invariant == 0 && n_ev < p.trace_events) {
            p.trace[(long long)FFT_BID * p.trace_events + n_ev] = FFT_CLOCK();
            n_ev++;
        }
    };
    ev();  // 0: team formed

    // LDS-DMA of column tile t of a transform into the landing image: element l = rA + TPCA*e of columns
    // V*jA.. goes to image slot (l * JA + jA) = e * nthreads + tid: lane-linear, as the DMA requires
    auto dma_column_tile = [&](const cpx<T>* inb, int t) __attribute__((always_inline)) {
        int tid = tid_invariant;
        FFT_OPAQUE(tid);
        const int jA = tid & ((1 << log2JA) - 1), rA = tid >> log2JA;
        const int c0 = ((t << log2TS) + c) << log2CA;  // the team's workgroups read one contiguous TS*CA-column band
        const cpx<T>* src = inb + ((long long)rA << log2L2) + c0 + V * jA;
        unsigned char* dst = land + (size_t)tid * 16;
        FFT_UNROLL
        for (int e = 0; e < E; e++)
            FFT_GLDS16(src + ((long long)e << (log2TPCA + log2L2)), dst + (size_t)e * nthreads * 16, 0);
    };

    bool have_first = false;
    for (int b = team; b < p.nb; b += p.n_teams) {
        const cpx<T>* inb = p.in + (long long)b * n;
        cpx<T>* outb = p.out + (long long)b * n;
        if (!have_first) dma_column_tile(inb, 0);

        // ================= step A: L2-strided column FFTs of length L1, results stay in registers
        FFT_NOUNROLL
        for (int t = 0; t < NT; t++) {
            int tid = tid_invariant;
            FFT_OPAQUE(tid);
            const int jA = tid & ((1 << log2JA) - 1), rA = tid >> log2JA;
            cpx<T> x[1][E][V];
            FFT_WAIT_VM0();   // my part of the tile has landed ...
            FFT_SYNC_LDS();   // ... everybody's has; the work image is free (previous tile's last stage has read it)
            ev();  // A: tile landed
            const bool more = (t + 1 < NT);
            if (!(p.ablate & 2)) {
                team_all_stages<T, E, V>(x, land, work, twA, rA, jA, log2JA, log2TPCA, log2L1, [&]() {
                    if (more) dma_column_tile(inb, t + 1);  // flies during the remaining stages
                }, p.inverse != 0);  // inverse = forward transform between two re<->im swaps: first one here
            } else if (more) {
                FFT_SYNC_LDS();
                dma_column_tile(inb, t + 1);
            }
            if (!(p.ablate & 1)) {  // W_n^(k1 n2), two-level LDS table
                const cpx<T>* t0 = tab + p.o_t0;
                const cpx<T>* t1 = tab + p.o_t1;
                const unsigned m0 = (1u << p.t0_bits) - 1u;
                const unsigned c0 = (unsigned)(((t << log2TS) + c) << log2CA);
                FFT_UNROLL
                for (int e = 0; e < E; e++) {
                    const unsigned K = (unsigned)(rA + (e << log2TPCA));
                    FFT_UNROLL
                    for (int vv = 0; vv < V; vv++) {
                        const unsigned m = K * (c0 + V * jA + vv);
                        x[0][e][vv] = cmul(x[0][e][vv], cmul(t0[m & m0], t1[m >> p.t0_bits]));
                    }
                }
            }
            FFT_UNROLL
            for (int tt = 0; tt < NT; tt++) {
                if (t == tt) {
                    FFT_UNROLL
                    for (int e = 0; e < E; e++) {
                        FFT_UNROLL
                        for (int vv = 0; vv < V; vv++) keep[tt][e][vv] = x[0][e][vv];
                    }
                }
            }
            ev();  // A: tile transformed
        }

        // ================= step B: NT phases of (hand over through L2, row FFTs of length L2, transposed store)
        FFT_NOUNROLL
        for (int ph = 0; ph < NT; ph++) {
            int tid = tid_invariant;
            FFT_OPAQUE(tid);
            const int jA = tid & ((1 << log2JA) - 1), rA = tid >> log2JA;
            const int jB = tid & ((1 << log2JB) - 1), rB = tid >> log2JB;
            unsigned char* const sb = sbase + (size_t)(gen & 1u) * phase_bytes;
            {
                // This phase hands over rows k1 = ph*L1/NT + rA + TPCA*ee, ee < EP.  Row cp*CB + i of the phase goes
                // to workgroup cp, whose window image is its row tile in the stage layout [n2][CB rows]: the 8
                // (or 16) bytes of one value per store, CB lanes filling one CB*SZ-byte segment.
                cpx<T> y[EP][NT][V];
                FFT_UNROLL
                for (int pp = 0; pp < NT; pp++) {
                    if (ph == pp) {
                        FFT_UNROLL
                        for (int ee = 0; ee < EP; ee++) {
                            FFT_UNROLL
                            for (int tt = 0; tt < NT; tt++) {
                                FFT_UNROLL
                                for (int vv = 0; vv < V; vv++) y[ee][tt][vv] = keep[tt][pp * EP + ee][vv];
                            }
                        }
                    }
                }
                FFT_UNROLL
                for (int ee = 0; ee < EP; ee++) {
                    const int row = rA + (ee << log2TPCA);  // cp * CB + i
                    const int cp = row >> log2CB, i = row & ((1 << log2CB) - 1);
                    unsigned char* dst = sb + (size_t)cp * tile_bytes + (size_t)i * SZ;
                    FFT_UNROLL
                    for (int tt = 0; tt < NT; tt++) {
                        const int n2 = (((tt << log2TS) + c) << log2CA) + V * jA;
                        FFT_UNROLL
                        for (int vv = 0; vv < V; vv++)
                            *reinterpret_cast<cpx<T>*>(dst + ((size_t)(n2 + vv) << log2CB) * SZ) = y[ee][tt][vv];
                    }
                }
            }
            // ---- team barrier: my stores are in L2, then everybody's are
            ev();  // B: hand-over stores issued
            FFT_WAIT_VM0();
            FFT_SYNC();
            ev();  // B: hand-over stores in L2
            gen++;
            if (tid < FFT_TEAM_POLL_LANES) {
                if (tid == 0) FFT_L2_FLAG_STORE(&flags[c], gen);
                const long long t0 = FFT_CLOCK();
                while (!team_all_arrived(flags, TS, gen, tid)) {
                    if (FFT_CLOCK() - t0 > p.timeout_ticks) {
                        if (tid == 0) {
                            FFT_ATOMIC_STORE_AGENT(&p.ctl[TEAM_CTL_STATUS], (unsigned)TEAM_STATUS_TIMEOUT);
                            sh[3] = 1;
                        }
                        break;
                    }
                    FFT_SLEEP();
                }
            }
            FFT_SYNC();
            if (sh[3]) return;
            ev();  // B: team barrier passed

            // ---- my row tile: tile_bytes contiguous bytes of the window, L2 -> LDS (sc1: never through my L1)
            {
                const unsigned char* src = sb + (size_t)c * tile_bytes + (size_t)tid * 16;
                unsigned char* dst = land + (size_t)tid * 16;
                FFT_UNROLL
                for (int e = 0; e < E; e++) FFT_GLDS16(src + (size_t)e * nthreads * 16, dst + (size_t)e * nthreads * 16, 16);
            }
            FFT_WAIT_VM0();
            FFT_SYNC_LDS();
            ev();  // B: row tile landed
            cpx<T> x[1][E][V];
            const bool next_transform = (ph == NT - 1) && (b + p.n_teams < p.nb);
            if (!(p.ablate & 2)) {
                team_all_stages<T, E, V>(x, land, work, twB, rB, jB, log2JB, log2TPCB, log2L2, [&]() {
                    // the next transform's first column tile flies during the last row FFTs and stores
                    if (next_transform) dma_column_tile(inb + (long long)p.n_teams * n, 0);
                }, false);
            } else if (next_transform) {
                FFT_SYNC_LDS();
                dma_column_tile(inb + (long long)p.n_teams * n, 0);
            }
            if (next_transform) have_first = true;
            ev();  // B: rows transformed
            // X[k1 + L1*k2]: slot e holds k2 = rB + TPCB*e of rows k1 = ph*L1/NT + c*CB + V*jB + (0..V-1)
            const long long k1 = ((long long)ph << (log2L1 - log2NT)) + ((long long)c << log2CB) + V * jB;
            FFT_UNROLL
            for (int e = 0; e < E; e++) {
                const long long K = rB + (e << log2TPCB);
                vec16<T> v;
                FFT_UNROLL
                for (int vv = 0; vv < V; vv++) {
                    cpx<T> z = p.inverse ? cswap(x[0][e][vv]) : x[0][e][vv];
                    v.c[vv] = (p.scale != (T)1) ? cscale(z, p.scale) : z;
                }
                *reinterpret_cast<vec16<T>*>(outb + (K << log2L1) + k1) = v;
            }
            ev();  // B: result stores issued
        }
    }
}

}  // namespace fftk
