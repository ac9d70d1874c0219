// fft_team.h -- team_fft_kernel: a whole transform per XCD, ONE HBM round trip.
//
// The two-pass (four-step) plan of fft_engine.h moves every element through HBM
// twice: pass A writes the n intermediate values to a scratch image, pass B reads
// them back.  That caps the algorithmic bandwidth at half of what the memory
// system moves.  This kernel keeps the intermediate ON the XCD:
//
//   * A "team" is TS workgroups (one per CU; TS = 32, a whole XCD, for n = 2^20
//     fp32, and 16, 8, 4, 2 of an XCD's 32 for n = 2^19 .. 2^16) that run on ONE
//     XCD and therefore share one 4 MiB L2.  Teams are formed at run time
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
//     reader's L1.  The team barrier is one arrival counter per team in that L2
//     (no-return atomic add; polled by the first wave of a workgroup with scalar
//     loads, the other waves wait at the workgroup barrier).  Every spin is bounded.
//
// HBM traffic per transform: n elements in, n elements out -- the algorithmic
// minimum; the transposition traffic stays inside the XCD.
#pragma once

#include "fft_kernels.h"

// chunks of a tile's landing DMA that go out from the first stage slot (the rest from the second)
#ifndef FFT_TEAM_DMA_FIRST
#define FFT_TEAM_DMA_FIRST(nch) ((nch) / 2)
#endif

namespace fftk {

// control block (32-bit words), zeroed before every launch
enum {
    TEAM_CTL_REGISTERED = 0,          // workgroups that have registered; TEAM_CTL_POISON set: formation was given up
    TEAM_CTL_NEXT = 2,                // team_quad_kernel, dynamic: transforms claimed beyond the first of every team (the next unclaimed one is n_teams + this)
    TEAM_CTL_STATUS = 1,              // 0 = done by this kernel; 1 = teams could not be formed (nothing touched); 2 = barrier timeout
    TEAM_CTL_COUNT = 32,              // + 32 * xcc : workgroups registered on that XCD (own 128-byte line each)
    TEAM_CTL_FLAGS = 32 + 32 * 16,    // + 32 * team : the team's barrier line, one generation word per member
    TEAM_CTL_MAX_TEAMS = 256,         // 8 XCDs x 32 seats, teams of one
    // team_quad_kernel's pair protocol (SLOTS = 3), + 32 * (2 TS + 1) * team for teams of TS seats: a line per seat whose first word counts
    // the sender units whose values of the seat's image are in L2; every seat's "rounds landed" word, one copy of that line PER READING SEAT
    // (256 waves polling one line of the L2 made a poll 2.3 us: profiles/r4_ab_pair_protocol_polls.txt); one line for the team's next
    // transform (8 bytes: index + 1, the iteration it is for).  3 lines per workgroup cover every team size.
    TEAM_CTL_PAIR = 32 + 32 * 16 + 32 * TEAM_CTL_MAX_TEAMS,
    TEAM_CTL_WORDS = TEAM_CTL_PAIR + 32 * 3 * TEAM_CTL_MAX_TEAMS
};
// sticky words: NOT zeroed per launch (a TIMEOUT of any queued execute must survive the next execute's memset of the
// control block); the host reads and clears them when it syncs (team_status_of)
enum { TEAM_STICKY_TIMEOUTS = 0, TEAM_STICKY_FALLBACKS = 1, TEAM_STICKY_WORDS = 32 };
enum { TEAM_STATUS_OK = 0, TEAM_STATUS_NO_TEAMS = 1, TEAM_STATUS_TIMEOUT = 2 };
constexpr unsigned TEAM_CTL_POISON = 0x80000000u;

template <typename T>
struct TeamParams {
    const cpx<T>* in;
    cpx<T>* out;
    const cpx<T>* tables;    // blob [sa1 | sb1 | sa2 | sb2 | t0 | t1]
    unsigned char* scratch;  // one window pair of 2 * (tile_bytes << log2TS) bytes per team
    unsigned* ctl;
    unsigned* sticky;        // TEAM_STICKY_WORDS words that survive the per-launch memset of ctl
    int tables_bytes;
    int data_bytes;          // LDS bytes of the data region (stage exchange / row staging image); tables follow
    int log2L1, log2L2, log2CA, log2CB, log2TS;
    int n_xcc;                // XCDs the launch must cover, with exactly 2^log2seats workgroups each
    int log2seats;            // an XCD's seats are dealt to 2^(log2seats - log2TS) teams of 2^log2TS
    int nb;
    int inverse;
    int o_sb1, o_sa2, o_sb2, o_t0, o_t1;
    int sa1_bits, sa2_bits, t0_bits;
    long long timeout_ticks;       // bound of every team wait, in FFT_CLOCK ticks: a deadlock breaker, never reached by a formed team
    long long form_timeout_ticks;  // bound of the formation spin: a shared device falls back to the multi-pass plan after this
    int tune;                 // experiments: bit 0 hand-over of phase 2 from the first hook of phase 0 too
    int nt_mask;              // cache policy: bit 0 column-tile DMA nt, bit 1 result stores nt, bit 2 window loads sc1 nt (read once)
    int force_no_teams;       // tests: pretend the placement check failed (exercises the two-pass fallback)
    int tile_rot;             // column_block(): seats rotate by this many blocks per tile
    int dynamic;              // team_quad_kernel: 1 = a team claims its next transform from a device-wide counter (TEAM_CTL_NEXT) instead of team + it * n_teams
    int seat_rot;             // experiments: seat = (registration order + seat_rot) mod TS
    int dma_split, dma_split2;  // (unused: the landing DMA goes out in two halves, from slots 0 and 1 -- compile-time, so that the chunk loop folds)
    int ablate;               // experiments: 1 skip the inter-pass twiddle, 2 skip the stages, 4 no result stores, 8 no column-tile DMA
    long long* trace;         // profiling: not NULL = every workgroup logs FFT_CLOCK at its first trace_events events
    int trace_events;
    T scale;
};

// Low-level index bits of a stage twiddle table of a length-2^log2L sub-transform (log2L: single-level table).
FFT_HOST_DEVICE int team_stage_table_bits(int elem_bytes, int log2L) {
    return ((elem_bytes << log2L) > (elem_bytes == 16 ? 4096 : 8192)) ? (log2L + 1) / 2 : log2L;
}

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

// ---- team formation: who shares my L2?  Called by thread 0 of every workgroup; fills sh[0..3] = [slot, xcc, ok, 0].
// The decision is ATOMIC for the whole launch: a workgroup proceeds only if it loads REGISTERED == the grid size,
// unpoisoned; a workgroup that gives up (the device is shared: not everybody became resident within the formation
// timeout) poisons the word with a compare-and-swap that fails once the count is complete -- so either every
// workgroup sees the complete count and runs as a team member, or every one sees the poison (a late arrival's add
// lands on a poisoned word and can never make it equal the grid size) and leaves before touching anything; the
// multi-pass plan queued behind the kernel reads STATUS == NO_TEAMS and does the work.
template <typename T>
FFT_DEVICE void team_form(const TeamParams<T>& p, unsigned* sh) {
    FFT_TEST_DELAY();
    const unsigned xcc = FFT_XCC_ID(p.n_xcc);
    // (my seat: relaxed; the RELEASE of the registration below orders it in front of that, the one acquire fence behind the poll loop
    // orders every reader's counter loads behind every registration -- no cache maintenance per poll)
    const unsigned slot = FFT_ATOMIC_ADD_AGENT_RELAXED(&p.ctl[TEAM_CTL_COUNT + 32 * xcc], 1u);
    FFT_ATOMIC_ADD_AGENT_RELEASE(&p.ctl[TEAM_CTL_REGISTERED], 1u);
    const unsigned nblocks = (unsigned)FFT_NBLOCKS;
    unsigned ok = 0;
    const long long t0 = FFT_CLOCK();
    for (;;) {
        const unsigned v = FFT_L2_FLAG_LOAD(&p.ctl[TEAM_CTL_REGISTERED]);
        if (v == nblocks) { ok = 1; break; }
        if (v & TEAM_CTL_POISON) break;
        if (FFT_CLOCK() - t0 > p.form_timeout_ticks) {
            const unsigned found = FFT_ATOMIC_CAS_AGENT(&p.ctl[TEAM_CTL_REGISTERED], v, v | TEAM_CTL_POISON);
            if (found == v) break;  // poisoned by me
            continue;               // the word moved on (another registration, the last one, or somebody's poison): look again
        }
        FFT_SLEEP();
    }
    FFT_FENCE_ACQUIRE_AGENT();
    unsigned xcc_rank = 0;  // my XCD's place among the XCDs the launch landed on
    if (ok) {  // every workgroup has registered: the per-XCD counts are final and the same for every reader
        // exactly n_xcc XCC ids must hold a full set of seats and no other id any workgroup.  WHICH ids is the hardware's
        // business: a partition of the device (DPX / QPX / CPX) need not number its XCDs from 0, so a team is
        // numbered by the RANK of its XCC id among the ids present, not by the id itself.
        // (sixteen RELAXED device-scope loads, all in flight at once: the acquire fence above has ordered them behind every
        // registration, and sixteen acquire loads in a row -- a memory round trip and a cache invalidate each -- made formation 40 us
        // of every launch, profiles/r3_quad_fill.txt)
        unsigned cnts[16];
        FFT_UNROLL
        for (int x = 0; x < 16; x++) cnts[x] = FFT_L2_FLAG_LOAD(&p.ctl[TEAM_CTL_COUNT + 32 * x]);
        int full = 0;
        FFT_UNROLL
        for (int x = 0; x < 16; x++) {
            const unsigned cnt = cnts[x];
            if (cnt == (1u << p.log2seats)) {
                full++;
                if ((unsigned)x < xcc) xcc_rank++;
            } else if (cnt != 0u) {
                ok = 0;
            }
        }
        if (full != p.n_xcc) ok = 0;
    }
    if (p.force_no_teams) ok = 0;
    if (!ok && FFT_ATOMIC_CAS_AGENT(&p.ctl[TEAM_CTL_STATUS], 0u, (unsigned)TEAM_STATUS_NO_TEAMS) == 0u)
        FFT_ATOMIC_ADD_AGENT(&p.sticky[TEAM_STICKY_FALLBACKS], 1u);  // once per launch
    sh[0] = slot;
    sh[1] = xcc_rank;
    sh[2] = ok;
    sh[3] = 0;  // set when a team wait timed out: later waits return at once, the status word tells the host
}

// A team wait ran into its bound (cannot happen to a formed team short of a hardware fault; the bound keeps a broken
// launch from hanging the device): results are invalid.  STATUS never goes back from TIMEOUT, NO_TEAMS is not
// overwritten (CAS from 0), and the sticky counter tells the host even if later executes zero the control block.
template <typename T>
FFT_DEVICE void team_report_timeout(const TeamParams<T>& p) {
    FFT_ATOMIC_CAS_AGENT(&p.ctl[TEAM_CTL_STATUS], 0u, (unsigned)TEAM_STATUS_TIMEOUT);
    FFT_ATOMIC_ADD_AGENT(&p.sticky[TEAM_STICKY_TIMEOUTS], 1u);
}

// twiddles by powers (stockham_stage_rw SWZ bit 3, team_interpass_twiddle): build switch for same-box A/B
#ifndef FFT_TEAM_TW_TREE
#define FFT_TEAM_TW_TREE 1
#endif
#ifndef FFT_TEAM_TW_TREE_F64      // fp64 too (4.2e-16 -> 6.9e-16 relative: far inside the 1e-6 budget): 2^19 +2.5 %, 2^18 +4.5 %, 2^17 +2 %, below +-0
#define FFT_TEAM_TW_TREE_F64 1
#endif
#ifndef FFT_TEAM_TW_TREE_F64_MIN  // smallest log2 n
#define FFT_TEAM_TW_TREE_F64_MIN 17
#endif

#define FFT_TEAM_GEO(l1, l2, ca, cb, ts) ((l1) | ((l2) << 5) | ((ca) << 10) | ((cb) << 15) | ((ts) << 20))

// All Stockham stages of one tile whose samples sit in the LDS-DMA landing image `land` ([element][column], the
// stage layout): radix-E stages (E = the thread's element count: 16 fp32, 8 fp64) plus one stage of the remaining
// power of two.  The first stage reads `land` and writes the work image `work`, the others run in `work`.
// `hook(s, total)` runs in the middle of stage s < total - 1, right after the barrier that says every wave has
// read that stage's inputs (at s == 0 the landing image is free again), and once more, s == total - 1, after the
// last stage.  These slots are where the kernel issues its memory traffic, spread out so that a CU's short memory
// queue never makes the issuing waves wait long, and so that it flies under the remaining stages.
template <class Hook>
struct StageHookAt {
    Hook& hook;
    int s, total;
    FFT_DEVICE void operator()() const { hook(s, total); }
};

// which geometries build their twiddles as powers (FFT_TEAM_TW_TREE; measured, profiles/r2_ab_team_variants.txt (3), (11)): fp32
// from n = 2^18 up (2^20 +7 %, 2^19 +1.4 %, 2^18 +0.8 %, 2^16 -1.4 %), fp64 from 2^17 up; GEO = 0 (the emulation's generic
// geometry) exercises the power path
template <typename T, int GEO>
struct TeamTwTree {
    static constexpr bool value = FFT_TEAM_TW_TREE && (sizeof(T) == 4 || FFT_TEAM_TW_TREE_F64) &&
                                  (GEO == 0 || ((GEO & 31) + ((GEO >> 5) & 31)) >= (sizeof(T) == 4 ? 18 : FFT_TEAM_TW_TREE_F64_MIN));
};

template <typename T, int E, bool TREE = (FFT_TEAM_TW_TREE && sizeof(T) == 4), class Hook>
FFT_DEVICE void team_all_stages(cpx<T> (&x)[1][E][1], const unsigned char* land, unsigned char* work, const StageTw<T>& tw,
                                int r, int j, int log2J, int log2TPC, int log2L, Hook&& hook, bool swap_in) {
    constexpr int log2E = Log2<E>::value;
    constexpr int TWB = TREE ? 8 : 0;  // stage twiddles by powers
    int log2Lprev = log2L, log2P = 0;
    const int n_full = log2L / log2E;
    const int rem = log2L - n_full * log2E;
    const int total = n_full + (rem ? 1 : 0);
    // a final radix-4 stage reads through the bank swizzle (stage_swizzle), the exchange before it writes through it
    const bool swz = (rem == 2) && n_full >= 2 && sizeof(cpx<T>) == 8;
    FFT_UNROLL
    for (int s = 0; s < n_full; s++) {
        StageHookAt<Hook> h{hook, s, total};
        if (swz && s == n_full - 1)
            stockham_stage_rw<T, E, E, 1, 1, 2 | 4 | TWB>(x, s == 0 ? land : work, work, 0, tw, r, j, log2J, log2TPC, log2Lprev, log2P,
                                                      false, s == total - 1, h, s == 0 && swap_in);
        else
            stockham_stage_rw<T, E, E, 1, 1, 4 | TWB>(x, s == 0 ? land : work, work, 0, tw, r, j, log2J, log2TPC, log2Lprev, log2P, false,
                                                  s == total - 1, h, s == 0 && swap_in);
    }
    // the remaining stage is always the last one: no exchange, no hook
    if (rem == 1) stockham_stage_rw<T, E, 2, 1, 1>(x, work, work, 0, tw, r, j, log2J, log2TPC, log2Lprev, log2P, false, true, StageNoHook());
    if (rem == 2) {
        if (swz) stockham_stage_rw<T, E, 4, 1, 1, 1>(x, work, work, 0, tw, r, j, log2J, log2TPC, log2Lprev, log2P, false, true, StageNoHook());
        else stockham_stage_rw<T, E, 4, 1, 1>(x, work, work, 0, tw, r, j, log2J, log2TPC, log2Lprev, log2P, false, true, StageNoHook());
    }
    if (E > 8 && rem == 3)
        stockham_stage_rw<T, E, (E > 8 ? 8 : 2), 1, 1>(x, work, work, 0, tw, r, j, log2J, log2TPC, log2Lprev, log2P, false, true, StageNoHook());
    hook(total - 1, total);  // the last slot: after the final stage (which has no exchange, hence no mid-stage barrier)
}

// x[e] *= W_n^(m_start + e * m_step) from the two-level LDS table (t0: low t0_bits of the exponent, t1: the rest).
// The table reads of FOUR slots are issued together and then consumed: left to itself the compiler emits read, read,
// wait, multiply per slot, i.e. sixteen LDS round trips in a row on the critical path of every column tile.
template <typename T, int E, bool TREE = (FFT_TEAM_TW_TREE && sizeof(T) == 4)>
FFT_DEVICE void team_interpass_twiddle(cpx<T> (&x)[1][E][1], const cpx<T>* t0, const cpx<T>* t1, int t0_bits, unsigned m_start,
                                       unsigned m_step) {
    constexpr int B = 4;
    const unsigned m0 = (1u << t0_bits) - 1u;
    if constexpr (TREE) {
        // W^(m_start + e m_step) = base * step^e: four table reads (base and step, two-level each), then step^2, ^4, ^8 by
        // squaring and w[e | bit] = w[e] * step^bit -- 18 products at most 4 deep instead of 32 reads and their index arithmetic
        const cpx<T> base = cmul(t0[m_start & m0], t1[m_start >> t0_bits]);
        cpx<T> sp = cmul(t0[m_step & m0], t1[m_step >> t0_bits]);
        cpx<T> w[E];
        w[0] = base;
        FFT_UNROLL
        for (int bit = 1; bit < E; bit <<= 1) {
            FFT_UNROLL
            for (int e = 0; e < bit; e++) w[e | bit] = cmul(w[e], sp);
            sp = cmul(sp, sp);
        }
        FFT_UNROLL
        for (int e = 0; e < E; e++) x[0][e][0] = cmul(x[0][e][0], w[e]);
        return;
    }
    FFT_UNROLL
    for (int e0 = 0; e0 < E; e0 += B) {
        cpx<T> a[B], b[B];
        FFT_UNROLL
        for (int k = 0; k < B; k++) {
            const unsigned m = m_start + (unsigned)(e0 + k) * m_step;
            a[k] = t0[m & m0];
            b[k] = t1[m >> t0_bits];
        }
        FFT_SCHED_BARRIER();
        FFT_UNROLL
        for (int k = 0; k < B; k++) x[0][e0 + k][0] = cmul(x[0][e0 + k][0], cmul(a[k], b[k]));
        FFT_SCHED_BARRIER();
    }
}

// Two lanes (l, l ^ mask) hold the same slots of two ADJACENT rows (even lane: row i, odd lane: row i + 1).  For the
// slot pair (s0, s1) each lane ends up with BOTH rows of ONE slot -- the even lane of s0, the odd lane of s1 -- i.e.
// 16 contiguous bytes of an image whose rows are adjacent in memory: half as many, twice as wide stores.
template <typename T>
FFT_DEVICE void pair_rows(cpx<T> own_s0, cpx<T> own_s1, bool odd, int mask, vec16<T>& out) {
    static_assert(vec16<T>::V == 2, "fp32 only: a 16-byte access holds two values");
    // selects on VALUES, component by component: a select between the two register-array elements themselves keeps
    // the optimizer from promoting the array to registers
    const T send_re = odd ? own_s0.re : own_s1.re;
    const T send_im = odd ? own_s0.im : own_s1.im;
    const T recv_re = FFT_XOR_EXCHANGE(send_re, mask, odd);  // odd <=> this lane has the mask bit set
    const T recv_im = FFT_XOR_EXCHANGE(send_im, mask, odd);
    out.c[0].re = odd ? recv_re : own_s0.re;  // row i
    out.c[0].im = odd ? recv_im : own_s0.im;
    out.c[1].re = odd ? own_s1.re : recv_re;  // row i + 1
    out.c[1].im = odd ? own_s1.im : recv_im;
}

// Hand over K values y[] of one column n2 (rows r + TPC*k of the current phase) into the window `sb`.  Row cp*CB + i of
// the phase goes to workgroup cp, whose window image is its row tile in the stage layout [n2][CB rows].  fp32: the lanes
// of rows i, i+1 (r even / odd, `pair_mask` lanes apart) pair up so that every store is 16 bytes.
template <typename T, int K>
FFT_DEVICE void team_hand_over(unsigned char* sb, const cpx<T> (&y)[K], int n2, int r, int log2TPC, int log2CB,
                               unsigned tile_bytes, int pair_mask) {
    constexpr int SZ = (int)sizeof(cpx<T>);
    if constexpr (vec16<T>::V == 2) {
        const bool odd = (r & 1) != 0;
        FFT_UNROLL
        for (int q = 0; q < K / 2; q++) {
            const int row = (r & ~1) + (((2 * q) + (odd ? 1 : 0)) << log2TPC);  // even row of the pair
            const int cp = row >> log2CB, i = row & ((1 << log2CB) - 1);
            vec16<T> v;
            pair_rows<T>(y[2 * q], y[2 * q + 1], odd, pair_mask, v);
            *reinterpret_cast<vec16<T>*>(sb + (size_t)cp * tile_bytes + (((size_t)n2 << log2CB) + i) * SZ) = v;
        }
    } else {
        FFT_UNROLL
        for (int ee = 0; ee < K; ee++) {
            const int row = r + (ee << log2TPC);
            const int cp = row >> log2CB, i = row & ((1 << log2CB) - 1);
            *reinterpret_cast<cpx<T>*>(sb + (size_t)cp * tile_bytes + (((size_t)n2 << log2CB) + i) * SZ) = y[ee];
        }
    }
}

// Which row tile of the row step does row k1 of the intermediate belong to?  L1 / CB blocks of CB rows = NT * TS tiles.
//   plain : block bi = k1 / CB goes to phase bi / TS, seat bi % TS        (a seat's four tiles are L1 / 4 rows apart)
//   PAIR  : blocks 2p and 2p + 1 go to the SAME seat (p % TS) in the two phases 2 (p / TS) and 2 (p / TS) + 1: a seat's
//           tiles of phases (0, 1) and of (2, 3) are ADJACENT blocks of rows, so that it can write their results
//           together as 2 CB-row = 128-byte segments (fp32, CB = 8) instead of two rounds of 64-byte ones.
template <bool PAIR>
FFT_DEVICE void team_row_dest(int k1, int log2CB, int log2TS, int& phase, int& seat, int& i) {
    i = k1 & ((1 << log2CB) - 1);
    const int bi = k1 >> log2CB;
    if (PAIR) {
        phase = ((bi >> (log2TS + 1)) << 1) | (bi & 1);
        seat = (bi >> 1) & ((1 << log2TS) - 1);
    } else {
        phase = bi >> log2TS;
        seat = bi & ((1 << log2TS) - 1);
    }
}
// first row of the tile of (phase, seat)
template <bool PAIR>
FFT_DEVICE int team_tile_row0(int phase, int seat, int log2CB, int log2TS) {
    if (PAIR) return (((((phase >> 1) << log2TS) + seat) << 1) | (phase & 1)) << log2CB;
    return ((phase << log2TS) + seat) << log2CB;
}

// Hand over the K register slots slot0 .. slot0 + K - 1 of one column n2 (slot e = row k1 = r + TPC * e of the
// intermediate): every value goes into the window of ITS phase (win_of(phase): the window's base, or NULL = not now)
// at its receiver's row tile.  fp32: the lanes of rows k1, k1 + 1 pair up so that every store is 16 bytes.
template <typename T, int K, bool PAIR, class WinOf>
FFT_DEVICE void team_hand_over_rows(WinOf&& win_of, const cpx<T> (&y)[K], int slot0, int n2, int r, int log2TPC, int log2CB, int log2TS,
                                    unsigned tile_bytes, int pair_mask) {
    constexpr int SZ = (int)sizeof(cpx<T>);
    if constexpr (vec16<T>::V == 2) {
        const bool odd = (r & 1) != 0;
        FFT_UNROLL
        for (int q = 0; q < K / 2; q++) {
            vec16<T> v;
            pair_rows<T>(y[2 * q], y[2 * q + 1], odd, pair_mask, v);  // every lane takes part (cross-lane moves)
            const int k1 = (r & ~1) + ((slot0 + 2 * q + (odd ? 1 : 0)) << log2TPC);  // even row of the pair
            int phase, cp, i;
            team_row_dest<PAIR>(k1, log2CB, log2TS, phase, cp, i);
            unsigned char* sb = win_of(phase);
            if (sb) *reinterpret_cast<vec16<T>*>(sb + (size_t)cp * tile_bytes + (((size_t)n2 << log2CB) + i) * SZ) = v;
        }
    } else {
        FFT_UNROLL
        for (int ee = 0; ee < K; ee++) {
            const int k1 = r + ((slot0 + ee) << log2TPC);
            int phase, cp, i;
            team_row_dest<PAIR>(k1, log2CB, log2TS, phase, cp, i);
            unsigned char* sb = win_of(phase);
            if (sb) *reinterpret_cast<cpx<T>*>(sb + (size_t)cp * tile_bytes + (((size_t)n2 << log2CB) + i) * SZ) = y[ee];
        }
    }
}

// ---------------------------------------------------------------------------
// The kernel.  Per transform a workgroup signals NT + 1 "arrivals" (generation numbers count up across transforms) on
// its team's counter (NT == 4: every member waits for everybody's arrival g before its own arrival g + 1, so
// count >= TS * g says "everybody has arrived at g") or flag line (NT < 4, tests: one generation word per member):
//   a0        my hand-over of phases 0 and 1 (written while the column tiles were transformed) is in L2
//   a1        my row tile of phase 0 has landed in LDS
//   a(ph+2)   end of row phase ph < NT-1: my row tile of phase ph+1 has landed, my hand-over of phase ph+2 is in L2
// Window S[q & 1] of the team holds the hand-over of phase q.  Who waits for what:
//   reading  phase q   (DMA L2 -> LDS)  needs everybody's hand-over of q     : a0 for q <= 1, a(q) for q >= 2
//   writing  phase q+2 (into S[q & 1])  needs everybody to have read phase q : a(q+1)
//   the next transform's hand-over of phases 0, 1 needs everybody's last reads: aNT of this transform
// NT = tiles per workgroup per step = n / (TS * tile elements); GEO != 0 bakes the geometry into the instantiation.
// ---------------------------------------------------------------------------
// E = elements per thread, all of ONE column = the radix of the stages.  A tile is 64 KiB (4096 * V16 elements), so the
// workgroup has 4096 * V16 / E threads: fp32 E = 16 -> 512 threads (2 waves per SIMD), E = 8 -> 1024 threads (4 waves per
// SIMD: one more LDS exchange per tile, twice the waves to hide LDS latency and barriers behind); fp64 E = 8 -> 512.
//
// ASPLIT: the column step reads 128-byte row segments instead of 64-byte ones (a CU's memory pipeline streams those
// half again as fast).  A column tile becomes HALF as high and TWICE as wide: the even rows, then the
// odd rows, of 2 CA columns; each half is a length-L1/2 transform, and a radix-2 butterfly in registers joins them,
//     Y[k'] = Ye[k'] + W_L1^k' Yo[k'],   Y[k' + L1/2] = Ye[k'] - W_L1^k' Yo[k'],
// which puts the low half of the rows (phases 0, 1: handed over at once) and the high half (phases 2, 3: kept) into
// the same thread.  Row step, windows and arrivals are untouched.
template <typename T, int NT, int E, int GEO, bool ASPLIT = false>
FFT_KERNEL void FFT_LAUNCH_BOUNDS2(4096 * vec16<T>::V / E, 16 * vec16<T>::V / E) team_fft_kernel(TeamParams<T> p) {
    constexpr int V16 = vec16<T>::V;  // complex values per 16-byte lane access of HBM / L2: 2 (fp32) or 1 (fp64)
    constexpr int log2V16 = Log2<V16>::value;
    constexpr int log2E = Log2<E>::value;
    constexpr int NCH = E / V16;      // 16-byte chunks of a tile per thread (LDS-DMA instructions, result stores)
    constexpr int SZ = (int)sizeof(cpx<T>);
    constexpr int EP = E / NT;                     // register slots (rows r + TPC*e) that one phase hands over
    constexpr int NK = NT > 2 ? NT - 2 : 0;        // phases whose hand-over waits in registers (phases 0, 1 go out at once)
    constexpr int log2NT = Log2<NT>::value;
    constexpr int NARR = NT + 1;                   // arrivals per transform
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
    // step A: thread (jA, rA) owns rows n1 = rA + TPCA*e of column jA of its tile;  step B: thread (jB, rB) owns
    // samples n2 = rB + TPCB*e of row jB of its tile
    const int log2TPCA = log2L1 - log2E;
    const int log2TPCB = log2L2 - log2E;
    const long long n = 1ll << (log2L1 + log2L2);
    const unsigned tile_bytes = (unsigned)SZ << (log2L1 + log2CA);
    const unsigned phase_bytes = tile_bytes << log2TS;

    // LDS: [ landing image | work image | tables | 4 words ]
    unsigned char* const land = smem;
    unsigned char* const work = smem + tile_bytes;
    unsigned char* const tab_bytes = smem + 2 * tile_bytes;
    const unsigned land_lds = FFT_LDS_ADDR(land);
    {
        const vec16<T>* src = reinterpret_cast<const vec16<T>*>(p.tables);
        vec16<T>* dst = reinterpret_cast<vec16<T>*>(tab_bytes);
        for (int i = tid_invariant; i < (p.tables_bytes >> 4); i += nthreads) dst[i] = src[i];
    }
    const cpx<T>* tab = reinterpret_cast<const cpx<T>*>(tab_bytes);
    unsigned* const sh = reinterpret_cast<unsigned*>(tab_bytes + p.tables_bytes);  // [slot, xcc, ok, timed out]
    // stage tables: single-level up to 8 KiB (fp32) / 4 KiB (fp64), else two-level -- TeamStageTable is the rule the
    // planner lays the blob out by; decided from the (baked-in) geometry, so no per-lookup branch survives
    StageTw<T> twA, twB;
    twA.sa = tab;
    twA.sb = tab + p.o_sb1;
    twA.sa_bits = team_stage_table_bits(SZ, ASPLIT ? log2L1 - 1 : log2L1);  // ASPLIT: the planner lays the table out for L1 / 2
    twA.log2L = ASPLIT ? log2L1 - 1 : log2L1;
    twB.sa = tab + p.o_sa2;
    twB.sb = tab + p.o_sb2;
    twB.sa_bits = team_stage_table_bits(SZ, log2L2);
    twB.log2L = log2L2;

    // ---- team formation: who shares my L2?
    if (tid_invariant == 0) team_form(p, sh);
    FFT_SYNC();
    FFT_LDS_FRESH();
    if (!sh[2]) return;
    // seats of an XCD in registration order; consecutive runs of TS seats form the teams of that XCD
    const unsigned seat = (FFT_UNIFORM(sh[0]) + (unsigned)p.seat_rot) & ((1u << p.log2seats) - 1u);
    const int c = (int)(seat & (unsigned)(TS - 1));  // my seat in the team
    const int team = (int)((FFT_UNIFORM(sh[1]) << (p.log2seats - log2TS)) + (seat >> log2TS));
    const int n_teams = p.n_xcc << (p.log2seats - log2TS);

    unsigned char* const sbase = p.scratch + (size_t)team * 2 * phase_bytes;
    unsigned* const flags = p.ctl + TEAM_CTL_FLAGS + 32 * team;

    int n_ev = 0;
    auto ev = [&]() __attribute__((always_inline)) {  // profiling timeline (tools/team_trace.py); one scalar branch when off
        if (p.trace && tid_invariant == 0 && n_ev < p.trace_events - 1) {  // the last slot holds (team << 8 | seat)
            p.trace[(long long)FFT_BID * p.trace_events + n_ev] = FFT_CLOCK();
            n_ev++;
        }
    };
    ev();  // 0: team formed
    if (p.trace && tid_invariant == 0 && p.trace_events > 1) p.trace[(long long)FFT_BID * p.trace_events + p.trace_events - 1] = (team << 8) | c;

    // Arrivals.  NT == 4: every member waits for everybody's arrival g before its own arrival g + 1 (see the table
    // above the kernel), so ONE counter per team says it all: everybody has arrived at g <=> count >= TS * g.  The first
    // wave polls it with scalar loads (FFT_L2_COUNT_POLL: not behind the wave's stores and DMA), the others wait at the
    // workgroup barrier.  NT < 4 (tests): one generation word per member, polled with a vector load of the flag line.
    constexpr bool COUNTER = (NT == 4);
    auto wait_all = [&](unsigned g) __attribute__((always_inline)) {
        FFT_LDS_FRESH();
        if (sh[3]) return;
        if (tid_invariant < FFT_TEAM_POLL_LANES) {
            const long long t0 = FFT_CLOCK();
            while (COUNTER ? (int)(FFT_L2_COUNT_POLL(flags) - (g << log2TS)) < 0
                           : !team_all_arrived(flags, TS, g, tid_invariant & (FFT_TEAM_POLL_LANES - 1))) {
                if (FFT_CLOCK() - t0 > p.timeout_ticks) {
                    team_report_timeout(p);
                    sh[3] = 1;
                    break;
                }
                FFT_SLEEP();
            }
        }
        FFT_SYNC_LDS();
    };
    // call with every wave's relevant memory operations complete and behind a workgroup barrier
    auto arrive = [&](unsigned g) __attribute__((always_inline)) {
        if (tid_invariant == 0) {
            if (COUNTER) FFT_L2_COUNT_ADD(flags);
            else FFT_L2_FLAG_STORE(&flags[c], g);
        }
    };

    cpx<T> keep[NT][NK * EP + 1];  // [tile][slot - 2*EP]: the hand-over of phases >= 2

    // Column tile t of this workgroup is block column_block(t) of the transform's L2/CA column blocks.  The team covers
    // one contiguous band of TS blocks per t; WHICH block of the band a seat takes rotates with t: on MI355X two of the
    // sixteen 128-byte line positions of the band read measurably slower (+1.3 us per tile), and the rotation hands
    // each seat at most one of them per transform instead of all four to the same four seats (skew at the team waits).
    auto column_block = [&](int t) __attribute__((always_inline)) { return (t << log2TS) + ((c + p.tile_rot * t) & (TS - 1)); };

    // LDS-DMA of column tile t of a transform into the landing image.  The image is [row][CA columns], filled
    // lane-linearly in 16-byte chunks: chunk g = i * nthreads + tid is row g / (CA/V16), columns V16 * (g mod CA/V16)...
    auto dma_column_tile = [&](const cpx<T>* inb, int t, int i0, int i1) __attribute__((always_inline)) {
        int tid = tid_invariant;
        FFT_OPAQUE(tid);
        const int log2CPR = log2CA - log2V16;  // 16-byte chunks per image row
        const int c0 = column_block(t) << log2CA;  // the team's workgroups read one contiguous TS*CA-column band
        const cpx<T>* src = inb + ((long long)(tid >> log2CPR) << log2L2) + c0 + V16 * (tid & ((1 << log2CPR) - 1));
        const long long step = (long long)(nthreads >> log2CPR) << log2L2;  // rows per wave-front of chunks
        if (FFT_ABLATE(p.ablate & 8)) return;  // profiling: no input stream
        if (p.nt_mask & 1) {  // one branch per call, not one per chunk
            FFT_UNROLL
            for (int i = 0; i < NCH; i++)
                if (i >= i0 && i < i1) FFT_DMA16_NT(src + i * step, land, land_lds, (unsigned)(i * nthreads + tid) * 16u);
        } else {
            FFT_UNROLL
            for (int i = 0; i < NCH; i++)
                if (i >= i0 && i < i1) FFT_DMA16(src + i * step, land, land_lds, (unsigned)(i * nthreads + tid) * 16u);
        }
    };
    // LDS-DMA of my row tile of the window `sb` (tile_bytes contiguous bytes), served by the XCD's L2
    auto dma_row_tile = [&](const unsigned char* sb) __attribute__((always_inline)) {
        int tid = tid_invariant;
        FFT_OPAQUE(tid);
        const unsigned char* src = sb + (size_t)c * tile_bytes + (size_t)tid * 16;
        if (p.nt_mask & 4) {
            FFT_UNROLL
            for (int i = 0; i < NCH; i++)
                FFT_DMA16_L2_NT(src + (size_t)i * nthreads * 16, land, land_lds, (unsigned)(i * nthreads + tid) * 16u);
        } else {
            FFT_UNROLL
            for (int i = 0; i < NCH; i++)
                FFT_DMA16_L2(src + (size_t)i * nthreads * 16, land, land_lds, (unsigned)(i * nthreads + tid) * 16u);
        }
    };
    // column tile t covers columns [column_block(t) * CA, + CA); ASPLIT: half tile t = (group t / 2, row parity t % 2)
    // covers the 2 CA columns [column_group(t / 2) * 2 CA, + 2 CA)
    auto column_group = [&](int g) __attribute__((always_inline)) { return (g << log2TS) + ((c + p.tile_rot * g) & (TS - 1)); };
    auto hand_over = [&](unsigned char* sb, const cpx<T> (&y)[EP], int tt, int rA, int jA) __attribute__((always_inline)) {
        team_hand_over<T, EP>(sb, y, (column_block(tt) << log2CA) + jA, rA, log2TPCA, log2CB, tile_bytes, 1 << log2CA);
    };
    // ASPLIT geometry: thread (jA2, rA2) owns rows k' = rA2 + TPCA2 * e of column jA2 of a half tile
    const int log2H1 = log2L1 - 1, log2CA2 = log2CA + 1, log2TPCA2 = log2H1 - log2E;
    cpx<T> keep2[ASPLIT ? NT / 2 : 1][E];  // ASPLIT: the high half of the rows (phases 2, 3) of each column group
    auto dma_half_tile = [&](const cpx<T>* inb, int t, int i0, int i1) __attribute__((always_inline)) {
        int tid = tid_invariant;
        FFT_OPAQUE(tid);
        const int log2CPR = log2CA2 - log2V16;
        const int c0 = column_group(t >> 1) << log2CA2;
        const cpx<T>* src = inb + ((long long)(2 * (tid >> log2CPR) + (t & 1)) << log2L2) + c0 + V16 * (tid & ((1 << log2CPR) - 1));
        const long long step = (long long)(2 * (nthreads >> log2CPR)) << log2L2;
        FFT_UNROLL
        for (int i = 0; i < NCH; i++)
            if (i >= i0 && i < i1) FFT_DMA16(src + i * step, land, land_lds, (unsigned)(i * nthreads + tid) * 16u);
    };
    auto dma_first_tile = [&](const cpx<T>* inb) __attribute__((always_inline)) {
        if constexpr (ASPLIT) dma_half_tile(inb, 0, 0, NCH);
        else dma_column_tile(inb, 0, 0, NCH);
    };

    bool have_first = false;
    unsigned g0 = 1;  // generation of this transform's a0
    for (int b = team; b < p.nb; b += n_teams, g0 += NARR) {
        const cpx<T>* inb = p.in + (long long)b * n;
        cpx<T>* outb = p.out + (long long)b * n;
        if (!have_first) dma_first_tile(inb);

        // ================= step A: L2-strided column FFTs of length L1; phases 0, 1 are handed over at once
        if constexpr (ASPLIT) {
            static_assert(!ASPLIT || NT == 4, "the even / odd row split is built for four tiles per workgroup");
            cpx<T> ye[E];
            FFT_NOUNROLL
            for (int t = 0; t < NT; t++) {  // half tile t: column group t / 2, rows of parity t % 2
                int tid = tid_invariant;
                FFT_OPAQUE(tid);
                const int jA2 = tid & ((1 << log2CA2) - 1), rA2 = tid >> log2CA2;
                cpx<T> x[1][E][1];
                FFT_WAIT_VM0();
                FFT_SYNC_LDS();
                ev();  // A: half tile landed
                const bool more = (t + 1 < NT);
                team_all_stages<T, E>(x, land, work, twA, rA2, jA2, log2CA2, log2TPCA2, log2H1, [&](int s, int total) {
                    if (more) {
                        if (s == 0) dma_half_tile(inb, t + 1, 0, total >= 2 ? FFT_TEAM_DMA_FIRST(NCH) : NCH);
                        if (s == 1) dma_half_tile(inb, t + 1, FFT_TEAM_DMA_FIRST(NCH), NCH);
                    }
                }, p.inverse != 0);
                if ((t & 1) == 0) {
                    FFT_UNROLL
                    for (int e = 0; e < E; e++) ye[e] = x[0][e][0];
                } else {
                    // join the halves, apply W_n^(k1 n2), hand the low rows over, keep the high rows
                    const cpx<T>* t0 = tab + p.o_t0;
                    const cpx<T>* t1 = tab + p.o_t1;
                    const unsigned m0 = (1u << p.t0_bits) - 1u;
                    const unsigned n2 = (unsigned)(column_group(t >> 1) << log2CA2) + (unsigned)jA2;
                    const unsigned mh = n2 << log2H1;  // W_n^(L1/2 * n2): what the high rows' twiddle has on top of the low rows'
                    const cpx<T> w_half = cmul(t0[mh & m0], t1[mh >> p.t0_bits]);
                    cpx<T> lo[E];
                    FFT_UNROLL
                    for (int e = 0; e < E; e++) {
                        const unsigned kk = (unsigned)(rA2 + (e << log2TPCA2));
                        const unsigned mj = kk << log2L2;  // W_L1^k' = W_n^(k' L2)
                        const cpx<T> tq = cmul(x[0][e][0], cmul(t0[mj & m0], t1[mj >> p.t0_bits]));
                        const unsigned m = kk * n2;
                        const cpx<T> w = cmul(t0[m & m0], t1[m >> p.t0_bits]);
                        lo[e] = cmul(cadd(ye[e], tq), w);
                        const cpx<T> hi = cmul(csub(ye[e], tq), cmul(w, w_half));
                        FFT_UNROLL
                        for (int gg = 0; gg < NT / 2; gg++)
                            if ((t >> 1) == gg) keep2[gg][e] = hi;
                    }
                    if (t == 1 && g0 > 1) wait_all(g0 - 1);  // the windows' last readers (previous transform) are done
                    FFT_UNROLL
                    for (int ph = 0; ph < 2; ph++) {
                        cpx<T> y[E / 2];
                        FFT_UNROLL
                        for (int k = 0; k < E / 2; k++) y[k] = lo[ph * (E / 2) + k];
                        team_hand_over<T, E / 2>(sbase + (size_t)ph * phase_bytes, y, (int)n2, rA2, log2TPCA2, log2CB, tile_bytes,
                                                 1 << log2CA2);
                    }
                }
                ev();  // A: half tile transformed
            }
        } else {
        FFT_NOUNROLL
        for (int t = 0; t < NT; t++) {
            int tid = tid_invariant;
            FFT_OPAQUE(tid);
            const int jA = tid & ((1 << log2CA) - 1), rA = tid >> log2CA;
            cpx<T> x[1][E][1];
            FFT_WAIT_VM0();   // my part of the tile has landed ...
            FFT_SYNC_LDS();   // ... everybody's has; the work image is free (previous tile's last stage has read it)
            ev();  // A: tile landed
            const bool more = (t + 1 < NT);
            if (!FFT_ABLATE(p.ablate & 2)) {
                team_all_stages<T, E>(x, land, work, twA, rA, jA, log2CA, log2TPCA, log2L1, [&](int s, int total) {
                    // the next tile flies during the remaining stages; issued in two halves (a CU's memory queue is short)
                    if (more) {
                        if (s == 0) dma_column_tile(inb, t + 1, 0, total >= 2 ? FFT_TEAM_DMA_FIRST(NCH) : NCH);
                        if (s == 1) dma_column_tile(inb, t + 1, FFT_TEAM_DMA_FIRST(NCH), NCH);
                    }
                }, p.inverse != 0);  // inverse = forward transform between two re<->im swaps: first one here
            } else if (more) {
                FFT_SYNC_LDS();
                dma_column_tile(inb, t + 1, 0, NCH);
            }
            if (!FFT_ABLATE(p.ablate & 1)) {  // W_n^(k1 n2), two-level LDS table
                const unsigned n2 = (unsigned)(column_block(t) << log2CA) + (unsigned)jA;
                team_interpass_twiddle<T, E>(x, tab + p.o_t0, tab + p.o_t1, p.t0_bits, (unsigned)rA * n2, n2 << log2TPCA);
            }
            // windows S0, S1 were last read by the previous transform's final phases: everybody is past them?
            if (t == 0 && g0 > 1) wait_all(g0 - 1);
            FFT_UNROLL
            for (int ph = 0; ph < (NT < 2 ? NT : 2); ph++) {
                cpx<T> y[EP];
                FFT_UNROLL
                for (int ee = 0; ee < EP; ee++) y[ee] = x[0][ph * EP + ee][0];
                hand_over(sbase + (size_t)ph * phase_bytes, y, t, rA, jA);
            }
            if constexpr (NK > 0) {
                FFT_UNROLL
                for (int tt = 0; tt < NT; tt++) {
                    if (t == tt) {
                        FFT_UNROLL
                        for (int k = 0; k < NK * EP; k++) keep[tt][k] = x[0][2 * EP + k][0];
                    }
                }
            }
            ev();  // A: tile transformed, phases 0/1 hand-over issued
        }
        }
        FFT_WAIT_VM0();
        FFT_SYNC_LDS();
        arrive(g0);  // a0
        ev();        // A: hand-over of phases 0, 1 in L2

        // ================= step B: NT phases of (row FFTs of length L2, transposed store)
        wait_all(g0);
        ev();  // B: everybody's hand-over of phases 0, 1 in L2
        dma_row_tile(sbase);
        FFT_WAIT_VM0();
        FFT_SYNC_LDS();
        arrive(g0 + 1);  // a1
        ev();  // B: row tile 0 landed
        FFT_NOUNROLL
        for (int ph = 0; ph < NT; ph++) {
            int tid = tid_invariant;
            FFT_OPAQUE(tid);
            const int jA = tid & ((1 << log2CA) - 1), rA = tid >> log2CA;
            const int jB = tid & ((1 << log2CB) - 1), rB = tid >> log2CB;
            cpx<T> x[1][E][1];
            const bool next_transform = (ph == NT - 1) && (b + n_teams < p.nb);
            auto traffic = [&](int s, int total) __attribute__((always_inline)) {
                if (s == 0) {  // the landing image is free
                    if (ph + 1 < NT) {
                        if (ph >= 1) wait_all(g0 + ph + 1);  // everybody's hand-over of phase ph+1 (a0 covers phase 1)
                        dma_row_tile(sbase + (size_t)((ph + 1) & 1) * phase_bytes);
                    } else if (next_transform) {
                        dma_first_tile(inb + (long long)n_teams * n);
                    }
                }
                // hand-over of phase ph+2: as early as its wait allows, so that the stores are long in L2 when the phase
                // closes -- phase 0 gives the others one more stage to report their read of phase 0 (a1)
                if (s == ((total >= 2 && ph == 0 && !FFT_ABLATE(p.tune & 1)) ? 1 : 0)) {
                    if constexpr (NK > 0) {
                        if (ph + 2 < NT) {  // hand over phase ph+2 into the window phase ph was read from
                            if (ph == 0) wait_all(g0 + 1);  // everybody has read phase 0 (phase 1: known since the wait above)
                            if constexpr (ASPLIT) {
                                const int jA2 = tid & ((1 << log2CA2) - 1), rA2 = tid >> log2CA2;
                                const bool second = (ph == 1);  // phase 3 = the upper half of the kept rows
                                FFT_UNROLL
                                for (int gg = 0; gg < NT / 2; gg++) {
                                    cpx<T> y[E / 2];
                                    FFT_UNROLL
                                    for (int k = 0; k < E / 2; k++) {  // value by value (see pair_rows)
                                        const cpx<T> a = keep2[gg][k], b2 = keep2[gg][E / 2 + k];
                                        y[k] = mk<T>(second ? b2.re : a.re, second ? b2.im : a.im);
                                    }
                                    team_hand_over<T, E / 2>(sbase + (size_t)(ph & 1) * phase_bytes, y,
                                                             (column_group(gg) << log2CA2) + jA2, rA2, log2TPCA2, log2CB, tile_bytes,
                                                             1 << log2CA2);
                                }
                            } else {
                                FFT_UNROLL
                                for (int pp = 0; pp < NK; pp++) {
                                    if (ph == pp) {
                                        FFT_UNROLL
                                        for (int tt = 0; tt < NT; tt++) {
                                            cpx<T> y[EP];
                                            FFT_UNROLL
                                            for (int ee = 0; ee < EP; ee++) y[ee] = keep[tt][pp * EP + ee];
                                            hand_over(sbase + (size_t)(ph & 1) * phase_bytes, y, tt, rA, jA);
                                        }
                                    }
                                }
                            }
                        }
                    }
                }
            };
            if (!FFT_ABLATE(p.ablate & 2)) {
                team_all_stages<T, E>(x, land, work, twB, rB, jB, log2CB, log2TPCB, log2L2, traffic, false);
            } else {
                FFT_SYNC_LDS();
                traffic(0, 2);
                traffic(1, 2);
            }
            if (next_transform) have_first = true;
            ev();  // B: rows transformed
            // X[k1 + L1*k2]: slot e holds k2 = rB + TPCB*e of row k1 = ph*L1/NT + c*CB + jB.  fp32: the lanes of rows
            // jB, jB+1 pair up (16-byte stores, as in the hand-over)
            const long long k1 = ((long long)ph << (log2L1 - log2NT)) + ((long long)c << log2CB);
            if (p.inverse) {
                FFT_UNROLL
                for (int e = 0; e < E; e++) x[0][e][0] = cswap(x[0][e][0]);
            }
            if (p.scale != (T)1) {
                FFT_UNROLL
                for (int e = 0; e < E; e++) x[0][e][0] = cscale(x[0][e][0], p.scale);
            }
            if constexpr (V16 == 2) {
                const bool odd = (jB & 1) != 0;
                vec16<T> v[E / 2];
                FFT_UNROLL
                for (int q = 0; q < E / 2; q++) pair_rows<T>(x[0][2 * q][0], x[0][2 * q + 1][0], odd, 1, v[q]);
                vec16<T>* const dst0 = reinterpret_cast<vec16<T>*>(outb + ((long long)(rB + ((odd ? 1 : 0) << log2TPCB)) << log2L1) + k1 + (jB & ~1));
                const long long dstep = (2ll << (log2TPCB + log2L1)) / V16;  // two slots further on, in 16-byte units
                if (FFT_ABLATE(p.ablate & 4)) {  // profiling: no result stream
                } else if (p.nt_mask & 2) {
                    FFT_UNROLL
                    for (int q = 0; q < E / 2; q++) FFT_STORE16_NT(dst0 + q * dstep, v[q]);
                } else {
                    FFT_UNROLL
                    for (int q = 0; q < E / 2; q++) dst0[q * dstep] = v[q];
                }
            } else {
                FFT_UNROLL
                for (int e = 0; e < E; e++) {
                    const long long K = rB + (e << log2TPCB);
                    *reinterpret_cast<cpx<T>*>(outb + (K << log2L1) + k1 + jB) = x[0][e][0];
                }
            }
            ev();  // B: result stores issued
            if (ph + 1 < NT) {
                // everything but my NCH result stores is complete: the next row tile has landed, my hand-over of phase
                // ph+2 is in L2
                FFT_WAIT_VM_LE(NCH);
                FFT_SYNC_LDS();
                arrive(g0 + ph + 2);  // a(ph+2)
                ev();  // B: phase closed
            }
        }
    }
}

}  // namespace fftk
