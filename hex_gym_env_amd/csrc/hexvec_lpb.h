// hexvec_lpb.h -- lane-per-board ("LPB") kernels: the layout of HEXVEC_F_ROLLOUT_LAYOUT handles
// (the device-resident rollouts; every other entry point works on it too).
//
// Why a second kernel family: with one wavefront per board every index
// computation of a ply is wave-uniform and lands on the CU's single scalar
// ALU; rocprofv3 shows that kernel at 0.94 SALU instructions per CU-cycle
// (profiles/r01) -- scalar-issue bound at 6 % of the HBM roofline.  Here ONE
// LANE owns one board (64 boards per wavefront), so the same work is VALU at
// 64 boards per instruction, and what has to be indexed per board lives in LDS:
//
//   UF[p][board][PS]  one byte per cell and colour p, in that colour's OWN
//                     frame (BLACK: y*N+x, WHITE: x*N+y, so both connect
//                     row 0 with row N-1 and the hex neighbourhood is the same
//                     six offsets).  0 = no stone of p; 1..C = child, parent
//                     index + 1; 0xE0 | rank<<2 | f = root, f bit0/bit1 = its
//                     group touches row 0 / row N-1.  A new stone reads its 6
//                     neighbours, keeps ONE cell per run of p-stones around that
//                     ring (ring-adjacent stones already share a root; at most
//                     three groups can meet at a hex cell), chases those <= 3
//                     chains to their roots (all chains of all 64 boards in
//                     lockstep), hangs itself and the lower-rank roots under the
//                     highest-rank root (union by rank keeps the chains short) and
//                     ORs the flags; the mover wins iff the flags become 3.  That
//                     is plain connectivity, which SURVEY 4 verified equals the
//                     reference's label merging (HexGame.py:124-142 + :101-105).
//   IMG[v][board][C]  the observation bytes themselves, packed exactly as the
//                     [E, N*N] output rows.  Encoding B shows the board in the
//                     mover's perspective (HexSingleGame.py:265-271: BLACK to move
//                     sees F, WHITE to move sees -F^T).  Two-sided handles keep both
//                     views (VIEWS = 2: v=0 BLACK-to-move, v=1 WHITE-to-move; a ply
//                     writes one byte in each); self-play handles keep ONE image per
//                     board, in the view of that board's AGENT (VIEWS = 1) -- the only
//                     view a self-play step ever returns; the rare other view (the
//                     opponent's, in the split step / a terminal observation) is read
//                     out of it as the negated transpose.  That halves the image (13x13 and
//                     14x14 at 65 536 boards fit the LDS in one pass).  Emitting obs + legal
//                     mask for 64 boards is a wave-cooperative copy LDS -> HBM with
//                     16-byte loads and stores of whole contiguous spans.
//
// Empty-cell bitboards (one per frame) stay in registers for the uniform
// random policy (k-th set bit) and the legality test.  Scalars (action,
// reward, done) are one coalesced store per wave (in the two rollouts the helper
// wavefront stores them together with the observation rows, and paces its stores where the
// write path would otherwise collapse under the bursts of 1 024 such wavefronts: Lpb::finish).
//
// Scope: handles with auto reset.  Then no ply is ever applied after the end
// of a game and an illegal action ends the episode at once, so the literal
// quirks (q1/q2 of SURVEY 8a) cannot be observed; handles created with
// HEXVEC_F_NO_AUTO_RESET or HEXVEC_F_EXACT_LABELS use the wave-per-board
// kernels, which keep the reference's label planes bit for bit.
// Included by hexvec_lpb64.hip with HV_BPW = 64 boards per wavefront (every lane owns a board;
// 32 was measured in round 1: same time, the per-lane part doubles) and HV_LPB_NAME(x) = lpb64_##x.
#pragma once
#include <stdlib.h>

#include "hexvec_common.h"

// Diagnostic build (-DHV_STAMPS, tools/stamp_rollout.py): s_memtime stamps around the sections of a
// self-play step (and inside place()), summed per playing wavefront into a buffer of their own.  No
// stamp exists in the product build.
#ifdef HV_STAMPS
__device__ unsigned long long g_stamps[4096][12];
__device__ unsigned long long g_phase[4096][4]; // s_memrealtime (100 MHz, one clock for the whole chip) at kernel entry / state in LDS / before the store-back / stores retired
#define HV_PHASE(blk, k)                                                                   \
    do {                                                                                   \
        __builtin_amdgcn_s_waitcnt(0);                                                     \
        if (lane == 0 && (blk) < 4096) g_phase[blk][k] = __builtin_amdgcn_s_memrealtime(); \
    } while (0)
struct StampBox {
    unsigned long long acc[4], prev;
};
#define HV_STAMP_INIT()                                                                               \
    unsigned long long st_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, st_prev = __builtin_amdgcn_s_memtime(); \
    g_box.acc[0] = g_box.acc[1] = g_box.acc[2] = 0;                                                   \
    g_box.acc[3] = __builtin_amdgcn_s_memrealtime() /* workgroup start, 100 MHz */;                   \
    st_acc[7] = (unsigned long long)__builtin_amdgcn_s_getreg((4 << 0) | (0 << 6) | (31 << 11)) /* HW_ID */;     \
    st_acc[3] = (unsigned long long)__builtin_amdgcn_s_getreg((20 << 0) | (0 << 6) | (31 << 11)) /* XCC_ID */
#define HV_MARK(k)                                                      \
    do {                                                                \
        const unsigned long long st_now = __builtin_amdgcn_s_memtime(); \
        st_acc[k] += st_now - st_prev;                                  \
        st_prev = st_now;                                               \
    } while (0)
#define HV_PMARK0(box) (box).prev = __builtin_amdgcn_s_memtime()
#define HV_PMARK(box, k)                                                \
    do {                                                                \
        const unsigned long long st_now = __builtin_amdgcn_s_memtime(); \
        (box).acc[k] += st_now - (box).prev;                            \
        (box).prev = st_now;                                            \
    } while (0)
#define HV_STAMP_FLUSH(blk)                                                                                  \
    do {                                                                                                     \
        if (lane == 0 && (blk) < 4096)                                                                       \
            for (int k_ = 0; k_ < 12; ++k_) g_stamps[blk][k_] = k_ < 8 ? st_acc[k_] : g_box.acc[k_ - 8];     \
    } while (0)
#else
struct StampBox {
};
#define HV_PHASE(blk, k) do { } while (0)
#define HV_STAMP_INIT() do { } while (0)
#define HV_MARK(k) do { } while (0)
#define HV_PMARK0(box) do { } while (0)
#define HV_PMARK(box, k) do { } while (0)
#define HV_STAMP_FLUSH(blk) do { } while (0)
#endif

namespace hv {
namespace lpb {

template <int N, int BPW> struct Geo {
    static constexpr int C = N * N;
    static constexpr int JW = (C + 31) / 32; // bitboard words
    static constexpr int PS0 = (C + 2 + 3) / 4 * 4;
    static constexpr int PS = ((PS0 / 4) % 2 == 0) ? PS0 + 4 : PS0; // odd dword stride: lanes spread over banks
    static constexpr int DUMMY = C;     // always 0: stands in for off-board neighbours
    static constexpr int TRASH = C + 1; // absorbs predicated-off writes
    static constexpr int UF_BYTES = 2 * BPW * PS;
    static constexpr int IMG_BYTES = BPW * C; // per view; multiple of 16
    static constexpr int NIT = (BPW * C / 16 + 63) / 64; // 16-byte vectors per view / 64 lanes
};

// per-lane registers persisted between launches: [REGW][64] words per block
template <int N> struct Regs {
    static constexpr int JW = (N * N + 31) / 32;
    static constexpr int REGW = 2 * JW + 8;
};

template <int N, int ENC, int BPW, int VIEWS> struct Lpb {
    using G = Geo<N, BPW>;
    static constexpr bool B = ENC == 1;
    static constexpr int NV = B ? VIEWS : 1;
    static constexpr bool TWOV = B && VIEWS == 2; // both views kept
    static constexpr bool ONE = B && VIEWS == 1;  // one image per board, in its agent's view
    static constexpr int C = G::C, JW = G::JW, PS = G::PS;
    static constexpr int IMG = G::UF_BYTES; // byte offset of IMG[0] in LDS
    static constexpr int LDS_BYTES = G::UF_BYTES + NV * G::IMG_BYTES;
    static constexpr unsigned char EMPTY_BYTE = B ? 0 : 2;

    struct LS {
        u32 E0[JW], E1[JW]; // empty cells in F order / FT order
        int cur;            // player to move (B: also the frame of the stored board)
        int done;
        int winner;
        int n_empty;
        int agent;
        int pend; // split step: 0 agent to move, 1 the opponent owes its reply, 2 its opening
        int prewon; // bit p: colour p was already connected in the start position (set_boards); the
                    // reference notices that on p's next ply, whatever it is (HexGame.py:101-105)
        u32 ctr, key;
        u32 kc; // key + ctr * 0x9E3779B9, kept in step with ctr (no multiply per draw)
        int ep_ret, ep_len;
        // episode sums since create, LOW words (sr biased by 2^31 so that a return sum around zero does not
        // carry).  The high words live behind the block's register rows in HBM and are touched only by the
        // lane whose low word carries (sum_len passes 2^31 after ~2 h of rollouts): add_sums().
        u32 sr, sl, se, sw;
    };
    static constexpr u32 SR_BIAS = 0x80000000u;
    // finished episode -> the 64-bit sums; hi = the block's [4][64] high words
    static __device__ __forceinline__ void add_sums(LS &s, int won, u32 *hi, int lane)
    {
        const u32 o_sr = s.sr, o_sl = s.sl;
        s.sr += (u32)s.ep_ret;
        s.sl += (u32)s.ep_len;
        s.se += 1u;
        s.sw += (u32)won;
        const int d_sr = (int)(s.sr < o_sr) - (int)(s.ep_ret < 0); // carry out of the low word - sign extension
        const bool c_sl = s.sl < o_sl, c_se = s.se == 0u, c_sw = won && s.sw == 0u;
        if (d_sr != 0 || c_sl || c_se || c_sw) { // once per 2^31 steps of a board
            hi[0 * 64 + lane] += (u32)d_sr;
            hi[1 * 64 + lane] += c_sl ? 1u : 0u;
            hi[2 * 64 + lane] += c_se ? 1u : 0u;
            hi[3 * 64 + lane] += c_sw ? 1u : 0u;
        }
    }

    static __device__ __forceinline__ int transpose(int a)
    {
        const int y = a / N;
        return (a - y * N) * N + y;
    }
    static __device__ __forceinline__ void clear_bit(u32 (&E)[JW], int idx)
    {
        const u32 bit = 1u << (idx & 31);
        const int w = idx >> 5;
#pragma unroll
        for (int j = 0; j < JW; ++j) E[j] &= ~(w == j ? bit : 0u);
    }
    static __device__ __forceinline__ bool test_bit(const u32 (&E)[JW], int idx)
    {
        const int w = idx >> 5;
        u32 word = 0;
#pragma unroll
        for (int j = 0; j < JW; ++j) word = w == j ? E[j] : word;
        return (word >> (idx & 31)) & 1u;
    }
    static __device__ __forceinline__ void fill_bits(u32 (&E)[JW])
    {
#pragma unroll
        for (int j = 0; j < JW; ++j) {
            const int left = C - 32 * j;
            E[j] = left >= 32 ? 0xffffffffu : ((1u << left) - 1u);
        }
    }
    // position of the k-th set bit (k < popcount), all per lane
    static __device__ __forceinline__ int kth_set(const u32 (&E)[JW], u32 k)
    {
        // the word that holds it: running popcounts (the popcount instruction accumulates), then one
        // select per word boundary -- k < popcount of all words, so the last word needs no test
        u32 word = E[0], base = 0;
        int widx = 0;
        u32 c = 0;
#pragma unroll
        for (int j = 0; j + 1 < JW; ++j) {
            c += (u32)__popc(E[j]); // cells in words 0..j
            const bool past = k >= c;
            word = past ? E[j + 1] : word;
            base = past ? c : base;
            widx = past ? j + 1 : widx;
        }
        k -= base;
        int pos = 0;
#pragma unroll
        for (int sh = 16; sh >= 1; sh >>= 1) {
            const u32 lowmask = (1u << sh) - 1u;
            const u32 cl = (u32)__popc(word & lowmask);
            const bool up = k >= cl;
            k = up ? k - cl : k;
            word = up ? word >> sh : word;
            pos += up ? sh : 0;
        }
        return widx * 32 + pos;
    }

    static __device__ __forceinline__ u32 draw(LS &s)
    {
        const u32 r = mix32(s.kc);
        s.kc += 0x9E3779B9U;
        s.ctr += 1;
        return r;
    }
    // Work that place() runs in the shadow of its LDS round trips (the playing wavefront is alone on its SIMD: what
    // it does not overlap itself is exposed).  NoFill: nothing.  NextDraw: the action of whoever moves NEXT, drawn
    // speculatively from the board as this stone leaves it -- a() once the stone's bits are cleared (covers the six
    // neighbour reads), b() behind the first hop of the chase.  The draw is not consumed here: the caller commits it
    // (commit_draw) only where that player really moves next, so the per-board draw sequence stays the plain one.
    struct NoFill {
        static constexpr bool ON = false;
        __device__ __forceinline__ void a(const LS &) {}
        __device__ __forceinline__ void b() {}
    };
    struct NextDraw {
        static constexpr bool ON = true;
        int f; // frame of the next mover's action (0: F order, 1: FT order)
        u32 k, word;
        int widx, act;
        __device__ __forceinline__ void a(const LS &s)
        { // the word that holds the k-th empty cell (kth_set, first half)
            const u32 r = mix32(s.kc);
            k = s.n_empty > 0 ? __umulhi(r, (u32)s.n_empty) : 0u;
            u32 base = 0, c = 0;
            word = f ? s.E1[0] : s.E0[0];
            widx = 0;
#pragma unroll
            for (int j = 0; j + 1 < JW; ++j) {
                c += (u32)__popc(f ? s.E1[j] : s.E0[j]);
                const bool past = k >= c;
                word = past ? (f ? s.E1[j + 1] : s.E0[j + 1]) : word;
                base = past ? c : base;
                widx = past ? j + 1 : widx;
            }
            k -= base;
        }
        __device__ __forceinline__ void b()
        { // its position in that word (kth_set, second half)
            int pos = 0;
            u32 kk = k, w = word;
#pragma unroll
            for (int sh = 16; sh >= 1; sh >>= 1) {
                const u32 cl = (u32)__popc(w & ((1u << sh) - 1u));
                const bool up = kk >= cl;
                kk = up ? kk - cl : kk;
                w = up ? w >> sh : w;
                pos += up ? sh : 0;
            }
            act = widx * 32 + pos;
        }
    };
    static __device__ __forceinline__ void commit_draw(LS &s)
    {
        s.kc += 0x9E3779B9U;
        s.ctr += 1;
    }
    // uniform-legal action in frame f's row-major order (SelfplayWrapper.py:16-22 shape)
    static __device__ __forceinline__ int rng_action(LS &s, int f)
    {
        const u32 r = draw(s);
        if (s.n_empty <= 0) return 0;
        const u32 k = __umulhi(r, (u32)s.n_empty);
        u32 Ef[JW];
#pragma unroll
        for (int j = 0; j < JW; ++j) Ef[j] = f ? s.E1[j] : s.E0[j];
        return kth_set(Ef, k);
    }

    static __device__ __forceinline__ void fresh_regs(LS &s, int first)
    {
        fill_bits(s.E0);
        fill_bits(s.E1);
        s.cur = first;
        s.done = 0;
        s.n_empty = C;
        s.ep_ret = 0;
        s.ep_len = 0;
        s.prewon = 0;
    }

    // the image byte(s) of a new stone of colour p
    static __device__ __forceinline__ void put_stone(unsigned char *lds, const LS &s, int idx0, int idx1, int p, int lane)
    {
        if (TWOV) {
            lds[IMG + lane * C + idx0] = p ? 1 : 0xFF;                 // view 0:  F, BLACK -1 / WHITE +1
            lds[IMG + G::IMG_BYTES + lane * C + idx1] = p ? 0xFF : 1;  // view 1: -F^T
        } else if (ONE) { // the agent's view: its own stones read -1
            lds[IMG + lane * C + (s.agent ? idx1 : idx0)] = p == s.agent ? 0xFF : 1;
        } else {
            lds[IMG + lane * C + idx0] = (unsigned char)p;             // absolute 0 / 1 (EMPTY 2)
        }
    }

    // One legal stone of colour p at idx0 (F order) == idx1 (FT order).  Returns
    // true iff p now connects its two edges (HexGame.py:98-105 via connectivity).
    template <class Fill>
    static __device__ __forceinline__ bool place(unsigned char *lds, LS &s, int idx0, int idx1, int p, int y, int x,
                                                 int lane, StampBox &box, Fill &fill)
    { // (y, x): coordinates of the stone in p's own frame
        HV_PMARK0(box);
        // union-find in p's own frame: issue the six neighbour reads first, their LDS latency is
        // covered by the bitboard / image updates below
        unsigned char *uf = lds + (p * BPW + lane) * PS;
        const int a = p ? idx1 : idx0;
        const bool up = y > 0, dn = y < N - 1, lf = x > 0, rt = x < N - 1;
        // the six neighbours in RING order (each is adjacent to the next, the last to the first):
        // up, up-right, right, down, down-left, left
        int nb[6];
        nb[0] = up ? a - N : G::DUMMY;
        nb[1] = (up && rt) ? a - N + 1 : G::DUMMY;
        nb[2] = rt ? a + 1 : G::DUMMY;
        nb[3] = dn ? a + N : G::DUMMY;
        nb[4] = (dn && lf) ? a + N - 1 : G::DUMMY;
        nb[5] = lf ? a - 1 : G::DUMMY;
        u32 e6[6];
#pragma unroll
        for (int i = 0; i < 6; ++i) e6[i] = uf[nb[i]];
        clear_bit(s.E0, idx0);
        clear_bit(s.E1, idx1);
        s.n_empty -= 1;
        put_stone(lds, s, idx0, idx1, p, lane);
        fill.a(s);
        HV_PMARK(box, 0); // neighbour reads issued, bitboards / image updated
        // Stones of p on consecutive ring cells touch each other, so they already share a root: ONE cell
        // per run of p-stones around the ring represents its group, and there are at most three runs.
        // Two run starts are never ring-adjacent, so each ring pair (0,1) (2,3) (4,5) holds at most one:
        // three chains to chase and three roots to unite instead of six.
        bool own[6];
#pragma unroll
        for (int i = 0; i < 6; ++i) own[i] = e6[i] != 0u;
        const bool ring_full = own[0] && own[1] && own[2] && own[3] && own[4] && own[5]; // one run, no start
        int ix[3];
        u32 e[3];
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const bool s0 = (own[2 * k] && !own[(2 * k + 5) % 6]) || (k == 0 && ring_full);
            const bool s1 = own[2 * k + 1] && !own[2 * k];
            ix[k] = s1 ? nb[2 * k + 1] : (s0 ? nb[2 * k] : G::DUMMY);
            e[k] = s1 ? e6[2 * k + 1] : (s0 ? e6[2 * k] : 0u);
        }
        // follow child pointers to the roots (all chains of all lanes together)
        if (Fill::ON) { // the first hop is issued unconditionally (a lane without a child pointer re-reads its cell)
#pragma unroll
            for (int i = 0; i < 3; ++i) {
                ix[i] = (e[i] - 1u) < 0xDFu ? (int)e[i] - 1 : ix[i];
                e[i] = uf[ix[i]];
            }
            fill.b();
        }
        for (;;) {
            bool any = false;
            bool ch[3];
#pragma unroll
            for (int i = 0; i < 3; ++i) {
                ch[i] = (e[i] - 1u) < 0xDFu; // 1..0xDF: child
                any = any || ch[i];
            }
            if (!__ballot(any)) break;
#pragma unroll
            for (int i = 0; i < 3; ++i) {
                ix[i] = ch[i] ? (int)e[i] - 1 : ix[i];
                e[i] = uf[ix[i]];
            }
        }
        HV_PMARK(box, 1); // chase
        // union by rank: everything hangs under the highest-rank root.  After the chase every e[i]
        // is 0 (no group) or a root byte 0xE0|rank<<2|flags; key = (rank+1)<<8 | root index.
        u32 fl = (y == 0 ? 1u : 0u) | (y == N - 1 ? 2u : 0u);
        u32 key[3];
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            fl |= e[i] & 3u;
            key[i] = e[i] != 0u ? ((((e[i] >> 2) & 7u) + 1u) << 8) | (u32)ix[i] : 0u;
        }
        const u32 best = max(max(key[0], key[1]), key[2]);
        const bool has = best != 0u;
        const int best_ix = has ? (int)(best & 0xffu) : G::TRASH;
        u32 second = 0u; // highest key among the OTHER roots
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            const bool other = key[i] != 0u && (key[i] & 0xffu) != (best & 0xffu);
            second = max(second, other ? key[i] : 0u);
            uf[other ? ix[i] : G::TRASH] = (unsigned char)(best_ix + 1);
        }
        const u32 rank = has ? (best >> 8) - 1u + ((second >> 8) == (best >> 8) ? 1u : 0u) : 0u;
        const u32 root_byte = 0xE0u | (rank << 2) | fl;
        uf[a] = has ? (unsigned char)(best_ix + 1) : (unsigned char)root_byte;
        uf[best_ix] = (unsigned char)root_byte; // TRASH when the stone has no neighbour of its colour
        HV_PMARK(box, 2); // union
        return fl == 3u;
    }

    // wave-cooperative: wipe the LDS rows of the boards in `mask`
    static __device__ __forceinline__ void clear_boards(unsigned char *lds, u64 mask, int lane)
    {
        while (mask) {
            const int d = __ffsll((long long)mask) - 1;
            mask &= mask - 1;
#pragma unroll
            for (int p = 0; p < 2; ++p)
                if (lane < PS / 4) reinterpret_cast<u32 *>(lds + (p * BPW + d) * PS)[lane] = 0u;
#pragma unroll
            for (int v = 0; v < NV; ++v)
                for (int i = lane; i < C; i += 64) lds[IMG + v * G::IMG_BYTES + d * C + i] = EMPTY_BYTE;
        }
    }

    // wave-cooperative: rows of the boards in `mask` -> dst[(board) * C ...] (terminal observations)
    // (agent1: bit d = board d's agent plays WHITE = the view its single image is kept in)
    static __device__ __forceinline__ void copy_rows(const unsigned char *lds, u64 mask, u64 view1, u64 agent1,
                                                     signed char *dst, int lane)
    {
        while (mask) {
            const int d = __ffsll((long long)mask) - 1;
            mask &= mask - 1;
            if (ONE) {
                const bool flip = ((view1 ^ agent1) >> d) & 1; // the other view: negated transpose
                for (int i = lane; i < C; i += 64) {
                    u32 x = lds[IMG + d * C + (flip ? transpose(i) : i)];
                    x = (flip && x) ? x ^ 0xFEu : x;
                    dst[(size_t)d * C + i] = (signed char)x;
                }
            } else {
                const int v = TWOV ? (int)((view1 >> d) & 1) : 0;
                for (int i = lane; i < C; i += 64)
                    dst[(size_t)d * C + i] = (signed char)lds[IMG + v * G::IMG_BYTES + d * C + i];
            }
        }
    }

    // view merge + legal mask of one 16-byte vector (r0 / r1 = the two views' bytes)
    static __device__ __forceinline__ void merge_vec(u64 view1, int byte0, const uint4 &r0, const uint4 &r1,
                                                     uint4 &o, uint4 &l)
    {
        o = r0;
        if (ONE) { // the image is the view: values 0x00 / 0x01 / 0xFF, empty <=> bit 0 clear
            l = make_uint4(~o.x & 0x01010101u, ~o.y & 0x01010101u, ~o.z & 0x01010101u, ~o.w & 0x01010101u);
        } else if (B) {
            const int b0 = byte0 / C;
            const int s = C - (byte0 - b0 * C); // bytes [0,s) of this vector belong to board b0
            const u32 ph = (u32)(view1 >> b0);
            const u32 m0 = 0u - (ph & 1u), m1 = 0u - ((ph >> 1) & 1u);
            u32 oo[4] = {r0.x, r0.y, r0.z, r0.w};
            const u32 rr[4] = {r1.x, r1.y, r1.z, r1.w};
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int sh = (s - 4 * k) * 8;
                const u32 T = sh <= 0 ? 0xffffffffu : (sh >= 32 ? 0u : (0xffffffffu << sh)); // bytes >= s
                const u32 M = (m0 & ~T) | (m1 & T);                                           // bytes taken from view 1
                oo[k] = (rr[k] & M) | (oo[k] & ~M);
            }
            o = make_uint4(oo[0], oo[1], oo[2], oo[3]);
            // values 0x00 / 0x01 / 0xFF: empty <=> bit 0 clear
            l = make_uint4(~o.x & 0x01010101u, ~o.y & 0x01010101u, ~o.z & 0x01010101u, ~o.w & 0x01010101u);
        } else { // values 0 / 1 / 2: empty <=> bit 1 set
            l = make_uint4((o.x >> 1) & 0x01010101u, (o.y >> 1) & 0x01010101u, (o.z >> 1) & 0x01010101u,
                           (o.w >> 1) & 0x01010101u);
        }
    }
    static __device__ __forceinline__ void emit_vec(const unsigned char *lds, u64 view1, int byte0, uint4 &o, uint4 &l)
    {
        const uint4 r0 = *reinterpret_cast<const uint4 *>(lds + IMG + byte0);
        uint4 r1 = r0;
        if (TWOV) r1 = *reinterpret_cast<const uint4 *>(lds + IMG + G::IMG_BYTES + byte0);
        merge_vec(view1, byte0, r0, r1, o, l);
    }

    // tail blocks / unaligned rows: byte stores with bounds (cold, deliberately not unrolled)
    static __device__ __noinline__ void emit_slow(const unsigned char *lds, u64 view1, signed char *obs,
                                                  unsigned char *legal, int live_bytes, int lane)
    {
#pragma unroll 1
        for (int it = 0; it < G::NIT; ++it) {
            const int byte0 = (it * 64 + lane) * 16;
            if (byte0 >= G::IMG_BYTES || byte0 >= live_bytes) continue;
            uint4 o, l;
            emit_vec(lds, view1, byte0, o, l);
            const u32 ow[4] = {o.x, o.y, o.z, o.w}, lw[4] = {l.x, l.y, l.z, l.w};
#pragma unroll 1
            for (int k = 0; k < 16; ++k) {
                if (byte0 + k >= live_bytes) break;
                const u32 wo = k < 4 ? ow[0] : k < 8 ? ow[1] : k < 12 ? ow[2] : ow[3];
                const u32 wl = k < 4 ? lw[0] : k < 8 ? lw[1] : k < 12 ? lw[2] : lw[3];
                if (obs) obs[byte0 + k] = (signed char)((wo >> ((k & 3) * 8)) & 0xffu);
                if (legal) legal[byte0 + k] = (unsigned char)((wl >> ((k & 3) * 8)) & 0xffu);
            }
        }
    }

    // wave-cooperative: obs (+ legal mask) of the block's boards, board d in view bit d of `view1`.
    // All LDS reads of the span are issued first, then merged and stored as whole 1 KiB pieces;
    // only the last piece is partial (lanes past the span re-read the last vector and skip the store).
    struct Snap {
        uint4 r0[G::NIT], r1[G::NIT];
    };
    static __device__ __forceinline__ void snapshot(const unsigned char *lds, Snap &sn, int lane)
    {
#pragma unroll
        for (int it = 0; it < G::NIT; ++it) {
            const int byte0 = min((it * 64 + lane) * 16, G::IMG_BYTES - 16);
            sn.r0[it] = *reinterpret_cast<const uint4 *>(lds + IMG + byte0);
            if (TWOV) sn.r1[it] = *reinterpret_cast<const uint4 *>(lds + IMG + G::IMG_BYTES + byte0);
        }
    }
    // GAP > 0 (the helper wavefront where the store path bounds the launch): s_sleep GAP (64 GAP cycles) after every
    // 1 KB store instruction.  Sixteen to twenty-five such stores issued back to back by each of 1 024 wavefronts are
    // what the write path handles worst: saturated with bursts it sustains 5.1-5.3 TB/s on most placements of the output
    // buffers, with the same pieces ~340 cycles apart 6.0-6.5 TB/s on every placement (tools/micro/store_pattern.hip
    // shows both levels without any game logic).  The optimum is sharp -- s_sleep 4 does not get there and 6 makes
    // the helper the bound: 13x13 / 65 536 at 100 steps per launch 4.45 / 4.56 / 3.68 / 4.25 us per step for no gap /
    // 4 / 5 / 6 (tools/ab_inproc.py, same process, same buffers, every buffer set alike).
    template <bool OBS, bool LEGAL, int GAP = 0>
    static __device__ __forceinline__ void finish(const Snap &sn, u64 view1, signed char *obs, unsigned char *legal,
                                                  int lane)
    {
        // streamed once, never re-read by this kernel: non-temporal stores (plain stores: +13...+25 % per step at
        // 11x11 / 65 536).  All observation pieces first, then all mask pieces: 1-2 % faster than alternating between
        // the two arrays piece by piece (tools/ab_inproc.py, same process, same buffers, six placements).
        typedef u32 __attribute__((ext_vector_type(4))) v4u;
        uint4 o[G::NIT]; // the merged observation vectors (two-view boards: the view merge is done once, not per array)
#pragma unroll
        for (int it = 0; it < G::NIT; ++it) {
            const int byte0 = (it * 64 + lane) * 16;
            uint4 l;
            merge_vec(view1, min(byte0, G::IMG_BYTES - 16), sn.r0[it], TWOV ? sn.r1[it] : sn.r0[it], o[it], l);
            if (OBS && ((it + 1) * 64 * 16 <= G::IMG_BYTES || byte0 < G::IMG_BYTES)) {
                __builtin_nontemporal_store(v4u{o[it].x, o[it].y, o[it].z, o[it].w}, reinterpret_cast<v4u *>(obs + byte0));
                if (GAP > 0) __builtin_amdgcn_s_sleep(GAP);
            }
        }
#pragma unroll
        for (int it = 0; it < G::NIT; ++it) {
            const int byte0 = (it * 64 + lane) * 16;
            if (LEGAL && ((it + 1) * 64 * 16 <= G::IMG_BYTES || byte0 < G::IMG_BYTES)) {
                // the mask of a merged vector: enc B values 0x00 / 0x01 / 0xFF: empty <=> bit 0 clear; enc A values
                // 0 / 1 / 2: empty <=> bit 1 set (as merge_vec computes it)
                const uint4 &x = o[it];
                const v4u l = B ? v4u{~x.x & 0x01010101u, ~x.y & 0x01010101u, ~x.z & 0x01010101u, ~x.w & 0x01010101u}
                                : v4u{(x.x >> 1) & 0x01010101u, (x.y >> 1) & 0x01010101u, (x.z >> 1) & 0x01010101u, (x.w >> 1) & 0x01010101u};
                __builtin_nontemporal_store(l, reinterpret_cast<v4u *>(legal + byte0));
                if (GAP > 0) __builtin_amdgcn_s_sleep(GAP);
            }
        }
    }
    template <int GAP = 0>
    static __device__ __forceinline__ void finish_any(const Snap &sn, u64 view1, signed char *obs,
                                                      unsigned char *legal, int lane)
    {
        if (obs && legal) finish<true, true, GAP>(sn, view1, obs, legal, lane);
        else if (obs) finish<true, false, GAP>(sn, view1, obs, legal, lane);
        else finish<false, true, GAP>(sn, view1, obs, legal, lane);
    }
    // wave-cooperative: obs (+ legal mask) of the block's boards, board d in view bit d of `view1`.
    // All LDS reads of the span are issued first, then merged and stored as whole 1 KiB pieces;
    // only the last piece is partial (lanes past the span re-read the last vector and skip the store).
    // one-image boards asked for their OTHER view (flip bit set): bytes gathered through the transpose
    static __device__ __noinline__ void emit_gather(const unsigned char *lds, u64 flip, signed char *obs,
                                                    unsigned char *legal, int live_bytes, int lane)
    {
#pragma unroll 1
        for (int j = lane; j < live_bytes; j += 64) {
            const int b = j / C, i = j - b * C;
            const bool f = (flip >> b) & 1;
            u32 x = lds[IMG + b * C + (f ? transpose(i) : i)];
            x = (f && x) ? x ^ 0xFEu : x;
            if (obs) obs[j] = (signed char)x;
            if (legal) legal[j] = x == 0u;
        }
    }
    static __device__ __forceinline__ void emit(const unsigned char *lds, u64 view1, u64 agent1, signed char *obs,
                                                unsigned char *legal, int nlive, bool vec_ok, int lane)
    {
        if (!obs && !legal) return;
        if (ONE) {
            const u64 livem = nlive >= 64 ? ~0ull : ((1ull << nlive) - 1ull);
            const u64 flip = (view1 ^ agent1) & livem;
            if (flip) {
                emit_gather(lds, flip, obs, legal, nlive * C, lane);
                return;
            }
        }
        if (!(vec_ok && nlive == BPW)) {
            emit_slow(lds, view1, obs, legal, nlive * C, lane);
        } else {
            Snap sn;
            snapshot(lds, sn, lane);
            finish_any(sn, view1, obs, legal, lane);
        }
    }
};

// fast state in HBM, per block of 64 boards: [LDS image][REGW x 64 words][4 x 64 high words of the episode sums]
// (the last four rows are not part of what a launch loads and stores: Lpb::add_sums)
template <int N, int ENC, int BPW, int VIEWS> struct FastState {
    using L = Lpb<N, ENC, BPW, VIEWS>;
    static constexpr int REGW = Regs<N>::REGW;
    static constexpr size_t HI_OFF = (size_t)L::LDS_BYTES + (size_t)REGW * 64 * 4;
    static constexpr size_t BLOCK_BYTES = HI_OFF + (size_t)4 * 64 * 4;
};

// HELPER = 1: 128-thread workgroups.  Wave 0 plays; wave 1 only copies observations: at every emission
// point wave 0 publishes the view mask and meets wave 1 at a barrier, wave 1 snapshots the images into
// registers, a second barrier releases wave 0, and wave 1 merges and stores while wave 0 computes on.
// SPEC = 1, self-play rollout: the opponent's reply is drawn inside the agent's ply (Lpb::NextDraw).  Same
// results; which form is faster depends on what bounds the launch (lpb_launch2).  SPEC = 1, two-sided rollout:
// a launch on a full chip (the helper paces its stores by the table below).
template <int N, int ENC, int MODE, int BPW, int HELPER, int VIEWS, int SPEC = 0>
__global__ __launch_bounds__(64 * (1 + HELPER)) void k_lpb(KArgs a)
{
    // The plain form of the self-play rollout runs where the store path is the bound (lpb_launch2): from 12x12 on its
    // helper wavefront spaces its stores (Lpb::finish).  At 11x11 the spaced helper and the playing wavefront take the
    // same time (2.70-2.77 us per step on every placement against 2.58-2.98 by placement without the gaps: no gain on
    // average), below that the playing wavefront is the bound and the gaps only cost (+2 %).
    // The two-sided rollout (SPEC = 1: launched on a full chip) records every ply and is store-bound from 9x9 on.  Its
    // helper merges two views per vector, so it already spends time between its stores; what is left to add was measured
    // per board size (us per ply at 65 536 boards, 64 plies per launch, no gap / s_sleep 3 / 4 / 5, same process and
    // buffers): 9x9 1.78 / 1.73 / 2.04 / 2.36, 10x10 2.13 / 2.46 / 2.82 / 3.19, 11x11 2.97-3.23 / 3.03 / 2.64 / 3.05,
    // 12x12 3.64 / 3.01 / 3.47 / 3.93, 13x13 3.5-4.1 / 5.39 / 6.52 / 7.66, 14x14 4.65 / 7.56 / 8.95 / 10.3.
    constexpr int W2_GAP = N == 9 ? 3 : (N == 11 ? 4 : (N == 12 ? 3 : 0));
    constexpr int STORE_GAP = (MODE == M_ROLLOUT1 && SPEC == 0 && N >= 12) ? 5 : ((MODE == M_ROLLOUT2 && SPEC == 1) ? W2_GAP : 0);
    using L = Lpb<N, ENC, BPW, VIEWS>;
    using G = Geo<N, BPW>;
    using LS = typename L::LS;
    constexpr bool B = ENC == 1;
    constexpr int C = G::C, JW = G::JW;
    // (helper kernels: + the view mask of the emission and one word of scalar outputs per board)
    __shared__ __attribute__((aligned(16))) unsigned char lds[L::LDS_BYTES + (HELPER ? 16 + 256 : 0)];
    u64 *sh_view = reinterpret_cast<u64 *>(lds + L::LDS_BYTES);
    u32 *sh_scal = reinterpret_cast<u32 *>(lds + L::LDS_BYTES + 16);

    const int lane = (int)threadIdx.x & 63;
    const int wave = HELPER ? __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6) : 0;
    const int nb = (int)gridDim.x, bi = (int)blockIdx.x;
    const int blk = (nb & 7) == 0 ? (bi & 7) * (nb >> 3) + (bi >> 3) : bi; // contiguous env ranges per XCD
    const int e0 = blk * BPW;
    const int env = e0 + lane;
    const bool live = lane < BPW && env < a.E;
    const int nlive = min(BPW, a.E - e0);
    unsigned char *sblk = a.fast + (size_t)blk * FastState<N, ENC, BPW, VIEWS>::BLOCK_BYTES;
    u32 *rblk = reinterpret_cast<u32 *>(sblk + L::LDS_BYTES);
    u32 *hblk = reinterpret_cast<u32 *>(sblk + FastState<N, ENC, BPW, VIEWS>::HI_OFF);
    // (the self-play rollout is launched on self-play handles only, the two-sided one on two-sided handles: lpb_launch1)
    const bool two = MODE == M_ROLLOUT1 ? false : (MODE == M_ROLLOUT2 ? true : (a.flags & HEXVEC_F_TWO_SIDED) != 0);
    const bool selfplay = !two;
    const int first = (!B && (a.flags & HEXVEC_F_FIRST_WHITE)) ? 1 : 0;
    const bool vec_ok = ((size_t)a.E * C) % 16 == 0;
    // the helper wave copies whole aligned blocks; tail blocks / unaligned rows stay on wave 0
    const bool helper_on = HELPER && (a.obs || a.legal) && vec_ok && nlive == BPW;
    if (HELPER && wave == 0) __builtin_amdgcn_s_setprio(2); // the playing wave is the critical path
    if (HELPER && wave == 1) {
        if (helper_on) {
            for (int t = 0; t < a.T; ++t) { // one emission per step (a.T == 1 outside the rollouts)
                const size_t ob = (size_t)t * (size_t)a.E + (size_t)e0;
                __syncthreads(); // images and view mask of this emission are final
                const u64 view = *sh_view;
                const u32 pk = sh_scal[lane]; // this step's action | done << 8 | reward << 16 of board `lane`
                typename L::Snap sn;
                L::snapshot(lds, sn, lane);
                __syncthreads(); // the snapshot is in registers: wave 0 may play on
                // the step's scalar rows too: a store issued by the playing wave would queue behind these
                // 16-byte stores in the CU's memory pipeline and stall it (measured: 3-5 % of a step)
                // (the two-sided rollout records a position BEFORE its ply: the scalars that arrive with emission t are
                // those of ply t - 1; wave 0 writes the last ply's itself)
                if (MODE != M_ROLLOUT2 || t > 0) {
                    const size_t os = MODE == M_ROLLOUT2 ? ob - (size_t)a.E : ob;
                    if (a.reward) a.reward[os + lane] = (float)(int)(signed char)(pk >> 16);
                    if (a.done) a.done[os + lane] = (unsigned char)((pk >> 8) & 1u);
                    if (a.actions_out) a.actions_out[os + lane] = (int)(pk & 0xffu);
                }
                L::template finish_any<STORE_GAP>(
                    sn, view, a.obs ? a.obs + ob * C : nullptr, a.legal ? a.legal + ob * C : nullptr, lane);
            }
        }
        return;
    }
    // With the helper wave an emission is split: do_emit() publishes the view mask and meets the
    // helper (barrier 1); emit_release() is the second barrier ("snapshot taken") and is called as
    // late as possible -- right before wave 0 next touches the images -- so that wave 0 draws its
    // next action while the helper reads LDS.
    StampBox g_box; // (diagnostic build only)
    u64 agent1 = 0ull; // one-image boards: bit d = board d's image is kept in the WHITE-to-move view
    bool release_owed = false;
    auto emit_release = [&]() {
        if (release_owed) {
            __syncthreads();
            release_owed = false;
        }
    };
    auto do_emit = [&](u64 view, signed char *o, unsigned char *l) {
        if (helper_on) {
            emit_release();
            if (lane == 0) *sh_view = view;
            __syncthreads();
            release_owed = true;
        } else {
            L::emit(lds, view, agent1, o, l, nlive, vec_ok, lane);
        }
    };

    HV_PHASE(blk, 0);
    LS s;
    s.key = mix32(((u32)env + a.env_offset) * 0x9E3779B9U + a.seed_lo) ^ a.seed_hi;
    if (MODE == M_RESET && a.init) {
        for (int i = lane; i < L::LDS_BYTES / 16; i += 64) {
            const bool img = i * 16 >= L::IMG;
            const u32 f = img ? L::EMPTY_BYTE * 0x01010101u : 0u;
            reinterpret_cast<uint4 *>(lds)[i] = make_uint4(f, f, f, f);
        }
        s.ctr = 0;
        s.winner = -1;
        s.agent = !B ? 0 : (a.flags & HEXVEC_F_AGENT_RANDOM) ? (int)(mix32(s.key ^ 0xA5A5A5A5U) & 1u)
                  : (a.flags & HEXVEC_F_AGENT_WHITE) ? 1 : 0;
        s.sr = L::SR_BIAS;
        s.sl = s.se = s.sw = 0u;
#pragma unroll
        for (int k = 0; k < 4; ++k) hblk[k * 64 + lane] = 0u;
        s.pend = 0;
        L::fresh_regs(s, first);
    } else {
#pragma unroll
        for (int j = 0; j < JW; ++j) {
            s.E0[j] = rblk[j * 64 + lane];
            s.E1[j] = rblk[(JW + j) * 64 + lane];
        }
        const u32 *r = rblk + 2 * JW * 64;
        s.ctr = r[0 * 64 + lane];
        s.ep_ret = (int)r[1 * 64 + lane];
        s.ep_len = (int)r[2 * 64 + lane];
        const u32 fl = r[3 * 64 + lane];
        s.cur = fl & 1;
        s.done = (fl >> 2) & 1;
        s.agent = (fl >> 3) & 1;
        s.winner = (int)((fl >> 4) & 7) - 1;
        s.n_empty = (int)(fl >> 8) & 0xff;
        s.pend = (int)(fl >> 16) & 3;
        s.prewon = (int)(fl >> 18) & 3;
        s.sr = r[4 * 64 + lane];
        s.sl = r[5 * 64 + lane];
        s.se = r[6 * 64 + lane];
        s.sw = r[7 * 64 + lane];
        // (the register rows are requested FIRST: loads return in order, so what depends on registers only -- the first
        // action draw -- runs while the block image below is still in flight)
        // The block image goes HBM -> LDS by LDS-DMA (no VGPR round trip): every 1 KB piece is in flight
        // before the first one lands.  (A load / ds_write loop waits for each piece in turn: +2 us a launch.)
        // The tail piece overlaps the one before it, so every instruction runs with all 64 lanes.
        {
            constexpr int V16 = L::LDS_BYTES / 16, PIECES = (V16 + 63) / 64;
            static_assert(V16 >= 64 && L::LDS_BYTES % 16 == 0, "block image smaller than one piece");
#pragma unroll
            for (int k = 0; k < PIECES; ++k) {
                const int at = (k * 64 + 64 <= V16 ? k * 64 : V16 - 64) * 16;
                __builtin_amdgcn_global_load_lds(
                    (const __attribute__((address_space(1))) void *)(sblk + at + lane * 16),
                    (__attribute__((address_space(3))) void *)(lds + at), 16, 0, 0);
            }
        }
    }

    HV_PHASE(blk, 1);
    if (L::ONE) agent1 = __ballot(live && s.agent != 0);
    s.kc = s.key + s.ctr * 0x9E3779B9U;

    // one ply by the player to move; `act` is in the frame `fr` (0: F order, 1: FT order).
    // Returns the reference's make_move code: -1 none, 0/1 winner, 3 illegal.
    // check = false for actions the library drew itself (always legal).
    typename L::NoFill nofill;
    auto ply_f = [&](int act, int fr, bool check, auto &fill) -> int {
        const bool inb = !check || (unsigned)act < (unsigned)C;
        const int yf = inb ? act / N : 0, xf = inb ? act - yf * N : 0;
        const int at = xf * N + yf;
        const int idx0 = fr ? at : (inb ? act : 0), idx1 = fr ? (inb ? act : 0) : at;
        if (check && (!inb || !L::test_bit(s.E0, idx0))) return 3;
        const int p = s.cur;
        const bool same = p == fr; // the plane of colour p is indexed in p's own frame
        const bool win = L::place(lds, s, idx0, idx1, p, same ? yf : xf, same ? xf : yf, lane, g_box, fill) || ((s.prewon >> p) & 1);
        int ret = -1;
        if (win) {
            s.done = 1;
            ret = p;
        } else if (B && s.n_empty <= 0) { // HexSingleGame.py:117-119
            s.done = 1;
        }
        s.cur = p ^ 1;
        return ret;
    };
    auto ply = [&](int act, int fr, bool check) -> int { return ply_f(act, fr, check, nofill); };
    // env B HexEnv.step (HexSingleGame.py:233-263) on top of ply(); returns reward[0]
    auto envb_step_f = [&](int act, bool check, auto &fill) -> int {
        const int mover = s.cur;
        const int w = ply_f(act, mover, check, fill);
        s.winner = w;
        int r = 0;
        if (w == 3) {
            s.done = 1;
            s.cur = mover ^ 1; // the env flips its player and view even then (:259-260)
        } else if (w == mover) {
            r = 1;
        }
        return mover == 0 ? r : -r;
    };
    auto envb_step = [&](int act, bool check) -> int { return envb_step_f(act, check, nofill); };
    // a ply of the in-env opponent: the reply inside step, or the opening after a reset
    // (SelfplayWrapper.py:79-80 / :187-190, HexGame.py:224-230 / :259-265); returns reward[0] (B)
    auto opponent_ply = [&](int oa, bool check) -> int {
        if (B) return envb_step(oa, check);
        s.winner = ply(oa, 1, check); // hex-v0: the opponent's view is the transposed frame
        return 0;
    };
    // The same opponent ply on a board that is known to be EMPTY (just reset inside this kernel):
    // the k-th empty cell is cell k and the stone has no neighbours, so there is nothing to search,
    // chase or union -- one root byte, two image bytes, two bitboard bits.
    auto opening_on_empty = [&]() {
        const u32 r = L::draw(s);
        const int act = (int)__umulhi(r, (u32)C); // n_empty == C
        const int p = s.cur, fr = B ? p : 1;      // B: the mover's frame; hex-v0: the opponent's view
        const int yf = act / N, xf = act - yf * N, at = xf * N + yf;
        const int idx0 = fr ? at : act, idx1 = fr ? act : at;
        L::clear_bit(s.E0, idx0);
        L::clear_bit(s.E1, idx1);
        s.n_empty -= 1;
        L::put_stone(lds, s, idx0, idx1, p, lane);
        const int yp = p == fr ? yf : xf; // row of the stone in p's own frame
        lds[(p * BPW + lane) * L::PS + (p ? idx1 : idx0)] =
            (unsigned char)(0xE0u | (yp == 0 ? 1u : 0u) | (yp == N - 1 ? 2u : 0u));
        s.winner = -1; // make_move returned None
        s.cur = p ^ 1;
    };
    const bool opens = selfplay && (B ? s.agent != 0 : first != 0); // the opponent moves first after a reset
    // wave-level: finished boards -> terminal obs, statistics, fresh board.  `won` per lane.
    auto handle_done = [&](bool dn, int won, signed char *tobs_row) {
        const u64 dmask = __ballot(dn);
        if (!dmask) return;
        if (tobs_row) L::copy_rows(lds, dmask, B ? __ballot(s.cur != 0) : 0ull, agent1, tobs_row, lane);
        L::clear_boards(lds, dmask, lane);
        if (dn) {
            L::add_sums(s, won, hblk, lane);
            L::fresh_regs(s, first);
        }
    };

    const size_t E = (size_t)a.E;

    if (MODE == M_SETBOARD) {
        // HexEnv.reset with a given start position (sample_board, HexSingleGame.py:217-222): the
        // stones are replayed row-major like the constructor does (:57-65); nobody has moved yet.
        const bool sel = live && (!a.env_mask || a.env_mask[env] != 0);
        const u64 smask = __ballot(sel);
        if (smask) {
            L::clear_boards(lds, smask, lane);
            if (sel) {
                s.pend = 0;
                L::fresh_regs(s, first);
                const signed char *bd = a.board_in + (size_t)env * C;
                for (int c = 0; c < C; ++c) {
                    const int v = bd[c];
                    const int p = B ? (v == -1 ? 0 : (v == 1 ? 1 : -1)) : (v == 0 ? 0 : (v == 1 ? 1 : -1));
                    if (p >= 0) {
                        const int y = c / N, x = c - y * N;
                        if (L::place(lds, s, c, x * N + y, p, p ? x : y, p ? y : x, lane, g_box, nofill)) s.prewon |= 1 << p;
                    }
                }
                if (opens) s.pend = 2; // the opponent still owes its opening (split step)
            }
        }
        if (live && a.pending) a.pending[env] = (unsigned char)s.pend;
        do_emit(B ? __ballot(s.cur != 0) : 0ull, a.obs ? a.obs + (size_t)e0 * C : nullptr,
                    a.legal ? a.legal + (size_t)e0 * C : nullptr);
    } else if (MODE == M_RESET) {
        const bool sel = live && (a.init || !a.env_mask || a.env_mask[env] != 0);
        if (!a.init) L::clear_boards(lds, __ballot(sel), lane);
        if (sel) {
            s.pend = 0;
            L::fresh_regs(s, first);
            if (opens) opponent_ply(a.opp_actions ? a.opp_actions[env] : L::rng_action(s, B ? s.cur : 1),
                                    a.opp_actions != nullptr);
        }
        do_emit(B ? __ballot(s.cur != 0) : 0ull, a.obs ? a.obs + (size_t)e0 * C : nullptr,
                    a.legal ? a.legal + (size_t)e0 * C : nullptr);
    } else {
        const bool ext_a = MODE == M_STEP && a.actions, ext_o = MODE == M_STEP && a.opp_actions;
        if (MODE == M_STEP && a.phase != 0) {
            // Split step for opponents that are policies on the device: the same slots as below,
            // one launch per slot, with the opponent's actions supplied by the caller.
            const size_t o = (size_t)env;
            signed char *tobs_row = a.tobs ? a.tobs + (size_t)e0 * C : nullptr;
            int rew = 0, won = 0;
            bool fin = false; // the env step completed in this call
            if (a.phase == 1) { // the agent's ply (slot 1)
                if (live && s.pend == 0) {
                    const int act = ext_a ? a.actions[o] : L::rng_action(s, B ? s.cur : 0);
                    if (B) {
                        const int rb = envb_step(act, ext_a);
                        rew = s.agent == 0 ? rb : -rb;
                    } else {
                        s.winner = ply(act, 0, ext_a);
                        if (s.winner == 3) s.done = 1;
                        rew = s.winner == 0 ? 1 : (s.winner == 3 ? -100 : 0);
                    }
                    fin = s.done != 0;
                    s.pend = fin ? 0 : 1;
                    if (!fin && a.done) a.done[o] = 0; // in progress: the caller need not clear the row
                }
            } else { // the opponent's plies: replies (slot 2) and openings (slot 3)
                const int was = s.pend;
                if (live && was != 0) {
                    const int oa = ext_o ? a.opp_actions[o] : L::rng_action(s, B ? s.cur : 1);
                    const int rb = opponent_ply(oa, ext_o);
                    s.pend = 0;
                    if (was == 1) {
                        rew = B ? (s.agent == 0 ? rb : -rb)
                                : (s.winner == 0 ? 1 : (s.winner == 1 ? -1 : (s.winner == 3 ? -100 : 0)));
                        fin = true;
                    }
                }
            }
            if (fin) {
                s.ep_ret += rew;
                s.ep_len += 1;
                won = s.done && rew > 0;
                if (a.reward) a.reward[o] = (float)rew;
                if (a.done) a.done[o] = (unsigned char)s.done;
                if (a.winner) a.winner[o] = (signed char)s.winner;
            }
            const bool dn = fin && s.done;
            handle_done(dn, won, tobs_row);
            if (dn && opens) s.pend = 2; // the opponent opens on the fresh board
            if (live && a.pending) a.pending[o] = (unsigned char)s.pend;
            do_emit(B ? __ballot(s.cur != 0) : 0ull, a.obs ? a.obs + (size_t)e0 * C : nullptr,
                    a.legal ? a.legal + (size_t)e0 * C : nullptr);
        } else {
        HV_STAMP_INIT();
        for (int t = 0; t < a.T; ++t) {
            const size_t ob = (size_t)t * E + (size_t)e0; // first board of this block in row t
            const size_t o = ob + (size_t)lane;
            signed char *tobs_row = a.tobs ? a.tobs + ob * C : nullptr;
            if (MODE == M_ROLLOUT2)
                do_emit(B ? __ballot(s.cur != 0) : 0ull, a.obs ? a.obs + ob * C : nullptr,
                    a.legal ? a.legal + ob * C : nullptr);
            // ---- slot 1: the agent's ply (two-sided: the mover's)
            int rew = 0, won = 0, f_done = 0, f_winner = -1;
            bool reply = false; // this lane still owes the opponent's reply
            // (SPEC: the reply is drawn inside the agent's ply, from the frame the opponent's actions are counted in)
            constexpr bool spec = SPEC != 0 && MODE == M_ROLLOUT1;
            typename L::NextDraw reply_draw;
            reply_draw.f = B ? (s.agent ^ 1) : 1;
            reply_draw.act = 0;
            // B: the mover's frame.  Drawn before the helper's release barrier: registers only.
            const int act = ext_a ? (live ? a.actions[o] : 0) : L::rng_action(s, B ? s.cur : 0);
            HV_MARK(0); // agent action (rng + k-th empty)
            emit_release();
            HV_MARK(1); // wait for the helper's snapshot
            if (live) {
                if (two) {
                    const int mover = s.cur;
                    if (B) { // HexEnv.step (HexSingleGame.py:233-263)
                        const int rb = envb_step(act, ext_a);
                        rew = mover == 0 ? rb : -rb;
                        s.ep_ret += rb;
                        won = s.done && s.winner == 0;
                    } else { // HexGame.make_move with hex-v0's convention for an illegal ply
                        const int w = ply(act, 0, ext_a);
                        s.winner = w;
                        if (w == 3) s.done = 1;
                        rew = w == mover ? 1 : (w == 3 ? -100 : 0);
                        s.ep_ret += w == 0 ? 1 : (w == 1 ? -1 : 0);
                        won = s.done && w == 0;
                    }
                } else if (B) { // SelfPlayEnv.step (SelfplayWrapper.py:174-199), first half
                    const int rb = spec ? envb_step_f(act, ext_a, reply_draw) : envb_step(act, ext_a);
                    rew = s.agent == 0 ? rb : -rb;
                    reply = !s.done;
                } else { // hex-v0 HexEnv.step (HexGame.py:244-295), agent BLACK, first half
                    s.winner = spec ? ply_f(act, 0, ext_a, reply_draw) : ply(act, 0, ext_a);
                    if (s.winner == 3) s.done = 1;
                    rew = s.winner == 0 ? 1 : (s.winner == 3 ? -100 : 0);
                    reply = !s.done;
                }
                if (!reply) { // the step's result is final for this lane
                    if (!two) {
                        s.ep_ret += rew;
                        won = s.done && rew > 0;
                    }
                    s.ep_len += 1;
                    f_done = s.done;
                    f_winner = s.winner;
                }
            }
            const bool dn1 = live && !reply && s.done;
            HV_MARK(2); // agent ply
            if (!selfplay) {
                handle_done(dn1, won, tobs_row);
            } else {
                // ---- slot 2: the opponent's reply
                if (reply) {
                    int oa;
                    if (spec) { // drawn inside the agent's ply; consumed now that the reply is really played
                        L::commit_draw(s);
                        oa = reply_draw.act;
                    } else {
                        oa = ext_o ? a.opp_actions[o] : L::rng_action(s, B ? s.cur : 1);
                    }
                    const int rb = opponent_ply(oa, ext_o);
                    rew = B ? (s.agent == 0 ? rb : -rb)
                            : (s.winner == 0 ? 1 : (s.winner == 1 ? -1 : (s.winner == 3 ? -100 : 0)));
                    s.ep_ret += rew;
                    s.ep_len += 1;
                    won = s.done && rew > 0;
                    f_done = s.done;
                    f_winner = s.winner;
                }
                HV_MARK(4); // opponent action + ply
                // ---- the boards that ended on either ply: terminal obs, statistics, fresh board (the
                // ones that ended on the agent's ply were not touched since), then
                // ---- slot 3: the opponent's opening where it moves first; always drawn by the library
                const bool dn = dn1 || (reply && s.done);
                handle_done(dn, won, tobs_row);
                if (dn && opens) opening_on_empty();
            }
            if (helper_on && (MODE == M_ROLLOUT1 || (MODE == M_ROLLOUT2 && t + 1 < a.T))) {
                // the helper stores them with the observation rows (two-sided: with the NEXT emission)
                sh_scal[lane] = ((u32)act & 0xffu) | ((u32)f_done << 8) | (((u32)rew & 0xffu) << 16);
            } else if (live) {
                if (a.reward) a.reward[o] = (float)rew;
                if (a.done) a.done[o] = (unsigned char)f_done;
                if (a.winner) a.winner[o] = (signed char)f_winner;
                if (a.actions_out) a.actions_out[o] = act;
            }
            HV_MARK(5); // finished boards 2, slot 3, scalar outputs
            if (MODE != M_ROLLOUT2)
                do_emit(B ? __ballot(s.cur != 0) : 0ull, a.obs ? a.obs + ob * C : nullptr,
                    a.legal ? a.legal + ob * C : nullptr);
            HV_MARK(6); // publish the emission (barrier 1)
        }
        HV_STAMP_FLUSH(blk);
        }
    }

    HV_PHASE(blk, 2);
    // state back to HBM (before the emission's last barrier: from here on both wavefronts only READ the LDS)
    {   // eight 1 KB pieces per trip: the LDS reads of a trip are issued together, then its stores
        constexpr int V16 = L::LDS_BYTES / 16, PIECES = (V16 + 63) / 64;
#pragma unroll
        for (int k0 = 0; k0 < PIECES; k0 += 8) {
            uint4 v[8];
#pragma unroll
            for (int k = 0; k < 8; ++k)
                if (k0 + k < PIECES) {
                    const int at = ((k0 + k) * 64 + 64 <= V16 ? (k0 + k) * 64 : V16 - 64) + lane;
                    v[k] = reinterpret_cast<const uint4 *>(lds)[at];
                }
#pragma unroll
            for (int k = 0; k < 8; ++k)
                if (k0 + k < PIECES) {
                    const int at = ((k0 + k) * 64 + 64 <= V16 ? (k0 + k) * 64 : V16 - 64) + lane;
                    reinterpret_cast<uint4 *>(sblk)[at] = v[k];
                }
        }
    }
#pragma unroll
    for (int j = 0; j < JW; ++j) {
        rblk[j * 64 + lane] = s.E0[j];
        rblk[(JW + j) * 64 + lane] = s.E1[j];
    }
    u32 *r = rblk + 2 * JW * 64;
    r[0 * 64 + lane] = s.ctr;
    r[1 * 64 + lane] = (u32)s.ep_ret;
    r[2 * 64 + lane] = (u32)s.ep_len;
    r[3 * 64 + lane] = (u32)s.cur | ((u32)s.done << 2) | ((u32)s.agent << 3) | ((u32)(s.winner + 1) << 4) |
                       ((u32)(s.n_empty & 0xff) << 8) | ((u32)s.pend << 16) | ((u32)s.prewon << 18);
    r[4 * 64 + lane] = s.sr;
    r[5 * 64 + lane] = s.sl;
    r[6 * 64 + lane] = s.se;
    r[7 * 64 + lane] = s.sw;
    emit_release();
    HV_PHASE(blk, 3);
}

// ---- cold kernels on the fast state ----
template <int N, int ENC, int BPW, int VIEWS>
__global__ void k_lpb_get_state(const unsigned char *fast, int E, signed char *board, unsigned char *cur)
{
    using L = Lpb<N, ENC, BPW, VIEWS>;
    constexpr int C = Geo<N, BPW>::C;
    const long long gid = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= (long long)E * (C + 1)) return;
    const int env = (int)(gid / (C + 1)), r = (int)(gid - (long long)env * (C + 1));
    const int blk = env / BPW, lane = env % BPW;
    const unsigned char *sblk = fast + (size_t)blk * FastState<N, ENC, BPW, VIEWS>::BLOCK_BYTES;
    const u32 *regs = reinterpret_cast<const u32 *>(sblk + L::LDS_BYTES) + 2 * Geo<N, BPW>::JW * 64;
    const int c = (int)(regs[3 * 64 + lane] & 1u);
    if (r < C) {
        if (board) {
            if (L::ONE) { // one image, in the agent's view; the stored frame is the mover's
                const bool flip = c != (int)((regs[3 * 64 + lane] >> 3) & 1u);
                u32 x = sblk[L::IMG + lane * C + (flip ? L::transpose(r) : r)];
                board[(size_t)env * C + r] = (signed char)((flip && x) ? x ^ 0xFEu : x);
            } else {
                const int v = (L::TWOV && c) ? 1 : 0;
                board[(size_t)env * C + r] = (signed char)sblk[L::IMG + v * Geo<N, BPW>::IMG_BYTES + lane * C + r];
            }
        }
    } else if (cur) {
        cur[env] = (unsigned char)c;
    }
}

template <int N, int ENC, int BPW, int VIEWS>
__global__ void k_lpb_stats(const unsigned char *fast, int E, double *out4, unsigned char *agents)
{
    using L = Lpb<N, ENC, BPW, VIEWS>;
    __shared__ double sh[4][256];
    double acc[4] = {0, 0, 0, 0};
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < E; i += gridDim.x * blockDim.x) {
        const unsigned char *sblk = fast + (size_t)(i / BPW) * FastState<N, ENC, BPW, VIEWS>::BLOCK_BYTES;
        const u32 *regs = reinterpret_cast<const u32 *>(sblk + L::LDS_BYTES) + 2 * Geo<N, BPW>::JW * 64;
        const u32 *hi = reinterpret_cast<const u32 *>(sblk + FastState<N, ENC, BPW, VIEWS>::HI_OFF);
        for (int k = 0; k < 4; ++k) { // 64-bit sums (exact in a double below 2^53); the return sum is kept biased
            const long long v = (long long)(((u64)hi[k * 64 + (i % BPW)] << 32) | (u64)regs[(4 + k) * 64 + (i % BPW)]);
            acc[k] += (double)(k == 0 ? v - (long long)L::SR_BIAS : v);
        }
        if (agents) agents[i] = (unsigned char)((regs[3 * 64 + (i % BPW)] >> 3) & 1u);
    }
    if (!out4) return;
    for (int k = 0; k < 4; ++k) sh[k][threadIdx.x] = acc[k];
    __syncthreads();
    for (int st = 128; st > 0; st >>= 1) {
        if ((int)threadIdx.x < st)
            for (int k = 0; k < 4; ++k) sh[k][threadIdx.x] += sh[k][threadIdx.x + st];
        __syncthreads();
    }
    // out4 was zeroed on the same stream; sums of integers are exact in any order
    if (threadIdx.x < 4) atomicAdd(&out4[threadIdx.x], sh[threadIdx.x][0]);
}

} // namespace lpb

// ---- launchers of this translation unit ----
#define HV_LPB_SIZES(X) X(4) X(5) X(6) X(7) X(8) X(9) X(10) X(11) X(12) X(13) X(14)

// enc B: two-sided handles keep both views, self-play handles one image per board (hexvec_lpb.h header)
template <int N, int ENC, int MODE, int VIEWS> static hipError_t lpb_launch2(const KArgs &a, hipStream_t st)
{
    const int blocks = (a.E + HV_BPW - 1) / HV_BPW;
    // the helper wave serves the two rollouts (the other modes emit once per launch and do not care): it takes the
    // stores off the playing wavefront and, where the store path is the bound, paces them (STORE_GAP)
    constexpr int HELPER = (HV_HELPER && (MODE == M_ROLLOUT1 || MODE == M_ROLLOUT2)) ? 1 : 0;
    if constexpr (MODE == M_ROLLOUT1) {
        // The self-play rollout exists in two forms with identical results.  Drawing the reply inside the agent's
        // ply shortens the playing wavefront's step by 7 % (2.61 -> 2.43 us at 11x11 without outputs); that is the
        // whole gain where the launch is bound by that wavefront: fewer playing wavefronts than SIMDs (7x7 / 32 768:
        // -7 %, 13x13 / 16 384: -4 %) and short launches (11x11 / 65 536 at 20..60 steps per launch: -3 %).  Long
        // launches of a full chip run against the store path instead (2.6 us per step = 6.3 TB/s), and there the plain
        // loop is the steadier one: over six placements of the output buffers 2.62-2.77 us per step against 2.52-2.98
        // (tools/ab_inproc.py, same process, same buffers).  From 12x12 on a full chip is store-bound at
        // any launch length (3.5-4.5 us per step against a playing wavefront of 2.7): always the plain form, whose helper
        // wavefront spaces its stores there (k_lpb, STORE_GAP).
        static const int force = [] { // A/B experiments: HEXVEC_LPB_SPEC=0 / 1 forces one form
            const char *ev = getenv("HEXVEC_LPB_SPEC");
            return ev ? atoi(ev) : -1;
        }();
        const bool spec = force >= 0 ? force != 0 : (blocks < 1024 || (a.T <= 64 && N <= 11));
        if (spec) hipLaunchKernelGGL((lpb::k_lpb<N, ENC, MODE, HV_BPW, HELPER, VIEWS, 1>), dim3(blocks), dim3(64 * (1 + HELPER)), 0, st, a);
        else hipLaunchKernelGGL((lpb::k_lpb<N, ENC, MODE, HV_BPW, HELPER, VIEWS, 0>), dim3(blocks), dim3(64 * (1 + HELPER)), 0, st, a);
        return hipGetLastError();
    } else if constexpr (MODE == M_ROLLOUT2) {
        // the two-sided rollout: a helper wavefront stores the rows as in the self-play one; on a full chip (every ply is
        // an emission: store-bound from 9x9 on) it spaces its stores (k_lpb, STORE_GAP), below that it does not
        if (blocks >= 1024 && N >= 9) hipLaunchKernelGGL((lpb::k_lpb<N, ENC, MODE, HV_BPW, HELPER, VIEWS, 1>), dim3(blocks), dim3(64 * (1 + HELPER)), 0, st, a);
        else hipLaunchKernelGGL((lpb::k_lpb<N, ENC, MODE, HV_BPW, HELPER, VIEWS, 0>), dim3(blocks), dim3(64 * (1 + HELPER)), 0, st, a);
        return hipGetLastError();
    } else {
        hipLaunchKernelGGL((lpb::k_lpb<N, ENC, MODE, HV_BPW, HELPER, VIEWS>), dim3(blocks), dim3(64 * (1 + HELPER)), 0, st, a);
        return hipGetLastError();
    }
}
template <int N, int ENC, int MODE> static hipError_t lpb_launch1(const KArgs &a, hipStream_t st)
{
    if (ENC == 0) return lpb_launch2<N, ENC, MODE, 1>(a, st);
    const bool two = (a.flags & HEXVEC_F_TWO_SIDED) != 0;
    // (the self-play rollout exists on self-play handles only, the two-sided one on two-sided handles)
    if (MODE == M_ROLLOUT1) return two ? hipErrorInvalidValue : lpb_launch2<N, ENC, MODE, 1>(a, st);
    if (MODE == M_ROLLOUT2) return two ? lpb_launch2<N, ENC, MODE, 2>(a, st) : hipErrorInvalidValue;
    return two ? lpb_launch2<N, ENC, MODE, 2>(a, st) : lpb_launch2<N, ENC, MODE, 1>(a, st);
}
template <int N, int ENC> static hipError_t lpb_launch_mode(int mode, const KArgs &a, hipStream_t st)
{
    switch (mode) {
    case M_ROLLOUT2: return lpb_launch1<N, ENC, M_ROLLOUT2>(a, st);
    case M_STEP: return lpb_launch1<N, ENC, M_STEP>(a, st);
    case M_ROLLOUT1: return lpb_launch1<N, ENC, M_ROLLOUT1>(a, st);
    case M_RESET: return lpb_launch1<N, ENC, M_RESET>(a, st);
    case M_SETBOARD: return lpb_launch1<N, ENC, M_SETBOARD>(a, st);
    default: return hipErrorInvalidValue;
    }
}
hipError_t HV_LPB_NAME(launch)(int N, int enc, int mode, const KArgs &a, hipStream_t st)
{
    switch (N) {
#define HV_CASE(n) case n: return enc ? lpb_launch_mode<n, 1>(mode, a, st) : lpb_launch_mode<n, 0>(mode, a, st);
        HV_LPB_SIZES(HV_CASE)
#undef HV_CASE
    default: return hipErrorInvalidValue;
    }
}
size_t HV_LPB_NAME(block_bytes)(int N, int enc, int two)
{
    switch (N) {
#define HV_CASE(n)                                                                              \
    case n:                                                                                     \
        return !enc ? lpb::FastState<n, 0, HV_BPW, 1>::BLOCK_BYTES                              \
                    : two ? lpb::FastState<n, 1, HV_BPW, 2>::BLOCK_BYTES : lpb::FastState<n, 1, HV_BPW, 1>::BLOCK_BYTES;
        HV_LPB_SIZES(HV_CASE)
#undef HV_CASE
    default: return 0;
    }
}
hipError_t HV_LPB_NAME(get_state)(const unsigned char *fast, int N, int enc, int two, int E, signed char *board,
                                  unsigned char *cur, hipStream_t st)
{
    const long long total = (long long)E * (N * N + 1);
    const dim3 g((unsigned)((total + 255) / 256)), b(256);
    switch (N) {
#define HV_CASE(n)                                                                                                  \
    case n:                                                                                                         \
        if (!enc) hipLaunchKernelGGL((lpb::k_lpb_get_state<n, 0, HV_BPW, 1>), g, b, 0, st, fast, E, board, cur);     \
        else if (two) hipLaunchKernelGGL((lpb::k_lpb_get_state<n, 1, HV_BPW, 2>), g, b, 0, st, fast, E, board, cur); \
        else hipLaunchKernelGGL((lpb::k_lpb_get_state<n, 1, HV_BPW, 1>), g, b, 0, st, fast, E, board, cur);          \
        break;
        HV_LPB_SIZES(HV_CASE)
#undef HV_CASE
    default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}
// out4 must have been zeroed on the same stream (the blocks add their partial sums)
hipError_t HV_LPB_NAME(stats)(const unsigned char *fast, int N, int enc, int two, int E, double *out4,
                              unsigned char *agents, hipStream_t st)
{
    const dim3 grid((E + 255) / 256 < 1024 ? (E + 255) / 256 : 1024), b(256);
    switch (N) {
#define HV_CASE(n)                                                                                                     \
    case n:                                                                                                            \
        if (!enc) hipLaunchKernelGGL((lpb::k_lpb_stats<n, 0, HV_BPW, 1>), grid, b, 0, st, fast, E, out4, agents);       \
        else if (two) hipLaunchKernelGGL((lpb::k_lpb_stats<n, 1, HV_BPW, 2>), grid, b, 0, st, fast, E, out4, agents);   \
        else hipLaunchKernelGGL((lpb::k_lpb_stats<n, 1, HV_BPW, 1>), grid, b, 0, st, fast, E, out4, agents);            \
        break;
        HV_LPB_SIZES(HV_CASE)
#undef HV_CASE
    default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

#ifdef HV_STAMPS
extern "C" int hexvec_debug_stamps(unsigned long long *host_dst, int nblocks)
{
    return (int)hipMemcpyFromSymbol(host_dst, HIP_SYMBOL(g_stamps), (size_t)nblocks * 12 * sizeof(unsigned long long));
}
extern "C" int hexvec_debug_phases(unsigned long long *host_dst, int nblocks)
{
    return (int)hipMemcpyFromSymbol(host_dst, HIP_SYMBOL(g_phase), (size_t)nblocks * 4 * sizeof(unsigned long long));
}
#endif

} // namespace hv
