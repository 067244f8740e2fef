// hexvec_qpb.h -- lanes-per-board kernels with the whole game state in registers as bitboards.
//
// L = 1 / 2 / 4 / 8 lanes own one board (4 at 9x9..11x11: a quad per board, 16 boards per wavefront);
// lane l of the group holds 32-bit word l of every bitboard of its board:
//
//   K0 K1 W0 W1   stones of BLACK / WHITE in frame F (cell y*N+x) and in the transposed frame F^T
//                 (cell x*N+y).  Encoding B observes the board in the mover's perspective
//                 (HexSingleGame.py:265-271): BLACK to move sees F, WHITE to move sees -F^T, so both
//                 frames are kept and nothing is ever transposed; empties are ~(K|W).
//   TK DK TW DW   the stones of BLACK (frame F) / WHITE (frame F^T) that are connected to row 0 (T) or
//                 to row N-1 (D) of that colour's OWN frame (both colours connect row 0 with row N-1 of
//                 their own frame and share the six hex offsets).
//
// A ply of colour p at cell a (HexGame.py:85-111, HexSingleGame.py:88-122):  nb = the six neighbours of
// a (one funnel shift + destination mask per direction; two DPP row shifts fetch the neighbouring
// words of the board); the stone touches the top iff a is in row 0 or nb meets T, the bottom likewise; the
// mover has connected its edges iff both hold -- that is the reference's test regions[p][-1,-1] == 1
// after flood_fill (HexGame.py:98-105; plain connectivity).  A stone that touches an edge set
// drags its floating component into that set: a frontier flood over S & ~(T|D), one dilation per level
// (0.29 levels per ply on average at 11x11; a wavefront waits for the deepest of its 16 boards, ~3).
// No LDS, no scratch: the state of a board is 8 bitboards + 3 words (144 B at 11x11), which is what a
// host-stepped launch moves in and out of HBM, and any number of boards of any size runs in one pass.
//
// Observations and legal masks leave the registers directly: a lane expands the 32 cells of its own
// word to 2 x 16 bytes per array and stores them at board*C + 32*l (byte-aligned 16-byte stores; the
// last word of a board stores the window [C-32, C), overlapping its neighbour with equal bytes).
//
// Scope: auto-reset handles (as hexvec_lpb.h): no ply is ever applied after the end of a game.
#pragma once
#include "hexvec_common.h"

// Diagnostic build (-DHV_QSTAMPS, tools/stamp_step.py): s_memrealtime (100 MHz, one clock for the whole chip) per
// wavefront at kernel entry / state loaded / step computed (before the state store) / stores retired.  No stamp
// exists in the product build.
#ifdef HV_QSTAMPS
__device__ unsigned long long g_qphase[8192][4];
#define HV_QPHASE(blk, k)                                                                    \
    do {                                                                                     \
        __builtin_amdgcn_s_waitcnt(0);                                                       \
        if (lane == 0 && (blk) < 8192) g_qphase[blk][k] = __builtin_amdgcn_s_memrealtime();  \
    } while (0)
#else
#define HV_QPHASE(blk, k) do { } while (0)
#endif

namespace hv {
namespace qpb {

template <int N> struct Geo {
    static constexpr int C = N * N;
    static constexpr int JW = (C + 31) / 32;                                          // words per bitboard
    static constexpr int L = JW <= 1 ? 1 : JW <= 2 ? 2 : JW <= 4 ? 4 : JW <= 8 ? 8 : 16; // lanes per board
    static constexpr int BPW = 64 / L;                                                // boards per wavefront
    static constexpr int NSC = 3;                   // persisted scalars per board: rng counter, flags, episode
    static constexpr int SROWS = (NSC + L - 1) / L; // 64-word rows that hold them (slot k of a board: lane k % L, row k / L)
    static constexpr int ROWS = 8 + SROWS;
    static constexpr size_t BLOCK_BYTES = (size_t)ROWS * 64 * 4; // HBM state of one wavefront
    static_assert(N >= 2 && JW <= 16, "board too large for one 16-lane DPP row");
};

enum MaskKind { MK_VALID, MK_NOT_COL0, MK_NOT_COLL, MK_NOT_ROW0, MK_NOT_ROWL, MK_DN_LEFT, MK_UP_RIGHT };

template <int N> constexpr u32 mask_word(int kind, int w)
{
    u32 m = 0;
    for (int b = 0; b < 32; ++b) {
        const int c = w * 32 + b;
        if (c >= N * N) break;
        const int y = c / N, x = c % N;
        bool in = true;
        if (kind == MK_NOT_COL0) in = x > 0;
        else if (kind == MK_NOT_COLL) in = x < N - 1;
        else if (kind == MK_NOT_ROW0) in = y > 0;
        else if (kind == MK_NOT_ROWL) in = y < N - 1;
        else if (kind == MK_DN_LEFT) in = y < N - 1 && x > 0;
        else if (kind == MK_UP_RIGHT) in = y > 0 && x < N - 1;
        if (in) m |= 1u << b;
    }
    return m;
}
// word l of a compile-time cell mask, selected by the lane's position in its group
template <int N, int KIND, int W = 0> __device__ __forceinline__ u32 lane_mask(int l)
{
    if constexpr (W >= Geo<N>::L) {
        return 0u;
    } else {
        constexpr u32 v = mask_word<N>(KIND, W);
        const u32 rest = lane_mask<N, KIND, W + 1>(l);
        return l == W ? v : rest;
    }
}

// DPP move; lanes whose source is outside the 16-lane row (or disabled) read 0
template <int CTRL> __device__ __forceinline__ u32 dpp0(u32 x)
{
    return (u32)__builtin_amdgcn_update_dpp(0, (int)x, CTRL, 0xF, 0xF, true);
}
constexpr int DPP_QUAD_XOR1 = 0xB1, DPP_QUAD_XOR2 = 0x4E, DPP_HALF_MIRROR = 0x141, DPP_ROW_MIRROR = 0x140;
constexpr int DPP_ROW_SHL1 = 0x101; // lane i reads lane i+1
constexpr int DPP_ROW_SHR1 = 0x111; // lane i reads lane i-1 (SHR n = 0x110 + n)

template <int N, int ENC> struct Q {
    using G = Geo<N>;
    static constexpr bool B = ENC == 1;
    static constexpr int C = G::C, JW = G::JW, L = G::L;

    struct LC { // per-lane constants
        int l;
        u32 valid, m_r, m_l, m_dn, m_up, m_dl, m_ur;
        u32 ge1, ge2, ge4, ge8;
    };
    struct LS {
        u32 K0, K1, W0, W1, TK, DK, TW, DW;
        int cur; // player to move (B: also the frame of the stored board)
        int done, winner, n_empty, agent;
        int pend;   // split step: 0 agent to move, 1 the opponent owes its reply, 2 its opening
        int prewon; // bit p: colour p was already connected in the start position (set_boards)
        u32 ctr, key;
        u32 kc; // key + ctr * 0x9E3779B9, kept in step with ctr (no multiply per draw)
        int ep_ret, ep_len;
        int sr, sl, se, sw; // episode sums of THIS launch (added to the stats rows at the end)
    };

    static __device__ __forceinline__ LC lane_consts(int lane)
    {
        LC c;
        c.l = lane & (L - 1);
        c.valid = lane_mask<N, MK_VALID>(c.l);
        c.m_r = lane_mask<N, MK_NOT_COLL>(c.l);
        c.m_l = lane_mask<N, MK_NOT_COL0>(c.l);
        c.m_dn = lane_mask<N, MK_NOT_ROWL>(c.l);
        c.m_up = lane_mask<N, MK_NOT_ROW0>(c.l);
        c.m_dl = lane_mask<N, MK_DN_LEFT>(c.l);
        c.m_ur = lane_mask<N, MK_UP_RIGHT>(c.l);
        c.ge1 = c.l >= 1 ? ~0u : 0u;
        c.ge2 = c.l >= 2 ? ~0u : 0u;
        c.ge4 = c.l >= 4 ? ~0u : 0u;
        c.ge8 = c.l >= 8 ? ~0u : 0u;
        return c;
    }

    // OR over the lanes of a group, result in every lane
    static __device__ __forceinline__ u32 group_or(u32 x)
    {
        if (L >= 2) x |= dpp0<DPP_QUAD_XOR1>(x);
        if (L >= 4) x |= dpp0<DPP_QUAD_XOR2>(x);
        if (L >= 8) x |= dpp0<DPP_HALF_MIRROR>(x);
        if (L >= 16) x |= dpp0<DPP_ROW_MIRROR>(x);
        return x;
    }
    // the six hex neighbours (-1,0) (-1,+1) (0,-1) (0,+1) (+1,-1) (+1,0) of every cell of m
    // (HexGame.py:128-131: the 3x3 neighbourhood without [0,0] and [2,2]).
    // A shift by S cells is one funnel shift (v_alignbit) over the lane's word and its neighbour's
    // (ONE DPP row shift per direction of travel serves all three shifts that way); what a shift
    // drags across a row end, a board end or in from the neighbouring board lands only on cells the
    // DESTINATION mask of that direction clears (m_l: column > 0, m_up: row > 0, ...; for a shift
    // towards higher cells the stray bits fall into row 0 or column 0, towards lower cells into
    // row N-1, column N-1 or the padding), so the source needs no masking.
    static __device__ __forceinline__ u32 dilate(u32 m, const LC &c)
    {
        if (L == 1) {
            return ((m << 1) & c.m_l) | ((m >> 1) & c.m_r) | ((m << N) & c.m_up) | ((m >> N) & c.m_dn) |
                   ((m << (N - 1)) & c.m_ur) | ((m >> (N - 1)) & c.m_dl);
        }
        const u32 prev = dpp0<DPP_ROW_SHR1>(m), next = dpp0<DPP_ROW_SHL1>(m); // word l-1 / l+1 of the board
        u32 r = __builtin_amdgcn_alignbit(m, prev, 31) & c.m_l;                // +1
        r |= __builtin_amdgcn_alignbit(next, m, 1) & c.m_r;                    // -1
        r |= __builtin_amdgcn_alignbit(m, prev, 32 - N) & c.m_up;              // +N
        r |= __builtin_amdgcn_alignbit(next, m, N) & c.m_dn;                   // -N
        r |= __builtin_amdgcn_alignbit(m, prev, 32 - (N - 1)) & c.m_ur;        // +(N-1): (y+1, x-1)
        r |= __builtin_amdgcn_alignbit(next, m, N - 1) & c.m_dl;               // -(N-1): (y-1, x+1)
        return r;
    }

    // position (0..31) of the k-th set bit of word (k < popcount)
    static __device__ __forceinline__ int kth_in_word(u32 word, u32 k)
    {
        int pos = 0;
#pragma unroll
        for (int sh = 16; sh >= 1; sh >>= 1) {
            const u32 cl = (u32)__popc(word & ((1u << sh) - 1u));
            const bool up = k >= cl;
            k = up ? k - cl : k;
            word = up ? word >> sh : word;
            pos += up ? sh : 0;
        }
        return pos;
    }
    // cell index of the k-th set bit of the word-sliced set E (k < its popcount); same value in
    // every lane of the group
    static __device__ __forceinline__ int kth_set(u32 Ew, u32 k, const LC &c)
    {
        const u32 pc = (u32)__popc(Ew);
        u32 inc = pc; // inclusive prefix sum over the group's lanes
        if (L >= 2) inc += dpp0<0x111>(inc) & c.ge1;
        if (L >= 4) inc += dpp0<0x112>(inc) & c.ge2;
        if (L >= 8) inc += dpp0<0x114>(inc) & c.ge4;
        if (L >= 16) inc += dpp0<0x118>(inc) & c.ge8;
        const u32 kk = k - (inc - pc);
        const bool here = kk < pc; // unsigned: also false when k is below this lane's range
        const int pos = kth_in_word(Ew, here ? kk : 0u);
        return (int)group_or(here ? (u32)(c.l * 32 + pos) : 0u);
    }

    static __device__ __forceinline__ u32 draw(LS &s)
    {
        const u32 r = mix32(s.kc);
        s.kc += 0x9E3779B9U;
        s.ctr += 1;
        return r;
    }
    static __device__ __forceinline__ u32 empties(const LS &s, int f, const LC &c)
    {
        return ~(f ? (s.K1 | s.W1) : (s.K0 | s.W0)) & c.valid;
    }
    // uniform-legal action in frame f's row-major order (SelfplayWrapper.py:16-22 shape)
    static __device__ __forceinline__ int rng_action(LS &s, int f, const LC &c)
    {
        const u32 r = draw(s);
        if (s.n_empty <= 0) return 0;
        return kth_set(empties(s, f, c), __umulhi(r, (u32)s.n_empty), c);
    }
    static __device__ __forceinline__ u32 cell_bit(int idx, const LC &c)
    {
        return (idx >> 5) == c.l ? 1u << (idx & 31) : 0u;
    }

    static __device__ __forceinline__ void fresh_regs(LS &s, int first)
    {
        s.K0 = s.K1 = s.W0 = s.W1 = 0u;
        s.TK = s.DK = s.TW = s.DW = 0u;
        s.cur = first;
        s.done = 0;
        s.n_empty = C;
        s.ep_ret = 0;
        s.ep_len = 0;
        s.prewon = 0;
    }

    // One legal stone of colour p at idx0 (F order) == idx1 (F^T order), row yo of p's own frame.
    // Returns true iff p now connects its two edges (HexGame.py:98-105 via connectivity).
    static __device__ __forceinline__ bool place(LS &s, int idx0, int idx1, int p, int yo, const LC &c)
    {
        const u32 b0 = cell_bit(idx0, c), b1 = cell_bit(idx1, c);
        s.K0 |= p ? 0u : b0;
        s.K1 |= p ? 0u : b1;
        s.W0 |= p ? b0 : 0u;
        s.W1 |= p ? b1 : 0u;
        s.n_empty -= 1;
        const u32 ab = p ? b1 : b0; // the stone in p's own frame
        const u32 S = p ? s.W1 : s.K0;
        u32 T = p ? s.TW : s.TK, D = p ? s.DW : s.DK;
        const u32 nb = dilate(ab, c);
        const u32 f = group_or(((nb & T) ? 1u : 0u) | ((nb & D) ? 2u : 0u)) | (yo == 0 ? 1u : 0u) |
                      (yo == N - 1 ? 2u : 0u);
        if (f != 0u) { // the stone and the floating stones connected to it join the edge set(s)
            const u32 Fl = S & ~(T | D | ab);
            u32 comp = ab, fr = nb & Fl;
            while (__ballot(fr != 0u)) {
                comp |= fr;
                fr = dilate(fr, c) & Fl & ~comp;
            }
            T |= (f & 1u) ? comp : 0u;
            D |= (f & 2u) ? comp : 0u;
            // (selects of values, not of members: the state must stay in registers)
            s.TK = p ? s.TK : T;
            s.DK = p ? s.DK : D;
            s.TW = p ? T : s.TW;
            s.DW = p ? D : s.DW;
        }
        return f == 3u;
    }

    // 4 cells (bits 4k..4k+3 of x) -> one byte each
    static __device__ __forceinline__ u32 spread1(u32 x, int k) // 0x01 per set bit
    {
        return (((x >> (4 * k)) & 0xFu) * 0x00204081u) & 0x01010101u;
    }
    static __device__ __forceinline__ u32 spread2(u32 x, int k) // 0x02 per set bit
    {
        return (((x >> (4 * k)) & 0xFu) * 0x00408102u) & 0x02020202u;
    }
    typedef u32 v4u __attribute__((ext_vector_type(4)));
    struct __attribute__((packed, aligned(1))) V16 {
        u32 x, y, z, w;
    };
    static __device__ __forceinline__ void store16(void *p, u32 x, u32 y, u32 z, u32 w)
    {
        V16 v = {x, y, z, w};
        *reinterpret_cast<V16 *>(p) = v; // byte-aligned 16-byte store (global_store_dwordx4)
    }
    // obs + legal mask of the lane's own 32 cells: E = empties, M = the cells that read -1 (enc B: the
    // stones of the player who is NOT to move ... see emit()) / WHITE (enc A), both in the emitted frame.
    // half h covers bits 16h..16h+15.
    static __device__ __forceinline__ void expand_store(u32 E, u32 M, int h, signed char *obs, unsigned char *legal)
    {
        u32 o[4], g[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const u32 es = spread1(E, 4 * h + k);
            g[k] = es;
            if (B) { // empty 0, M cells 0xFF (-1), other stones 0x01 (+1)
                const u32 m2 = spread2(M, 4 * h + k);
                o[k] = (es ^ 0x01010101u) | ((m2 << 7) - m2);
            } else { // BLACK 0, WHITE 1, EMPTY 2
                o[k] = (es << 1) | spread1(M, 4 * h + k);
            }
        }
        if (obs) store16(obs, o[0], o[1], o[2], o[3]);
        if (legal) store16(legal, g[0], g[1], g[2], g[3]);
    }
    // Observation (+ legal mask) rows of the boards whose lanes pass `on`, each in its stored frame
    // (B: the mover's perspective; A: absolute).  obs / legal point at row 0 of the wavefront's first
    // board; brow = the lane's board within the wavefront.
    static __device__ __forceinline__ void emit(const LS &s, bool on, signed char *obs, unsigned char *legal, int brow,
                                                const LC &c)
    {
        if (!obs && !legal) return;
        const int v = B ? s.cur : 0;
        u32 E = empties(s, v, c);
        u32 M = B ? (v ? s.W1 : s.K0) : s.W0;
        int w0 = 32 * c.l; // first cell of the lane's window
        if (JW > 1 && (C & 31) != 0) {
            // the last word holds C & 31 cells: its lane emits the window [C-32, C) instead
            const u32 pe = dpp0<DPP_ROW_SHR1>(E), pm = dpp0<DPP_ROW_SHR1>(M);
            const bool last = c.l == JW - 1;
            E = last ? __builtin_amdgcn_alignbit(E, pe, C & 31) : E;
            M = last ? __builtin_amdgcn_alignbit(M, pm, C & 31) : M;
            w0 = last ? C - 32 : w0;
        }
        if (!(on && c.l < JW)) return;
        const size_t off = (size_t)brow * C + (size_t)w0;
        signed char *po = obs ? obs + off : nullptr;
        unsigned char *pl = legal ? legal + off : nullptr;
        if (JW == 1) { // C <= 25: [0,16) and, beyond 16 cells, the overlapping window [C-16, C)
            expand_store(E, M, 0, po, pl);
            if (C > 16) expand_store(E >> (C - 16), M >> (C - 16), 0, po ? po + (C - 16) : nullptr, pl ? pl + (C - 16) : nullptr);
        } else {
            expand_store(E, M, 0, po, pl);
            expand_store(E, M, 1, po ? po + 16 : nullptr, pl ? pl + 16 : nullptr);
        }
    }
};

// persisted flags word
__device__ __forceinline__ u32 pack_flags(int cur, int done, int agent, int winner, int pend, int prewon, int n_empty)
{
    return (u32)cur | ((u32)done << 2) | ((u32)agent << 3) | ((u32)(winner + 1) << 4) | ((u32)pend << 7) |
           ((u32)prewon << 9) | ((u32)n_empty << 12);
}

template <int N, int ENC, int MODE> __global__ __launch_bounds__(256) void k_qpb(KArgs a)
{
    using QQ = Q<N, ENC>;
    using G = Geo<N>;
    using LS = typename QQ::LS;
    constexpr bool B = ENC == 1;
    constexpr int C = G::C, L = G::L, BPW = G::BPW;

    const int lane = (int)threadIdx.x & 63;
    const typename QQ::LC c = QQ::lane_consts(lane);
    // a workgroup is 1..4 independent wavefronts (nothing is shared between them: fewer, larger workgroups only make
    // the dispatch of a launch and its retirement shorter); workgroups go round-robin over the XCDs
    const int nb = (a.E + BPW - 1) / BPW; // wavefronts of the launch = state blocks of the handle
    const int wpg = (int)blockDim.x >> 6, nwg = (int)gridDim.x, wg = (int)blockIdx.x;
    const int wgr = (nwg & 7) == 0 ? (wg & 7) * (nwg >> 3) + (wg >> 3) : wg; // contiguous env ranges per XCD
    const int blk = wgr * wpg + __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);
    if (blk >= nb) return;
    const int brow = lane / L;                                             // the lane's board within the wavefront
    const int e0 = blk * BPW;
    const int env = e0 + brow;
    const bool live = env < a.E;
    u32 *rblk = reinterpret_cast<u32 *>(a.fast + (size_t)blk * G::BLOCK_BYTES);
    // [E][4] 64-bit episode sums since create (sum_len alone passes 2^31 after hours of rollouts), behind the blocks
    longlong2 *stats = reinterpret_cast<longlong2 *>(a.fast + (size_t)nb * G::BLOCK_BYTES) + (size_t)env * 2;
    // (the self-play rollout is launched on self-play handles only, the two-sided one on two-sided handles: hexvec.hip)
    const bool two = (MODE == M_ROLLOUT1 || MODE == M_STEP_RNG || MODE == M_STEP_AGENT || MODE == M_STEP_OPP) ? false : (MODE == M_ROLLOUT2 ? true : (a.flags & HEXVEC_F_TWO_SIDED) != 0);
    const bool selfplay = !two;
    const int first = (!B && (a.flags & HEXVEC_F_FIRST_WHITE)) ? 1 : 0;
    const int gbase = lane & ~(L - 1); // first lane of the group
    // Lanes past the last env leave now: a group is all live or all dead, nothing below is shared
    // between groups (DPP carries between groups are zero by construction, ballots only ask "any").
    if (!live) return;
    HV_QPHASE(blk, 0);

    LS s;
    s.key = mix32(((u32)env + a.env_offset) * 0x9E3779B9U + a.seed_lo) ^ a.seed_hi;
    s.sr = s.sl = s.se = s.sw = 0;
    if (MODE == M_RESET && a.init) {
        s.ctr = 0;
        s.winner = -1;
        s.agent = !B ? 0 : (a.flags & HEXVEC_F_AGENT_RANDOM) ? (int)(mix32(s.key ^ 0xA5A5A5A5U) & 1u)
                  : (a.flags & HEXVEC_F_AGENT_WHITE) ? 1 : 0;
        s.pend = 0;
        QQ::fresh_regs(s, first);
        if (c.l == 0) stats[0] = stats[1] = make_longlong2(0, 0);
    } else {
        s.K0 = rblk[0 * 64 + lane];
        s.K1 = rblk[1 * 64 + lane];
        s.W0 = rblk[2 * 64 + lane];
        s.W1 = rblk[3 * 64 + lane];
        s.TK = rblk[4 * 64 + lane];
        s.DK = rblk[5 * 64 + lane];
        s.TW = rblk[6 * 64 + lane];
        s.DW = rblk[7 * 64 + lane];
        s.ctr = rblk[(8 + 0 / L) * 64 + gbase + 0 % L];
        const u32 fl = rblk[(8 + 1 / L) * 64 + gbase + 1 % L];
        const u32 ep = rblk[(8 + 2 / L) * 64 + gbase + 2 % L];
        s.cur = fl & 1;
        s.done = (fl >> 2) & 1;
        s.agent = (fl >> 3) & 1;
        s.winner = (int)((fl >> 4) & 7) - 1;
        s.pend = (int)(fl >> 7) & 3;
        s.prewon = (int)(fl >> 9) & 3;
        s.n_empty = (int)(fl >> 12) & 0xfff;
        s.ep_len = (int)(ep & 0xffffu);
        s.ep_ret = (int)ep >> 16;
    }

    s.kc = s.key + s.ctr * 0x9E3779B9U;
    HV_QPHASE(blk, 1);

    // one ply by the player to move; `act` is in the frame `fr` (0: F order, 1: F^T order).
    // Returns the reference's make_move code: -1 none, 0/1 winner, 3 illegal.
    // check = false for actions the library drew itself (always legal).
    auto ply = [&](int act, int fr, bool check) -> int {
        const bool inb = !check || (unsigned)act < (unsigned)C;
        const int yf = inb ? act / N : 0, xf = inb ? act - yf * N : 0;
        const int at = xf * N + yf;
        const int idx0 = fr ? at : (inb ? act : 0), idx1 = fr ? (inb ? act : 0) : at;
        if (check && (!inb || QQ::group_or(QQ::empties(s, 0, c) & QQ::cell_bit(idx0, c)) == 0u)) return 3;
        const int p = s.cur;
        const bool win = QQ::place(s, idx0, idx1, p, p == fr ? yf : xf, c) || ((s.prewon >> p) & 1);
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
    // env B HexEnv.step (HexSingleGame.py:233-263) on top of ply(); returns reward[0]
    auto envb_step = [&](int act, bool check) -> int {
        const int mover = s.cur;
        const int w = ply(act, mover, check);
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
    // a ply of the in-env opponent: the reply inside step, or the opening after a reset
    // (SelfplayWrapper.py:79-80 / :187-190, HexGame.py:224-230 / :259-265); returns reward[0] (B)
    auto opponent_ply = [&](int oa, bool check) -> int {
        if (B) return envb_step(oa, check);
        s.winner = ply(oa, 1, check); // hex-v0: the opponent's view is the transposed frame
        return 0;
    };
    // The same opponent ply on a board that is known to be EMPTY (just reset inside this kernel):
    // the k-th empty cell is cell k and the stone has no neighbours: nothing to search or flood.
    // (ext >= 0: the caller's action -- nothing is drawn from the board's own stream, as for any external action)
    auto opening_on_empty = [&](int ext) {
        int act = ext;
        if (ext < 0) act = (int)__umulhi(QQ::draw(s), (u32)C); // n_empty == C
        const int p = s.cur, fr = B ? p : 1;      // B: the mover's frame; hex-v0: the opponent's view
        const int yf = act / N, xf = act - yf * N, at = xf * N + yf;
        const int idx0 = fr ? at : act, idx1 = fr ? act : at;
        const u32 b0 = QQ::cell_bit(idx0, c), b1 = QQ::cell_bit(idx1, c);
        const int yp = p == fr ? yf : xf; // row of the stone in p's own frame
        const u32 ab = p ? b1 : b0;
        const u32 tb = yp == 0 ? ab : 0u, db = yp == N - 1 ? ab : 0u;
        s.K0 |= p ? 0u : b0;
        s.K1 |= p ? 0u : b1;
        s.W0 |= p ? b0 : 0u;
        s.W1 |= p ? b1 : 0u;
        s.TK |= p ? 0u : tb;
        s.DK |= p ? 0u : db;
        s.TW |= p ? tb : 0u;
        s.DW |= p ? db : 0u;
        s.n_empty -= 1;
        s.winner = -1; // make_move returned None
        s.cur = p ^ 1;
    };
    const bool opens = selfplay && (B ? s.agent != 0 : first != 0); // the opponent moves first after a reset
    // finished boards -> terminal obs, statistics, fresh board
    auto handle_done = [&](bool dn, int won, signed char *tobs_row) {
        if (!__ballot(dn)) return;
        if (tobs_row) QQ::emit(s, dn, tobs_row, nullptr, brow, c);
        if (dn) {
            s.sr += s.ep_ret;
            s.sl += s.ep_len;
            s.se += 1;
            s.sw += won;
            QQ::fresh_regs(s, first);
        }
    };
    auto do_emit = [&](signed char *o, unsigned char *lg) { QQ::emit(s, true, o, lg, brow, c); };
    const bool wr = c.l == 0; // the lane that writes the board's scalar outputs

    const size_t E = (size_t)a.E;

    if (MODE == M_SETBOARD) {
        // HexEnv.reset with a given start position (sample_board, HexSingleGame.py:217-222): the
        // stones are replayed row-major like the constructor does (:57-65); nobody has moved yet.
        const bool sel = !a.env_mask || a.env_mask[env] != 0;
        if (sel) {
            s.pend = 0;
            QQ::fresh_regs(s, first);
            const signed char *bd = a.board_in + (size_t)env * C;
            for (int cc = 0; cc < C; ++cc) {
                const int v = bd[cc];
                const int p = B ? (v == -1 ? 0 : (v == 1 ? 1 : -1)) : (v == 0 ? 0 : (v == 1 ? 1 : -1));
                if (p >= 0) {
                    const int y = cc / N, x = cc - y * N;
                    if (QQ::place(s, cc, x * N + y, p, p ? x : y, c)) s.prewon |= 1 << p;
                }
            }
            if (opens) s.pend = 2; // the opponent still owes its opening (split step)
        }
        if (wr && a.pending) a.pending[env] = (unsigned char)s.pend;
        do_emit(a.obs ? a.obs + (size_t)e0 * C : nullptr, a.legal ? a.legal + (size_t)e0 * C : nullptr);
    } else if (MODE == M_RESET) {
        const bool sel = a.init || !a.env_mask || a.env_mask[env] != 0;
        if (sel) {
            s.pend = 0;
            QQ::fresh_regs(s, first);
            if (opens) opponent_ply(a.opp_actions ? a.opp_actions[env] : QQ::rng_action(s, B ? s.cur : 1, c),
                                    a.opp_actions != nullptr);
        }
        do_emit(a.obs ? a.obs + (size_t)e0 * C : nullptr, a.legal ? a.legal + (size_t)e0 * C : nullptr);
    } else {
        constexpr bool STEPPED = MODE == M_STEP || MODE == M_STEP_AGENT || MODE == M_STEP_OPP; // actions may come from the caller
        const bool ext_a = STEPPED && a.actions, ext_o = STEPPED && a.opp_actions;
        // the half of the split step this launch is: kernels of their own in this family (hexvec.hip never launches M_STEP
        // with a.phase != 0 on a bitboard handle)
        constexpr int phase = MODE == M_STEP_AGENT ? 1 : (MODE == M_STEP_OPP ? 2 : 0);
        if (phase != 0) {
            // Split step for opponents that are policies on the device: the same slots as below,
            // one launch per slot, with the opponent's actions supplied by the caller.
            const size_t o = (size_t)env;
            signed char *tobs_row = a.tobs ? a.tobs + (size_t)e0 * C : nullptr;
            int rew = 0, won = 0;
            bool fin = false; // the env step completed in this call
            // hexvec_attach_pool: what the launch's tail needs from memory is requested HERE, before the ply, so that the
            // tail of this latency-bound launch (one wavefront per SIMD at the geometries that use it) waits for no
            // round trip of its own: the step counter, the slot -> policy table (lane l holds entry l) and, with one
            // policy, that policy's cumulative opening distribution (lane l holds cells l, l + 64, ...)
            constexpr int NCV = (C + 63) / 64;
            unsigned long long nstep = 0;
            int pos_of_lane = 0;
            float cdf_of_lane[NCV], cdf_last = 1.0f;
            const int nact = __popcll(__ballot(true)); // the live lanes are lanes 0 .. nact-1
            const bool cdf_preloaded = a.pool_slot && phase != 1 && a.pool_cdf && a.pool_npol == 1 && nact == 64;
            if (a.pool_slot) {
                nstep = a.pool_ctr[phase == 1 ? 0 : 1];
                if (phase != 1 && a.pool_cdf) {
                    if (a.pool_pos) pos_of_lane = a.pool_pos[min(lane, a.pool_size)];
                    if (cdf_preloaded) {
#pragma unroll
                        for (int k = 0; k < NCV; ++k) cdf_of_lane[k] = a.pool_cdf[min(k * 64 + lane, C - 1)];
                        cdf_last = a.pool_cdf[C - 1];
                    }
                }
            }
            if (phase == 1) { // the agent's ply (slot 1)
                if (s.pend == 0) {
                    const int act = ext_a ? a.actions[o] : QQ::rng_action(s, B ? s.cur : 0, c);
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
                    if (!fin && c.l == 0 && a.done) a.done[o] = 0; // in progress: the caller need not clear the row
                }
            } else { // the opponent's plies: replies (slot 2) and openings (slot 3)
                const int was = s.pend;
                if (was != 0) {
                    const int oa = ext_o ? a.opp_actions[o] : QQ::rng_action(s, B ? s.cur : 1, c);
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
                if (c.l == 0) {
                    if (a.reward) a.reward[o] = (float)rew;
                    if (a.done) a.done[o] = (unsigned char)s.done;
                    if (a.winner) a.winner[o] = (signed char)s.winner;
                }
            }
            const bool dn = fin && s.done;
            handle_done(dn, won, tobs_row);
            if (dn && opens) s.pend = 2; // the opponent opens on the fresh board
            if (a.pool_slot) {
                // hexvec_attach_pool: SelfPlayEnv.setup_opponents (SelfplayWrapper.py:97-104) for the boards that
                // start a new episode here -- exactly the boards and draws of hexvec_pick_opponents(when = 3, salt 1)
                // behind the agent's launch / (when = 4, salt 4) behind the opponent's.  The step counter comes from
                // the handle's own two words: each launch reads the one the OTHER phase writes, so no launch reads a
                // word it writes.
                const bool first_half = phase == 1;
                if (first_half && wr) a.pool_ended[o] = dn ? 1 : 0;
                // a board qualifies at one point at most: ended and the agent moves first on the fresh board
                // (salt 1 after the agent's ply, 5 after the reply), or a fresh board on which the opponent is about
                // to open (salt 2 / 4)
                const bool opp_opens = s.pend == 2 && (first_half || dn);
                const bool sel = opp_opens || (dn && s.pend == 0);
                long long nslot = 0;
                if (sel) {
                    const u32 salt = first_half ? (opp_opens ? 2u : 1u) : (opp_opens ? 4u : 5u);
                    nslot = pick_slot((u32)env, nstep, a.pool_seed_lo, a.pool_seed_hi, salt, a.pool_prob, a.pool_size);
                    if (wr) a.pool_slot[o] = nslot;
                }
                if (!first_half) {
                    // the opening of a board that ended on the reply, from the picked opponent's cumulative opening
                    // distribution on the empty board (SelfplayWrapper.py:79-80): first cell whose entry exceeds u
                    // (wave-cooperative: few boards of a wavefront end in one step, so for each of them ALL lanes test one
                    // cell of its opponent's table each -- one load and one ballot per 64 cells instead of a scan by the
                    // board's own lanes)
                    u64 om = a.pool_cdf ? __ballot(opp_opens && c.l == 0) : 0ull;
                    if (om) {
                        float u = 0.0f;
                        if (opp_opens) {
                            const u32 k = row_key((u32)env, nstep + 1ULL, a.pool_seed_lo, a.pool_seed_hi, 6u);
                            u = ((float)(mix32(k) >> 8) + 0.5f) * (1.0f / 16777216.0f);
                        }
                        const int slot_ix = min(max((int)nslot + 1, 0), a.pool_size); // index into the slot -> policy table
                        int open_act = 0;
                        while (om) {
                            const int src = __ffsll((long long)om) - 1; // first lane of the board's group
                            om &= om - 1;
                            const float ub = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(u), src));
                            int below = 0; // cells whose entry is <= u
                            bool bad;
                            if (cdf_preloaded) {
#pragma unroll
                                for (int k = 0; k < NCV; ++k)
                                    below += __popcll(__ballot(k * 64 + lane < C && cdf_of_lane[k] <= ub));
                                bad = !(cdf_last == 1.0f);
                            } else {
                                int pol = 0;
                                if (a.pool_pos) { // entry slot_ix of the table: from the lane that holds it, or from memory
                                    const int six = __builtin_amdgcn_readlane(slot_ix, src);
                                    pol = six < nact ? __builtin_amdgcn_readlane(pos_of_lane, six) : a.pool_pos[six];
                                }
                                pol = min(max(pol, 0), a.pool_npol - 1);
                                const float *cdf = a.pool_cdf + (size_t)pol * C;
                                for (int c0 = 0; c0 < C; c0 += nact) {
                                    const int cc = c0 + lane;
                                    below += __popcll(__ballot(cc < C && cdf[cc] <= ub));
                                }
                                bad = !(cdf[C - 1] == 1.0f);
                            }
                            // (bad: not a distribution, e.g. a NaN logit: cell 0, counted)
                            if (bad && lane == src && a.pool_bad) atomicAdd(a.pool_bad, 1ULL);
                            if (gbase == src) open_act = bad ? 0 : min(below, C - 1);
                        }
                        if (opp_opens) {
                            opening_on_empty(open_act);
                            s.pend = 0;
                        }
                    }
                    if (blk == 0 && lane == 0) { // the step is complete: publish its number for the sampling calls
                        a.pool_ctr[0] = nstep + 1ULL;
                        *a.pool_state = nstep + 1ULL;
                    }
                } else if (blk == 0 && lane == 0) {
                    a.pool_ctr[1] = nstep;
                }
            }
            if (wr && a.pending) a.pending[o] = (unsigned char)s.pend;
            do_emit(a.obs ? a.obs + (size_t)e0 * C : nullptr, a.legal ? a.legal + (size_t)e0 * C : nullptr);
        } else
        for (int t = 0; t < a.T; ++t) {
            const size_t ob = (size_t)t * E + (size_t)e0; // first board of this block in row t
            const size_t o = ob + (size_t)brow;
            signed char *tobs_row = a.tobs ? a.tobs + ob * C : nullptr;
            if (MODE == M_ROLLOUT2) do_emit(a.obs ? a.obs + ob * C : nullptr, a.legal ? a.legal + ob * C : nullptr);
            // ---- slot 1: the agent's ply (two-sided: the mover's)
            int rew = 0, won = 0, f_done = 0, f_winner = -1, act = 0;
            bool reply = false; // this board still owes the opponent's reply
            {
                act = ext_a ? a.actions[o] : QQ::rng_action(s, B ? s.cur : 0, c); // B: the mover's frame
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
                    const int rb = envb_step(act, ext_a);
                    rew = s.agent == 0 ? rb : -rb;
                    reply = !s.done;
                } else { // hex-v0 HexEnv.step (HexGame.py:244-295), agent BLACK, first half
                    s.winner = ply(act, 0, ext_a);
                    if (s.winner == 3) s.done = 1;
                    rew = s.winner == 0 ? 1 : (s.winner == 3 ? -100 : 0);
                    reply = !s.done;
                }
                if (!reply) { // the step's result is final for this board
                    if (!two) {
                        s.ep_ret += rew;
                        won = s.done && rew > 0;
                    }
                    s.ep_len += 1;
                    f_done = s.done;
                    f_winner = s.winner;
                }
            }
            const bool dn1 = !reply && s.done;
            handle_done(dn1, won, tobs_row);
            // ---- slot 2: the opponent's reply, or its opening on the boards just reset
            if (selfplay) {
                const bool open2 = dn1 && opens;
                if (reply || open2) {
                    const bool ext = reply && ext_o;
                    const int oa = ext ? a.opp_actions[o] : QQ::rng_action(s, B ? s.cur : 1, c);
                    const int rb = opponent_ply(oa, ext);
                    if (reply) {
                        rew = B ? (s.agent == 0 ? rb : -rb)
                                : (s.winner == 0 ? 1 : (s.winner == 1 ? -1 : (s.winner == 3 ? -100 : 0)));
                        s.ep_ret += rew;
                        s.ep_len += 1;
                        won = s.done && rew > 0;
                        f_done = s.done;
                        f_winner = s.winner;
                    }
                }
                const bool dn2 = reply && s.done;
                handle_done(dn2, won, tobs_row);
                // ---- slot 3: openings of the boards that ended on the reply
                if (dn2 && opens) opening_on_empty(-1);
            }
            if (wr) {
                if (a.reward) a.reward[o] = (float)rew;
                if (a.done) a.done[o] = (unsigned char)f_done;
                if (a.winner) a.winner[o] = (signed char)f_winner;
                if (a.actions_out) a.actions_out[o] = act;
            }
            if (MODE != M_ROLLOUT2) do_emit(a.obs ? a.obs + ob * C : nullptr, a.legal ? a.legal + ob * C : nullptr);
        }
    }

    // state back to HBM
    HV_QPHASE(blk, 2);
    rblk[0 * 64 + lane] = s.K0;
    rblk[1 * 64 + lane] = s.K1;
    rblk[2 * 64 + lane] = s.W0;
    rblk[3 * 64 + lane] = s.W1;
    rblk[4 * 64 + lane] = s.TK;
    rblk[5 * 64 + lane] = s.DK;
    rblk[6 * 64 + lane] = s.TW;
    rblk[7 * 64 + lane] = s.DW;
    const u32 fl = pack_flags(s.cur, s.done, s.agent, s.winner, s.pend, s.prewon, s.n_empty);
    const u32 ep = ((u32)s.ep_len & 0xffffu) | ((u32)s.ep_ret << 16);
#pragma unroll
    for (int k = 0; k < G::NSC; ++k)
        if (c.l == k % L) rblk[(8 + k / L) * 64 + lane] = k == 0 ? s.ctr : (k == 1 ? fl : ep);
    if (c.l == 0 && s.se != 0) { // episode sums: touched only when an episode ended in this launch
        longlong2 v0 = stats[0], v1 = stats[1];
        v0.x += s.sr;
        v0.y += s.sl;
        v1.x += s.se;
        v1.y += s.sw;
        stats[0] = v0;
        stats[1] = v1;
    }
    HV_QPHASE(blk, 3);
}

// ---- cold kernels on the state ----
template <int N, int ENC>
__global__ void k_qpb_get_state(const unsigned char *fast, int E, signed char *board, unsigned char *cur)
{
    using G = Geo<N>;
    constexpr int C = G::C, L = G::L, BPW = G::BPW;
    const long long gid = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= (long long)E * (C + 1)) return;
    const int env = (int)(gid / (C + 1)), r = (int)(gid - (long long)env * (C + 1));
    const int blk = env / BPW, g0 = (env % BPW) * L;
    const u32 *rb = reinterpret_cast<const u32 *>(fast + (size_t)blk * G::BLOCK_BYTES);
    const int cp = (int)(rb[(8 + 1 / L) * 64 + g0 + 1 % L] & 1u);
    if (r < C) {
        if (board) {
            const int v = (ENC == 1 && cp) ? 1 : 0; // the stored frame
            const u32 k = rb[(0 + v) * 64 + g0 + (r >> 5)], w = rb[(2 + v) * 64 + g0 + (r >> 5)];
            const int isk = (k >> (r & 31)) & 1, isw = (w >> (r & 31)) & 1;
            signed char val;
            if (ENC == 1) val = (signed char)(v ? (isk ? 1 : (isw ? -1 : 0)) : (isk ? -1 : (isw ? 1 : 0)));
            else val = (signed char)(isk ? 0 : (isw ? 1 : 2));
            board[(size_t)env * C + r] = val;
        }
    } else if (cur) {
        cur[env] = (unsigned char)cp;
    }
}

template <int N>
__global__ __launch_bounds__(256) void k_qpb_stats(const unsigned char *fast, int nblocks, int E, double *out4,
                                                   unsigned char *agents)
{
    using G = Geo<N>;
    constexpr int L = G::L, BPW = G::BPW;
    __shared__ double sh[4][256];
    double acc[4] = {0, 0, 0, 0};
    const longlong2 *stats = reinterpret_cast<const longlong2 *>(fast + (size_t)nblocks * G::BLOCK_BYTES);
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < E; i += gridDim.x * blockDim.x) {
        if (out4) {
            const longlong2 v0 = stats[(size_t)i * 2], v1 = stats[(size_t)i * 2 + 1];
            acc[0] += (double)v0.x; // exact below 2^53
            acc[1] += (double)v0.y;
            acc[2] += (double)v1.x;
            acc[3] += (double)v1.y;
        }
        if (agents) {
            const u32 *rb = reinterpret_cast<const u32 *>(fast + (size_t)(i / BPW) * G::BLOCK_BYTES);
            agents[i] = (unsigned char)((rb[(8 + 1 / L) * 64 + (i % BPW) * L + 1 % L] >> 3) & 1u);
        }
    }
    if (!out4) return;
    for (int k = 0; k < 4; ++k) sh[k][threadIdx.x] = acc[k];
    __syncthreads();
    for (int st = 128; st > 0; st >>= 1) {
        if ((int)threadIdx.x < st)
            for (int k = 0; k < 4; ++k) sh[k][threadIdx.x] += sh[k][threadIdx.x + st];
        __syncthreads();
    }
    if (threadIdx.x < 4) atomicAdd(&out4[threadIdx.x], sh[threadIdx.x][0]); // sums of integers: exact in any order
}

} // namespace qpb

#ifdef HV_QSTAMPS
extern "C" int hexvec_debug_qphases(unsigned long long *host_dst, int nblocks)
{
    return (int)hipMemcpyFromSymbol(host_dst, HIP_SYMBOL(g_qphase), (size_t)nblocks * 4 * sizeof(unsigned long long));
}
#endif

// ---- launchers of this translation unit (HV_QPB_SIZES(X) = the board sizes it instantiates) ----
template <int N, int ENC, int MODE> static hipError_t qpb_launch1(const KArgs &a, hipStream_t st)
{
    const int blocks = (a.E + qpb::Geo<N>::BPW - 1) / qpb::Geo<N>::BPW;
    // From one wavefront per SIMD of the chip on, four wavefronts per workgroup: a quarter of the workgroups to
    // dispatch and retire (11x11 / 65 536 host-stepped: 9.63 -> 9.53 us per launch; the config-3 collection step
    // 50.85 -> 50.4 us; same process order, alternating).  Smaller launches keep one wavefront per workgroup, spread
    // over as many CUs as there are wavefronts.  HEXVEC_QPB_WPG = 1..4 forces one form (A/B experiments).
    static const int wpg_env = [] { const char *ev = getenv("HEXVEC_QPB_WPG"); return ev ? atoi(ev) : 0; }();
    const int wpg = wpg_env >= 1 ? (wpg_env > 4 ? 4 : wpg_env) : (blocks >= 1024 ? 4 : 1);
    hipLaunchKernelGGL((qpb::k_qpb<N, ENC, MODE>), dim3((blocks + wpg - 1) / wpg), dim3(64 * wpg), 0, st, a);
    return hipGetLastError();
}
template <int N, int ENC> static hipError_t qpb_launch_mode(int mode, const KArgs &a, hipStream_t st)
{
    switch (mode) {
    case M_ROLLOUT2: return qpb_launch1<N, ENC, M_ROLLOUT2>(a, st);
    case M_STEP: return qpb_launch1<N, ENC, M_STEP>(a, st);
    case M_STEP_RNG: return qpb_launch1<N, ENC, M_STEP_RNG>(a, st);
    case M_STEP_AGENT: return qpb_launch1<N, ENC, M_STEP_AGENT>(a, st);
    case M_STEP_OPP: return qpb_launch1<N, ENC, M_STEP_OPP>(a, st);
    case M_ROLLOUT1: return qpb_launch1<N, ENC, M_ROLLOUT1>(a, st);
    case M_RESET: return qpb_launch1<N, ENC, M_RESET>(a, st);
    case M_SETBOARD: return qpb_launch1<N, ENC, M_SETBOARD>(a, st);
    default: return hipErrorInvalidValue;
    }
}
hipError_t HV_QPB_NAME(launch)(int N, int enc, int mode, const KArgs &a, hipStream_t st)
{
    switch (N) {
#define HV_CASE(n) case n: return enc ? qpb_launch_mode<n, 1>(mode, a, st) : qpb_launch_mode<n, 0>(mode, a, st);
        HV_QPB_SIZES(HV_CASE)
#undef HV_CASE
    default: return hipErrorInvalidValue;
    }
}
hipError_t HV_QPB_NAME(get_state)(const unsigned char *fast, int N, int enc, int E, signed char *board,
                                  unsigned char *cur, hipStream_t st)
{
    const long long total = (long long)E * (N * N + 1);
    const dim3 g((unsigned)((total + 255) / 256)), b(256);
    switch (N) {
#define HV_CASE(n)                                                                                  \
    case n:                                                                                         \
        if (enc) hipLaunchKernelGGL((qpb::k_qpb_get_state<n, 1>), g, b, 0, st, fast, E, board, cur); \
        else hipLaunchKernelGGL((qpb::k_qpb_get_state<n, 0>), g, b, 0, st, fast, E, board, cur);     \
        break;
        HV_QPB_SIZES(HV_CASE)
#undef HV_CASE
    default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}
// out4 must have been zeroed on the same stream (the blocks add their partial sums)
hipError_t HV_QPB_NAME(stats)(const unsigned char *fast, int N, int E, double *out4, unsigned char *agents,
                              hipStream_t st)
{
    const int grid = (E + 255) / 256 < 1024 ? (E + 255) / 256 : 1024;
    switch (N) {
#define HV_CASE(n)                                                                                           \
    case n:                                                                                                  \
        hipLaunchKernelGGL((qpb::k_qpb_stats<n>), dim3(grid), dim3(256), 0, st, fast,                        \
                           (E + qpb::Geo<n>::BPW - 1) / qpb::Geo<n>::BPW, E, out4, agents);                  \
        break;
        HV_QPB_SIZES(HV_CASE)
#undef HV_CASE
    default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

} // namespace hv
