// hexvec_wpb.hip -- wave-per-board kernels of libhexvec: the LITERAL engine (gfx950 / CDNA4).
// (Throughput path: hexvec_lpb.h.  Host side / C ABI: hexvec.hip.)
//
// One 64-lane wavefront owns one board.  Everything the reference keeps in
// numpy arrays lives in four VGPRs of that wave, one dword per lane:
//
//   L0, L1 : the two padded label planes `regions[2][N+2][N+2]`
//            (HexGame.py:38-45), one BYTE per cell, striped: padded cell q
//            is byte (q >> 6) of lane (q & 63).  (N+2)^2 <= 256  =>  N <= 14.
//   F      : the board in BLACK's frame, one byte per cell (0 empty, 1 BLACK,
//            2 WHITE, 3 off-board), striped the same way by c = y*N + x.
//   FT     : the same board transposed (c = x*N + y): the frame the WHITE
//            mover / the hex-v0 opponent is shown (invert_board,
//            HexSingleGame.py:265-271, HexGame.py:297-303) without ever
//            transposing anything: a stone is one byte written in each.
//
// A ply is wave-uniform control flow (the action is a scalar) around a few
// full-wave vector ops:
//   * stone placement       : one lane ORs one byte into F and FT
//   * neighbour labels      : lanes 0..6 fetch the 6 hex neighbours + centre
//                             of the padded cell with ONE ds_bpermute_b32
//   * min label             : 3 DPP row_shr v_min_u32 + v_readlane
//   * relabel (flood_fill's `regions[regions == label] = new`,
//     HexGame.py:141-142)   : byte-parallel compare/select on the whole plane,
//                             4 cells per lane, ~10 VALU per merged label
//   * k-th empty cell (random policy): ballot per stripe + mbcnt rank select
//   * obs / legal mask      : v_perm_b32 as a 4-entry byte LUT, byte stores
//                             that are contiguous across the wave
// No LDS allocation is needed: the planes never leave the register file
// between plies (rollout kernels) and the LDS crossbar is used only through
// ds_bpermute.  Integer/byte work only: no MFMA.
//
// The engine semantics (illegal ply -> 3 with no state change, plies after the
// end still applied, the draw branch of engine B, the sticky win corner) are
// literal restatements of HexGame.py:85-142 and HexSingleGame.py:88-153; the
// env layers follow HexSingleGame.py:233-263, SelfplayWrapper.py:69-89/174-199
// and HexGame.py:206-295.  Bit-exactness is checked against oracle/ and the
// golden fixtures in tests/.

#include "hexvec_common.h"

namespace hv {

template <int N> struct Geo {
    static constexpr int W = N + 2, C = N * N, P = W * W;
    static constexpr int J = (C + 63) / 64, JP = (P + 63) / 64;
};

// per-wave state.  The u32 members at the top are VGPRs, the rest is uniform.
struct WS {
    u32 L0, L1, F, FT;
    u32 li0, li1, fi; // reset images of L0, L1, F (per-lane constants)
    int nboff;        // lanes 0..5: padded offset of hex neighbour i; else 0
    int cur_e;        // simulator.current_player_num
    int phi;          // HexEnv.current_player_num of env B (= whose frame the stored board is in)
    int sim_done;     // simulator.done
    int winner;       // HexEnv.winner: -1 None, 0, 1, 3
    int n_empty;      // empty cells (== empty_fields of engine B)
    int nx0, nx1;     // region_counter
    int agent;
    u32 ctr, key;
    int ep_ret, ep_len;
};

template <int N, int ENC> struct Eng {
    using G = Geo<N>;
    static constexpr bool B = ENC == 1;

    static __device__ __forceinline__ void consts(WS &s, u32 lane)
    {
        u32 l0 = 0, l1 = 0, f = 0;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int q = j * 64 + (int)lane;
            if (q < G::P) {
                const int r = q / G::W, c = q - r * G::W;
                const u32 b0 = r == 0 ? 1u : (r == N + 1 ? 2u : 0u); // BLACK: rows   (HexGame.py:43,45)
                const u32 b1 = c == 0 ? 1u : (c == N + 1 ? 2u : 0u); // WHITE: columns (HexGame.py:42,44)
                l0 |= b0 << (8 * j);
                l1 |= b1 << (8 * j);
            }
            f |= (q < G::C ? 0u : 3u) << (8 * j);
        }
        s.li0 = l0;
        s.li1 = l1;
        s.fi = f;
        const int W = G::W;
        s.nboff = lane == 0 ? -W : lane == 1 ? -W + 1 : lane == 2 ? -1 : lane == 3 ? 1
                : lane == 4 ? W - 1 : lane == 5 ? W : 0;
    }

    // HexEnv.reset / HexGame.__init__ on an empty board (regions as cached, HexGame.py:191)
    static __device__ __forceinline__ void fresh(WS &s, int first)
    {
        s.L0 = s.li0;
        s.L1 = s.li1;
        s.F = s.fi;
        s.FT = s.fi;
        s.cur_e = first;
        s.phi = 0;
        s.sim_done = 0;
        s.n_empty = G::C;
        s.nx0 = 3;
        s.nx1 = 3;
        s.ep_ret = 0;
        s.ep_len = 0;
    }

    // flood_fill (HexGame.py:124-142) on plane `Lc` at padded cell q; returns new plane.
    static __device__ __forceinline__ u32 flood_fill(u32 Lc, int q, int &next_label, u32 lane, int nboff)
    {
        // neighbourhood: 6 hex neighbours (+ the centre, lane 6) of the padded cell
        const int qi = q + nboff;
        const u32 w = (u32)__builtin_amdgcn_ds_bpermute((qi & 63) << 2, (int)Lc);
        u32 lab = (w >> ((qi >> 6) << 3)) & 0xffu;
        lab = lane < 7 ? lab : 0u;
        // smallest non-zero label over lanes 0..7
        u32 t = (lab - 1u) & 0xffu; // 0 -> 255
        t = min(t, (u32)__builtin_amdgcn_update_dpp(0xff, (int)t, 0x111, 0xf, 0xf, false));
        t = min(t, (u32)__builtin_amdgcn_update_dpp(0xff, (int)t, 0x112, 0xf, 0xf, false));
        t = min(t, (u32)__builtin_amdgcn_update_dpp(0xff, (int)t, 0x114, 0xf, 0xf, false));
        const u32 m1 = rdlane(t, 7);
        u32 m;
        if (m1 == 0xffu) { // no adjacent region: fresh label (:135-137)
            m = (u32)next_label;
            next_label += 1;
        } else {
            m = m1 + 1u;
        }
        Lc = put_byte(Lc, q, m, lane); // :136 / :140
        // :141-142 every other adjacent label is rewritten plane-wide
        u64 pend = __ballot(lab != 0u && lab != m);
        const u32 mrep = m * 0x01010101u;
        while (pend) {
            const int i = __ffsll((long long)pend) - 1;
            const u32 l = rdlane(lab, i);
            const u32 x = Lc ^ (l * 0x01010101u);
            const u32 z = ~(((x & 0x7f7f7f7fu) + 0x7f7f7f7fu) | x | 0x7f7f7f7fu); // 0x80 where byte == l
            const u32 msk = (z >> 7) * 0xffu;
            Lc = (Lc & ~msk) | (mrep & msk);
            pend &= ~__ballot(lab == l);
        }
        return Lc;
    }

    // HexGame.fast_move (HexGame.py:85-111 / HexSingleGame.py:88-122).
    // `a` is in the stored view.  Returns -1 None / 0 / 1 winner / 3 illegal.
    static __device__ __forceinline__ int make_move(WS &s, int a, u32 lane)
    {
        if ((unsigned)a >= (unsigned)G::C) return 3; // outside the board: treated as illegal
        const int y = a / N, x = a - y * N;
        const int at = x * N + y;
        int cF, cT, py, px;
        u32 code;
        if (B) {
            // stored = F when phi == 0, -F^T when phi == 1; the stone written is
            // always "-1 in the stored view" (HexSingleGame.py:100)
            cF = s.phi ? at : a;
            cT = s.phi ? a : at;
            code = s.phi ? 2u : 1u;
            py = s.cur_e ? x : y; // :103-106
            px = s.cur_e ? y : x;
        } else {
            cF = a;
            cT = at;
            code = (u32)s.cur_e + 1u; // HexGame.py:95
            py = y;
            px = x;
        }
        if (byte_at(s.F, cF) != 0u) return 3; // is_valid_move (:74-76 / :77-79)
        s.F = put_byte(s.F, cF, code, lane);
        s.FT = put_byte(s.FT, cT, code, lane);
        s.n_empty -= 1;
        const int q = (py + 1) * G::W + (px + 1);
        int ret = -1;
        u32 Lc = s.cur_e ? s.L1 : s.L0;
        int nl = s.cur_e ? s.nx1 : s.nx0;
        Lc = flood_fill(Lc, q, nl, lane, s.nboff);
        if (s.cur_e) {
            s.L1 = Lc;
            s.nx1 = nl;
        } else {
            s.L0 = Lc;
            s.nx0 = nl;
        }
        if (byte_at(Lc, G::P - 1) == 1u) { // regions[cur][-1,-1] == 1 (:101-105 / :110-116)
            s.sim_done = 1;
            ret = s.cur_e;
        } else if (B && s.n_empty <= 0) { // HexSingleGame.py:117-119
            s.sim_done = 1;
        }
        s.cur_e ^= 1; // :110 / :121
        return ret;
    }

    // HexEnv.step of env B (HexSingleGame.py:233-263); returns reward[0]
    static __device__ __forceinline__ int envb_step(WS &s, int a, u32 lane)
    {
        const int w = make_move(s, a, lane);
        s.winner = w;
        if (w == 3) s.sim_done = 1;
        const int r = w == s.phi ? 1 : (w == (s.phi ^ 1) ? -1 : 0);
        const int rb = s.phi == 0 ? r : -r;
        s.phi ^= 1; // :259-260 (the inversion itself is index arithmetic here)
        return rb;
    }

    // k-th empty cell of view V in row-major order (requires k < #empties)
    static __device__ __forceinline__ int kth_empty(u32 V, u32 k, u32 lane)
    {
        int cell = 0;
        bool found = false;
#pragma unroll
        for (int j = 0; j < G::J; ++j) {
            const bool e = ((V >> (8 * j)) & 0xffu) == 0u;
            const u64 m = __ballot(e);
            const u32 cnt = (u32)__popcll(m);
            if (!found) {
                if (k < cnt) {
                    const u64 hit = __ballot(e && mbcnt(m) == k);
                    cell = j * 64 + (__ffsll((long long)hit) - 1);
                    found = true;
                } else {
                    k -= cnt;
                }
            }
        }
        return cell;
    }

    // uniform-legal action for the player to move, in the view it is shown
    // (SelfplayWrapper.py:16-22 / minihex/__init__.py:8-12 with the counter RNG)
    static __device__ __forceinline__ int rng_action(WS &s, u32 V, u32 lane)
    {
        const u32 r = mix32(s.key + s.ctr * 0x9E3779B9U);
        s.ctr += 1;
        if (s.n_empty <= 0) return 0;
        const u32 k = __umulhi(r, (u32)s.n_empty);
        return kth_empty(V, k, lane);
    }
    static __device__ __forceinline__ u32 stored_view(const WS &s) { return (B && s.phi) ? s.FT : s.F; }

    // reset of one env incl. the opponent's opening ply in self-play handles
    static __device__ __forceinline__ void reset_env(WS &s, u32 flags, const int *opp, int env, u32 lane)
    {
        const bool two = (flags & HEXVEC_F_TWO_SIDED) != 0;
        if (B) {
            fresh(s, 0);
            if (!two && s.agent != 0) { // SelfplayWrapper.py:79-80
                const int oa = opp ? rfl(opp[env]) : rng_action(s, stored_view(s), lane);
                envb_step(s, oa, lane);
            }
        } else {
            const int first = (flags & HEXVEC_F_FIRST_WHITE) ? 1 : 0;
            fresh(s, first);
            if (!two && first != 0) { // HexGame.py:224-230
                const int oa = opp ? rfl(opp[env]) : rng_action(s, s.FT, lane);
                const int oy = oa / N, ox = oa - oy * N;
                s.winner = make_move(s, ox * N + oy, lane); // :341-346
            }
        }
    }

    // write the stored board and its legal mask for one env
    static __device__ __forceinline__ void emit(const WS &s, signed char *obs, unsigned char *legal, u32 lane)
    {
        const u32 V = stored_view(s);
        if (obs) {
            const u32 lut = B ? (s.phi ? 0x00FF0100u : 0x0001FF00u) : 0x00010002u;
            const u32 o = __builtin_amdgcn_perm(0u, lut, V);
#pragma unroll
            for (int j = 0; j < G::J; ++j) {
                const int c = j * 64 + (int)lane;
                if (c < G::C) obs[c] = (signed char)((o >> (8 * j)) & 0xffu);
            }
        }
        if (legal) {
            const u32 o = __builtin_amdgcn_perm(0u, 0x00000001u, V);
#pragma unroll
            for (int j = 0; j < G::J; ++j) {
                const int c = j * 64 + (int)lane;
                if (c < G::C) legal[c] = (unsigned char)((o >> (8 * j)) & 0xffu);
            }
        }
    }
};

// hot word 3: cur_e | phi<<1 | sim_done<<2 | agent<<3 | (winner+1)<<4
// hot word 4: n_empty | nx0<<8 | nx1<<16
template <int N, int ENC, int MODE>
__global__ __launch_bounds__(64 * WPB) void k_hex(KArgs a)
{
    using E_ = Eng<N, ENC>;
    using G = Geo<N>;
    constexpr bool B = ENC == 1;
    const u32 lane = threadIdx.x & 63u;
    const int wave = rfl((int)(threadIdx.x >> 6));
    // consecutive envs on one XCD (its L2 then sees whole output lines):
    // workgroups are dealt round-robin over the 8 XCDs, so give XCD x the
    // x-th contiguous slice of the grid.  Placement only affects speed.
    const int nb = (int)gridDim.x, b = (int)blockIdx.x;
    const int blk = (nb & 7) == 0 ? (b & 7) * (nb >> 3) + (b >> 3) : b;
    const int env = blk * WPB + wave;
    if (env >= a.E) return;

    WS s;
    E_::consts(s, lane);
    s.key = mix32(((u32)env + a.env_offset) * 0x9E3779B9U + a.seed_lo) ^ a.seed_hi;
    u32 *pl = a.planes + (size_t)env * PLANE_WORDS;
    u32 *hot = a.hot + (size_t)env * HOT_WORDS;
    const bool two = (a.flags & HEXVEC_F_TWO_SIDED) != 0;
    const bool auto_reset = (a.flags & HEXVEC_F_NO_AUTO_RESET) == 0;

    if (MODE == M_SETSTATE) {
        const u32 hw = lane < HOT_WORDS ? hot[lane] : 0u;
        s.ctr = rdlane(hw, 0);
        s.agent = (rdlane(hw, 3) >> 3) & 1;
        s.winner = -1;
        E_::fresh(s, 0);
    } else if (MODE == M_RESET && a.init) {
        s.ctr = 0;
        s.winner = -1;
        s.agent = !B ? 0 : (a.flags & HEXVEC_F_AGENT_RANDOM) ? (int)(mix32(s.key ^ 0xA5A5A5A5U) & 1u)
                  : (a.flags & HEXVEC_F_AGENT_WHITE) ? 1 : 0;
        if (lane < STAT_WORDS) a.stats[(size_t)env * STAT_WORDS + lane] = 0;
        E_::fresh(s, 0);
    } else {
        s.L0 = pl[lane];
        s.L1 = pl[64 + lane];
        const u32 fb = pl[128 + lane];
        s.F = fb & 0x0f0f0f0fu;
        s.FT = (fb >> 4) & 0x0f0f0f0fu;
        const u32 hw = lane < HOT_WORDS ? hot[lane] : 0u;
        s.ctr = rdlane(hw, 0);
        s.ep_ret = (int)rdlane(hw, 1);
        s.ep_len = (int)rdlane(hw, 2);
        const u32 fl = rdlane(hw, 3), cn = rdlane(hw, 4);
        s.cur_e = fl & 1;
        s.phi = (fl >> 1) & 1;
        s.sim_done = (fl >> 2) & 1;
        s.agent = (fl >> 3) & 1;
        s.winner = (int)((fl >> 4) & 7) - 1;
        s.n_empty = cn & 0xff;
        s.nx0 = (cn >> 8) & 0xff;
        s.nx1 = (cn >> 16) & 0xff;
    }

    const size_t E = (size_t)a.E;
    int sr = 0, sl = 0, se = 0, sw = 0; // episode sums finished in this launch

    if (MODE == M_SETSTATE) {
        // HexGame.__init__ (HexGame.py:21-68 / HexSingleGame.py:26-71) on a given board
        const signed char *bd = a.board_in + (size_t)env * G::C;
        E_::fresh(s, a.cur_in);
        if (a.labels_in) { // connected_stones adopted (:46-47), region_counter = max + 1 (:49-51)
            u32 f = s.fi, ft = s.fi, l0 = 0, l1 = 0, mx0 = 0, mx1 = 0;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int q = j * 64 + (int)lane;
                if (q < G::P) {
                    const u32 b0 = a.labels_in[((size_t)env * 2 + 0) * G::P + q];
                    const u32 b1 = a.labels_in[((size_t)env * 2 + 1) * G::P + q];
                    l0 |= b0 << (8 * j);
                    l1 |= b1 << (8 * j);
                    mx0 = max(mx0, b0);
                    mx1 = max(mx1, b1);
                }
                if (q < G::C) {
                    const int y = q / N, x = q - y * N;
                    const int v = bd[q], vt = bd[x * N + y];
                    const u32 c = B ? (v == -1 ? 1u : (v == 1 ? 2u : 0u)) : (v == 0 ? 1u : (v == 1 ? 2u : 0u));
                    const u32 ct = B ? (vt == -1 ? 1u : (vt == 1 ? 2u : 0u)) : (vt == 0 ? 1u : (vt == 1 ? 2u : 0u));
                    f |= c << (8 * j);
                    ft |= ct << (8 * j);
                }
            }
            for (int off = 32; off >= 1; off >>= 1) {
                mx0 = max(mx0, (u32)__shfl_xor((int)mx0, off));
                mx1 = max(mx1, (u32)__shfl_xor((int)mx1, off));
            }
            s.L0 = l0;
            s.L1 = l1;
            s.F = f;
            s.FT = ft;
            s.nx0 = (int)rflu(mx0) + 1;
            s.nx1 = (int)rflu(mx1) + 1;
            int ne = 0;
#pragma unroll
            for (int j = 0; j < G::J; ++j) ne += __popcll(__ballot(((f >> (8 * j)) & 0xffu) == 0u));
            s.n_empty = ne;
        } else { // replay the stones row-major through flood_fill (:53-61 / :57-65)
            for (int c = 0; c < G::C; ++c) {
                const int v = rfl((int)bd[c]);
                const int p = B ? (v == -1 ? 0 : (v == 1 ? 1 : -1)) : (v == 0 ? 0 : (v == 1 ? 1 : -1));
                if (p < 0) continue;
                const int y = c / N, x = c - y * N;
                s.F = put_byte(s.F, c, (u32)p + 1u, lane);
                s.FT = put_byte(s.FT, x * N + y, (u32)p + 1u, lane);
                s.n_empty -= 1;
                const int q = (y + 1) * G::W + (x + 1); // untransposed for both colours
                if (p) s.L1 = E_::flood_fill(s.L1, q, s.nx1, lane, s.nboff);
                else s.L0 = E_::flood_fill(s.L0, q, s.nx0, lane, s.nboff);
            }
        }
        s.winner = -1;
    } else if (MODE == M_RESET) {
        const bool sel = a.env_mask ? (rfl((int)a.env_mask[env]) != 0) : true;
        if (sel || a.init) E_::reset_env(s, a.flags, a.opp_actions, env, lane);
        E_::emit(s, a.obs ? a.obs + (size_t)env * G::C : nullptr,
                 a.legal ? a.legal + (size_t)env * G::C : nullptr, lane);
    } else {
        for (int t = 0; t < a.T; ++t) {
            const size_t o = (size_t)t * E + (size_t)env;
            if (MODE == M_APPLY) {
                const int act = rfl(a.actions[o]);
                int ret, rb = 0;
                if (B) {
                    rb = E_::envb_step(s, act, lane);
                    ret = s.winner;
                } else {
                    ret = E_::make_move(s, act, lane);
                }
                if (lane == 0) {
                    if (a.winner) a.winner[o] = (signed char)ret;
                    if (a.done) a.done[o] = (unsigned char)s.sim_done;
                    if (a.rblack) a.rblack[o] = (signed char)rb;
                }
                E_::emit(s, a.obs ? a.obs + o * G::C : nullptr, a.legal ? a.legal + o * G::C : nullptr, lane);
                continue;
            }
            if (MODE == M_ROLLOUT2) // record what the mover sees
                E_::emit(s, a.obs ? a.obs + o * G::C : nullptr, a.legal ? a.legal + o * G::C : nullptr, lane);

            const int act = (MODE == M_STEP && a.actions) ? rfl(a.actions[o])
                                                           : E_::rng_action(s, E_::stored_view(s), lane);
            int rew, won;
            if (two) {
                if (B) { // HexEnv.step, HexSingleGame.py:233-263
                    const int mover = s.phi;
                    const int rb = E_::envb_step(s, act, lane);
                    rew = mover == 0 ? rb : -rb;
                    s.ep_ret += rb;
                    won = s.sim_done && s.winner == 0;
                } else { // HexGame.make_move with hex-v0's convention for an illegal ply
                    const int mover = s.cur_e;
                    const int w = E_::make_move(s, act, lane);
                    s.winner = w;
                    if (w == 3) s.sim_done = 1;
                    rew = w == mover ? 1 : (w == 3 ? -100 : 0);
                    s.ep_ret += w == 0 ? 1 : (w == 1 ? -1 : 0);
                    won = s.sim_done && w == 0;
                }
            } else if (B) { // SelfPlayEnv.step, SelfplayWrapper.py:174-199
                int rb = E_::envb_step(s, act, lane);
                if (!s.sim_done) { // :187-190 continue_game
                    const int oa = (MODE == M_STEP && a.opp_actions) ? rfl(a.opp_actions[o])
                                                                     : E_::rng_action(s, E_::stored_view(s), lane);
                    rb = E_::envb_step(s, oa, lane);
                }
                rew = s.agent == 0 ? rb : -rb; // :192
                s.ep_ret += rew;
                won = s.sim_done && rew > 0;
            } else { // HexEnv.step of hex-v0, HexGame.py:244-295, agent BLACK
                if (!s.sim_done) {
                    s.winner = E_::make_move(s, act, lane);
                    if (s.winner == 3) s.sim_done = 1;
                }
                if (!s.sim_done) { // opponent_move :332-349 (its view is the transposed frame)
                    const int oa = (MODE == M_STEP && a.opp_actions) ? rfl(a.opp_actions[o])
                                                                     : E_::rng_action(s, s.FT, lane);
                    const int oy = oa / N, ox = oa - oy * N;
                    s.winner = E_::make_move(s, ox * N + oy, lane);
                }
                rew = s.winner == 0 ? 1 : (s.winner == 1 ? -1 : (s.winner == 3 ? -100 : 0));
                s.ep_ret += rew;
                won = s.sim_done && rew > 0;
            }
            s.ep_len += 1;
            const int dn = s.sim_done;
            if (lane == 0) {
                if (a.reward) a.reward[o] = (float)rew;
                if (a.done) a.done[o] = (unsigned char)dn;
                if (a.winner) a.winner[o] = (signed char)s.winner;
                if (a.actions_out) a.actions_out[o] = act;
            }
            if (dn) {
                sr += s.ep_ret;
                sl += s.ep_len;
                se += 1;
                sw += won;
                if (a.tobs) E_::emit(s, a.tobs + o * G::C, nullptr, lane);
                if (auto_reset) E_::reset_env(s, a.flags, nullptr, env, lane);
            }
            if (MODE != M_ROLLOUT2)
                E_::emit(s, a.obs ? a.obs + o * G::C : nullptr, a.legal ? a.legal + o * G::C : nullptr, lane);
        }
    }

    // write the state back
    pl[lane] = s.L0;
    pl[64 + lane] = s.L1;
    pl[128 + lane] = s.F | (s.FT << 4);
    const u32 fl = (u32)s.cur_e | ((u32)s.phi << 1) | ((u32)s.sim_done << 2) | ((u32)s.agent << 3) |
                   ((u32)(s.winner + 1) << 4);
    const u32 cn = (u32)(s.n_empty & 0xff) | ((u32)s.nx0 << 8) | ((u32)s.nx1 << 16);
    const u32 hwv = lane == 0 ? s.ctr : lane == 1 ? (u32)s.ep_ret : lane == 2 ? (u32)s.ep_len
                  : lane == 3 ? fl : lane == 4 ? cn : 0u;
    if (lane < HOT_WORDS) hot[lane] = hwv;
    if (se && lane < STAT_WORDS) {
        long long *st = a.stats + (size_t)env * STAT_WORDS;
        const int add = lane == 0 ? sr : lane == 1 ? sl : lane == 2 ? se : sw;
        st[lane] += (long long)add;
    }
    // completion ticket (hexvec_set_ticket): every output of this board is out -- also those in host-pinned
    // memory across the bus -- before its wavefront adds 1; the host spins on the word instead of a stream wait
    if (a.ticket) {
        __threadfence_system();
        if (lane == 0) __hip_atomic_fetch_add(a.ticket, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}

// ---- cold kernels: state in the reference's own layout, statistics ----
__global__ void k_get_state(const u32 *planes, const u32 *hot, int E, int N, int enc,
                            signed char *board, unsigned char *labels, unsigned char *cur)
{
    const int C = N * N, W = N + 2, P = W * W;
    const int per = C + 2 * P + 1;
    const long long gid = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= (long long)E * per) return;
    const int env = (int)(gid / per), r = (int)(gid - (long long)env * per);
    const u32 *pl = planes + (size_t)env * PLANE_WORDS;
    const u32 fl = hot[(size_t)env * HOT_WORDS + 3];
    if (r < C) {
        if (!board) return;
        const int phi = (fl >> 1) & 1;
        const bool tr = enc == 1 && phi;
        const u32 w = pl[128 + (r & 63)] >> ((r >> 6) * 8);
        const u32 code = tr ? ((w >> 4) & 0xf) : (w & 0xf);
        int v;
        if (enc == 0) v = code == 0 ? 2 : (int)code - 1;
        else if (!tr) v = code == 1 ? -1 : (code == 2 ? 1 : 0);
        else v = code == 1 ? 1 : (code == 2 ? -1 : 0);
        board[(size_t)env * C + r] = (signed char)v;
    } else if (r < C + 2 * P) {
        if (!labels) return;
        const int k = r - C, p = k / P, q = k - p * P;
        labels[((size_t)env * 2 + p) * P + q] = (unsigned char)((pl[p * 64 + (q & 63)] >> ((q >> 6) * 8)) & 0xff);
    } else if (cur) {
        cur[env] = (unsigned char)(fl & 1);
    }
}

__global__ void k_stats(const long long *stats, int E, double *out4)
{
    __shared__ double sh[4][256];
    double acc[4] = {0, 0, 0, 0};
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < E; i += gridDim.x * blockDim.x)
        for (int k = 0; k < 4; ++k) acc[k] += (double)stats[(size_t)i * STAT_WORDS + k];
    for (int k = 0; k < 4; ++k) sh[k][threadIdx.x] = acc[k];
    __syncthreads();
    for (int st = 128; st > 0; st >>= 1) {
        if ((int)threadIdx.x < st)
            for (int k = 0; k < 4; ++k) sh[k][threadIdx.x] += sh[k][threadIdx.x + st];
        __syncthreads();
    }
    if (threadIdx.x < 4) atomicAdd(&out4[threadIdx.x], sh[threadIdx.x][0]); // sums of integers: exact in any order
}

__global__ void k_agents(const u32 *hot, int E, unsigned char *out)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < E) out[i] = (unsigned char)((hot[(size_t)i * HOT_WORDS + 3] >> 3) & 1);
}



template <int N, int ENC, int MODE> static hipError_t launch1(const KArgs &a, hipStream_t st)
{
    const int blocks = (a.E + WPB - 1) / WPB;
    hipLaunchKernelGGL((k_hex<N, ENC, MODE>), dim3(blocks), dim3(64 * WPB), 0, st, a);
    return hipGetLastError();
}
template <int N, int ENC> static hipError_t launch_mode(int mode, const KArgs &a, hipStream_t st)
{
    switch (mode) {
    case M_APPLY: return launch1<N, ENC, M_APPLY>(a, st);
    case M_ROLLOUT2: return launch1<N, ENC, M_ROLLOUT2>(a, st);
    case M_STEP: return launch1<N, ENC, M_STEP>(a, st);
    case M_ROLLOUT1: return launch1<N, ENC, M_ROLLOUT1>(a, st);
    case M_SETSTATE: return launch1<N, ENC, M_SETSTATE>(a, st);
    default: return launch1<N, ENC, M_RESET>(a, st);
    }
}
template <int N> static hipError_t launch_enc(int enc, int mode, const KArgs &a, hipStream_t st)
{
    return enc ? launch_mode<N, 1>(mode, a, st) : launch_mode<N, 0>(mode, a, st);
}
hipError_t wpb_launch(int N, int enc, int mode, const KArgs &a, hipStream_t st)
{
    switch (N) {
#define HV_CASE(n) case n: return launch_enc<n>(enc, mode, a, st);
        HV_CASE(2) HV_CASE(3) HV_CASE(4) HV_CASE(5) HV_CASE(6) HV_CASE(7) HV_CASE(8)
        HV_CASE(9) HV_CASE(10) HV_CASE(11) HV_CASE(12) HV_CASE(13) HV_CASE(14)
#undef HV_CASE
    default: return hipErrorInvalidValue;
    }
}


hipError_t wpb_get_state(const u32 *planes, const u32 *hot, int E, int N, int enc, signed char *board,
                         unsigned char *labels, unsigned char *cur, hipStream_t st)
{
    const int C = N * N, P = (N + 2) * (N + 2);
    const long long total = (long long)E * (C + 2 * P + 1);
    hipLaunchKernelGGL(k_get_state, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, planes, hot, E,
                       N, enc, board, labels, cur);
    return hipGetLastError();
}
hipError_t wpb_stats(const long long *stats, int E, double *out4, hipStream_t st)
{
    // out4 was zeroed on the same stream (hexvec_episode_stats)
    const int grid = (E + 255) / 256 < 1024 ? (E + 255) / 256 : 1024;
    hipLaunchKernelGGL(k_stats, dim3(grid), dim3(256), 0, st, stats, E, out4);
    return hipGetLastError();
}
hipError_t wpb_agents(const u32 *hot, int E, unsigned char *out, hipStream_t st)
{
    hipLaunchKernelGGL(k_agents, dim3((E + 255) / 256), dim3(256), 0, st, hot, E, out);
    return hipGetLastError();
}

} // namespace hv
