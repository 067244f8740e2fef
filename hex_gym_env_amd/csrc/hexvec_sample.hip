// hexvec_sample.hip -- one categorical sample per row from softmax(logits) restricted to legal != 0:
// what OpponentPolicy.choose_action obtains from model.predict(board, action_masks=mask)
// (SelfplayWrapper.py:26-35; SB3's masked categorical) and, with logits == NULL, the uniform choice of
// BaseRandomPolicy.choose_action (SelfplayWrapper.py:16-24) -- for every board of a batch in ONE launch
// (the torch formulation is eight small kernels per policy call: rand, 2 x log, clamp, add, where, max,
// counting; three policy calls per self-play collection step).
//
// Gumbel-max: argmax_c (logit_c + G_c), G_c = -log(-log(u_c)), u_c uniform in (0,1) from a counter hash
// of (seed, draw counter, row, c) -- the same distribution as sampling softmax(logits) over the legal cells.
// One 16-lane DPP row of a wavefront owns one logits row: lane s takes cells s, s+16, ...; value and index
// are reduced with DPP row shuffles.  A row that cannot be sampled (no legal cell, NaN logit on a legal
// cell, all legal logits -inf) yields action 0 and is counted in *bad_rows.  The draw counter lives on the
// device and is advanced by a one-thread kernel behind the sampling kernel, so a captured graph draws fresh
// numbers on every replay without any host involvement.
#include "hexvec_common.h"

namespace hv {

__device__ __forceinline__ float dpp_f(float x, int sel)
{
    // lane i reads lane i ^ sel of its 16-lane row (sel = 1, 2: quad permutes; 4, 8: row rotations)
    int v = __float_as_int(x);
    switch (sel) {
    case 1: v = __builtin_amdgcn_update_dpp(0, v, 0xB1, 0xF, 0xF, true); break;  // quad_perm [1,0,3,2]
    case 2: v = __builtin_amdgcn_update_dpp(0, v, 0x4E, 0xF, 0xF, true); break;  // quad_perm [2,3,0,1]
    case 4: v = __builtin_amdgcn_update_dpp(0, v, 0x141, 0xF, 0xF, true); break; // row_half_mirror
    default: v = __builtin_amdgcn_update_dpp(0, v, 0x140, 0xF, 0xF, true); break; // row_mirror
    }
    return __int_as_float(v);
}
__device__ __forceinline__ int dpp_i(int x, int sel) { return __float_as_int(dpp_f(__int_as_float(x), sel)); }

// state[0] = draw counter (state[1] unused)
__global__ __launch_bounds__(256) void k_sample_masked(const float *logits, long long logits_row_stride,
                                                       const unsigned char *legal, const unsigned char *row_active,
                                                       int rows, int cols, u32 seed_lo, u32 seed_hi, u32 salt,
                                                       const unsigned long long *state, int *actions,
                                                       unsigned long long *bad_rows)
{
    const int sub = (int)threadIdx.x & 15;
    // consecutive row ranges per XCD (workgroups go round-robin over the 8 XCDs): the board ranges the env kernels give
    // an XCD, so the masks an env launch wrote and the actions the next one reads stay in that XCD's L2
    const int nwg = (int)gridDim.x, wg = (int)blockIdx.x;
    const int wgr = (nwg & 7) == 0 ? (wg & 7) * (nwg >> 3) + (wg >> 3) : wg;
    const int row = (int)((wgr * blockDim.x + threadIdx.x) >> 4);
    if (row >= rows) return; // a 16-lane row is all in or all out
    if (row_active && row_active[row] == 0) { // nobody asked for this board's action
        if (sub == 0) actions[row] = 0;
        return;
    }
    const unsigned long long ctr = state[0];
    const size_t base = (size_t)row * (size_t)cols;
    const size_t lbase = (size_t)row * (size_t)logits_row_stride; // stride 0: every board shares one logits row
    const u32 rkey = row_key((u32)row, ctr, seed_lo, seed_hi, salt);
    float best = -INFINITY;
    int besti = 0x7fffffff;
    bool bad = false;
    for (int c0 = sub; c0 < cols; c0 += 64) { // four cells per lane and trip, their loads issued together
        unsigned char lg[4];
        float z[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int c = c0 + 16 * k;
            lg[k] = c < cols ? legal[base + c] : (unsigned char)0;
            z[k] = (logits && c < cols) ? logits[lbase + c] : 0.0f;
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int c = c0 + 16 * k;
            const u32 h = mix32(rkey + (u32)c * 0x9E3779B9U);
            const float v = z[k] + gumbel_from_hash(h);
            const bool on = lg[k] != 0;
            bad = bad || (on && z[k] != z[k]);
            if (on && v > best) {
                best = v;
                besti = c;
            }
        }
    }
    // (value, index) maximum over the 16 lanes of the row; ties go to the lower index
#pragma unroll
    for (int sel = 1; sel <= 8; sel <<= 1) {
        const float ob = dpp_f(best, sel);
        const int oi = dpp_i(besti, sel);
        const int obad = dpp_i(bad ? 1 : 0, sel);
        const bool take = ob > best || (ob == best && oi < besti);
        best = take ? ob : best;
        besti = take ? oi : besti;
        bad = bad || obad != 0;
    }
    if (sub == 0) {
        const bool degenerate = bad || !(best > -INFINITY);
        actions[row] = degenerate ? 0 : besti;
        if (degenerate && bad_rows) atomicAdd(bad_rows, 1ULL);
    }
}
// advances the draw counter once every sampling workgroup is done (a kernel of its own: the stream
// orders it behind them, also inside a captured graph; a ticket per workgroup would serialise
// thousands of atomics on one word)
__global__ void k_sample_bump(unsigned long long *state) { state[0] += 1ULL; }

// SelfPlayEnv.setup_opponents for the boards that start a new episode (see hexvec_pick_opponents)
__global__ __launch_bounds__(256) void k_pick_opponents(long long *slot, int when, const unsigned char *done,
                                                        const unsigned char *pending, unsigned char *ended,
                                                        const unsigned char *white, int rows, float prob_best,
                                                        int pool_size, u32 seed_lo, u32 seed_hi, u32 salt,
                                                        const unsigned long long *state)
{
    const int nwg = (int)gridDim.x, wg = (int)blockIdx.x; // (consecutive board ranges per XCD, as in k_sample_masked)
    const int e = (int)(((nwg & 7) == 0 ? (wg & 7) * (nwg >> 3) + (wg >> 3) : wg) * blockDim.x + threadIdx.x);
    if (e >= rows) return;
    // when 3 = points 0 and 1 of the step in one launch, 4 = points 1 and 2: a board qualifies at one point at most
    // (0 needs pending == 0, 1 pending == 2; 2 a BLACK agent, 1 after the replies a WHITE one); the second point
    // draws under salt + 1, exactly as its own launch would
    bool sel = false;
    if (when == 0 || when == 3) {
        const bool d = done[e] != 0;
        ended[e] = d ? 1 : 0;
        sel = d && pending[e] == 0;
    }
    if (!sel && (when == 1 || when == 3 || when == 4)) {
        sel = pending[e] == 2;
        if (sel && when == 3) salt += 1u;
    }
    if (!sel && (when == 2 || when == 4)) {
        sel = done[e] != 0 && ended[e] == 0 && white[e] == 0;
        if (sel && when == 4) salt += 1u;
    }
    if (!sel) return;
    slot[e] = pick_slot((u32)e, state[0], seed_lo, seed_hi, salt, prob_best, pool_size);
}

} // namespace hv


extern "C" int hexvec_sample_masked(const float *logits, long long logits_row_stride, const uint8_t *legal,
                                    const uint8_t *row_active, int rows, int cols, uint64_t seed, uint32_t salt,
                                    int advance, unsigned long long *state2, int32_t *actions,
                                    unsigned long long *bad_rows, void *stream)
{
    if (!legal || !actions || !state2 || rows < 0 || cols < 1) {
        hv::set_error("hexvec_sample_masked", "legal / actions / state NULL or bad shape");
        return HEXVEC_E_ARG;
    }
    if (rows == 0) return HEXVEC_OK;
    hv::PointerDeviceGuard guard(actions);
    if (!guard.ok) {
        hv::set_error("hexvec_sample_masked", "actions is not memory of a visible GPU");
        return HEXVEC_E_ARG;
    }
    const long long threads = (long long)rows * 16;
    const unsigned blocks = (unsigned)((threads + 255) / 256);
    hipLaunchKernelGGL(hv::k_sample_masked, dim3(blocks), dim3(256), 0, (hipStream_t)stream, logits, logits_row_stride,
                       legal, row_active, rows, cols, (hv::u32)seed, (hv::u32)(seed >> 32), salt, state2, actions, bad_rows);
    if (advance) hipLaunchKernelGGL(hv::k_sample_bump, dim3(1), dim3(1), 0, (hipStream_t)stream, state2);
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        hv::set_error("hexvec_sample_masked", hipGetErrorString(e));
        return HEXVEC_E_HIP;
    }
    return HEXVEC_OK;
}

extern "C" int hexvec_pick_opponents(int64_t *slot, int when, const uint8_t *done, const uint8_t *pending,
                                     uint8_t *ended, const uint8_t *white, int rows, float prob_best, int pool_size,
                                     uint64_t seed, uint32_t salt, int advance, unsigned long long *state2,
                                     void *stream)
{
    const bool ok = slot && state2 && rows >= 0 && pool_size >= 1 && when >= 0 && when <= 4 &&
                    (when == 0 || when == 3 ? (done && pending && ended)
                     : when == 1            ? pending != nullptr
                     : when == 2            ? (done && ended && white)
                                            : (done && pending && ended && white));
    if (!ok) {
        hv::set_error("hexvec_pick_opponents", "NULL argument or bad `when` / shape");
        return HEXVEC_E_ARG;
    }
    if (rows == 0) return HEXVEC_OK;
    hv::PointerDeviceGuard guard(slot);
    if (!guard.ok) {
        hv::set_error("hexvec_pick_opponents", "slot is not memory of a visible GPU");
        return HEXVEC_E_ARG;
    }
    hipLaunchKernelGGL(hv::k_pick_opponents, dim3((unsigned)((rows + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       (long long *)slot, when, done, pending, ended, white, rows, prob_best, pool_size, (hv::u32)seed,
                       (hv::u32)(seed >> 32), salt, state2);
    if (advance) hipLaunchKernelGGL(hv::k_sample_bump, dim3(1), dim3(1), 0, (hipStream_t)stream, state2);
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        hv::set_error("hexvec_pick_opponents", hipGetErrorString(e));
        return HEXVEC_E_HIP;
    }
    return HEXVEC_OK;
}
