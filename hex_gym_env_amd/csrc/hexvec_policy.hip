// hexvec_policy.hip -- the actor network of the self-play opponents (and of the agent) evaluated, masked and
// sampled for a whole batch of boards in ONE launch: what OpponentPolicy.choose_action obtains from
// model.predict(board, deterministic, action_masks=mask) (SelfplayWrapper.py:26-35) with SB3's default
// MlpPolicy actor: flatten -> float -> Linear-Tanh-Linear-Tanh -> Linear -> logits -> masked categorical.
// Every board may name its own network (its opponent slot, SelfplayWrapper.py:97-104): the boards are first
// grouped by network with a counting sort (three small kernels) so that a tile of boards shares one
// network; the torch spelling evaluates every distinct pool member on the whole batch.
//
// fp32 throughout: the logits must match the fp32 reference closely enough to give the same samples.
//
// k_mlp64_mfma (hidden widths 64-64, SB3's default = what the reference's scripts train): one wavefront =
// 32 boards, the three layers as chains of v_mfma_f32_32x32x2_f32 with the NEURONS on the M side and the
// BOARDS on the N side: D[neuron][board] leaves every lane holding its own board's column, 16 neurons in
// 16 registers -- which is exactly the B operand (k = neuron, n = board) of the next layer's MFMAs if that
// layer's weights are stored in the matching k order.  So the activations never leave the registers: no
// LDS, no transposes between the layers; tanh and the Gumbel-max epilogue run on the accumulators in place.
// The bias rides as one more k step against a constant-one input.  Weights stream from L2 as one coalesced
// 256-byte read per MFMA (hexvec_mlp_pack stores them in that order).
//
// k_mlp_actions (any widths that are multiples of 16): a workgroup = 64 boards x 4 wavefronts; lane =
// board; the wavefronts split the OUTPUT neurons of each layer in chunks of 16; weights are wave-uniform
// and come through the scalar cache (s_load_dwordx16 of a transposed row) into v_pk_fma_f32; activations
// pass between the layers through LDS as [neuron][lane].
//
// Sampling is the Gumbel-max of hexvec_sample.hip with the same counter hash: the fused call returns the
// actions hexvec_sample_masked returns on this kernel's logits.
#include "hexvec_common.h"

namespace hv {

constexpr int POL_WAVES = 4;    // wavefronts per workgroup of k_mlp_actions
constexpr int GRP_BLOCKS = 64;  // workgroups of the grouping kernels
constexpr int GRP_THREADS = 256;
constexpr int MAX_POLICIES = 256;

struct MlpShape {
    int cells, cp, h1, h2;
    long long o_w1, o_b1, o_w2, o_b2, o_w3, o_b3, stride; // offsets in floats, all multiples of 16
};
__host__ __device__ inline MlpShape mlp_shape(int cells, int h1, int h2)
{
    MlpShape s;
    s.cells = cells;
    s.cp = (cells + 15) & ~15;
    s.h1 = h1;
    s.h2 = h2;
    s.o_w1 = 0;
    s.o_b1 = s.o_w1 + (long long)cells * h1;
    s.o_w2 = s.o_b1 + h1;
    s.o_b2 = s.o_w2 + (long long)h1 * h2;
    s.o_w3 = s.o_b2 + h2;
    s.o_b3 = s.o_w3 + (long long)h2 * s.cp;
    s.stride = s.o_b3 + s.cp;
    return s;
}

// the MFMA layout (h1 == h2 == 64): per MFMA one 64-float record, lane l's element = the A operand of lane l.
//   layer 1: T1 = ceil((cells + 1) / 2) steps x 2 neuron tiles; step t pairs the inputs (t, T1 + t) -- input
//            `cells` is the constant one (its weights = the bias), inputs beyond it have zero weights;
//   layer 2: 33 steps x 2 tiles; step s < 32 pairs the neurons that one accumulator register of layer 1 holds in
//            the two half-waves (mfma_in()); step 32 is the bias against {1, 0};
//   layer 3: MT3 = ceil(cells / 32) rounded up to even output tiles x 33 steps.
struct MfmaShape {
    int cells, T1, MT3;
    long long o_w2, o_w3, stride;
};
__host__ __device__ inline MfmaShape mfma_shape(int cells)
{
    MfmaShape s;
    s.cells = cells;
    s.T1 = (cells + 2) / 2;
    s.MT3 = ((cells + 31) / 32 + 1) & ~1;
    s.o_w2 = 64LL * 2 * s.T1;
    s.o_w3 = s.o_w2 + 64LL * 2 * 33;
    s.stride = s.o_w3 + 64LL * 33 * s.MT3;
    return s;
}
__host__ __device__ inline bool mlp_on_mfma(int h1, int h2) { return h1 == 64 && h2 == 64; }
// neuron (row of the 32x32 tile) that accumulator register v holds in half-wave `half`
__host__ __device__ inline int mfma_row(int v, int half) { return 8 * (v >> 2) + 4 * half + (v & 3); }
// input neuron of k step s (< 32) in half-wave `half`: register s % 16 of neuron tile s / 16 of the layer before
__host__ __device__ inline int mfma_in(int s, int half) { return (s >> 4) * 32 + mfma_row(s & 15, half); }

struct MlpArgs {
    const signed char *obs;
    const unsigned char *legal;
    const unsigned char *row_active;
    const long long *slot;
    const int *policy_of_slot;
    int slot_bias, n_slots, n_policies;
    int rows;
    MlpShape sh;
    MfmaShape ms;
    const float *weights;
    int deterministic;
    u32 seed_lo, seed_hi, salt;
    const unsigned long long *state;
    int *actions;
    float *logits_out;
    unsigned long long *bad_rows;
    const int *order;  // grouped launches: board of every position (-1 = hole), else NULL
    const int *ntiles; // grouped launches: tiles in use
};

// tanh(x) = 1 - 2 / (exp(2x) + 1): |error| < 2e-7 absolute over the whole range (saturates cleanly at +-1)
__device__ __forceinline__ float tanh_fast(float x)
{
    const float t = __expf(2.0f * x);
    return 1.0f - 2.0f * __builtin_amdgcn_rcpf(t + 1.0f);
}

// the network of a board: its slot (+ bias) looked up in the pool's slot -> network table
__device__ __forceinline__ int policy_key(const MlpArgs &a, int r)
{
    if (!a.slot) return 0;
    long long s = a.slot[r] + (long long)a.slot_bias;
    s = s < 0 ? 0 : (s >= a.n_slots ? a.n_slots - 1 : s);
    int p = a.policy_of_slot ? a.policy_of_slot[s] : (int)s;
    return p < 0 ? 0 : (p >= a.n_policies ? a.n_policies - 1 : p);
}

// acc[k] = bias[q*16+k] + sum_i wt[i][q*16+k] * x_i for this lane's board; X(i) fetches x_i
template <typename X>
__device__ __forceinline__ void dense16(const float *__restrict__ wt, const float *__restrict__ bias, int K, int OP, int q,
                                        X x_of, float (&acc)[16])
{
    const float *__restrict__ w = wt + q * 16;
#pragma unroll
    for (int k = 0; k < 16; ++k) acc[k] = bias[q * 16 + k];
#pragma unroll 4
    for (int i = 0; i < K; ++i) {
        const float x = x_of(i);
        const float *__restrict__ wr = w + (size_t)i * (size_t)OP;
#pragma unroll
        for (int k = 0; k < 16; ++k) acc[k] = __builtin_fmaf(wr[k], x, acc[k]);
    }
}

__global__ __launch_bounds__(64 * POL_WAVES) void k_mlp_actions(MlpArgs a)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int lane = (int)threadIdx.x & 63;
    const int wave = rfl((int)threadIdx.x >> 6);
    const int tile = (int)blockIdx.x;
    if (a.ntiles && tile >= *a.ntiles) return;
    const int cells = a.sh.cells, cp = a.sh.cp, h1 = a.sh.h1, h2 = a.sh.h2;

    const int pos = tile * 64 + lane;
    int r = a.order ? a.order[pos] : (pos < a.rows ? pos : -1);
    if (!a.order && r >= 0 && a.row_active && a.row_active[r] == 0) { // nobody asked for this board's action
        if (wave == 0) a.actions[r] = 0;
        r = -1;
    }
    const u64 vmask = __ballot(r >= 0); // the same in every wavefront of the workgroup
    if (vmask == 0) return;
    const int r0 = __builtin_amdgcn_readlane(r, (int)__ffsll((long long)vmask) - 1);
    // grouped launches: one network per tile; with ONE network nothing about the weights waits for the loads above
    const int p = (a.slot && a.n_policies > 1) ? rfl(policy_key(a, r0)) : 0;
    const float *__restrict__ wp = a.weights + (size_t)p * (size_t)a.sh.stride;

    // LDS: obs bytes [64][cells] | legal bytes [64][cells] | hA [h1][64] | hB [h2][64] | reduction [W][64] x 3 | rows [64]
    const int img = (64 * cells + 15) & ~15;
    signed char *obs_b = reinterpret_cast<signed char *>(smem);
    unsigned char *legal_b = smem + img;
    float *hA = reinterpret_cast<float *>(smem + 2 * img);
    float *hB = hA + h1 * 64;
    float *red_v = hB + h2 * 64;
    int *red_i = reinterpret_cast<int *>(red_v + POL_WAVES * 64);
    int *red_b = red_i + POL_WAVES * 64;
    int *rows_l = red_b + POL_WAVES * 64;

    if (wave == 0) rows_l[lane] = r;
    __syncthreads();
    // the tile's observation and mask rows: wavefront w copies boards w, w+4, ... (a row is contiguous)
    for (int b = wave; b < 64; b += POL_WAVES) {
        const int src = rows_l[b];
        for (int c = lane; c < cells; c += 64) {
            obs_b[b * cells + c] = src >= 0 ? a.obs[(size_t)src * cells + c] : (signed char)0;
            legal_b[b * cells + c] = src >= 0 ? a.legal[(size_t)src * cells + c] : (unsigned char)0;
        }
    }
    __syncthreads();

    float acc[16];
    // layer 1: cells -> h1, tanh
    for (int q = wave; q < h1 / 16; q += POL_WAVES) {
        dense16(wp + a.sh.o_w1, wp + a.sh.o_b1, cells, h1, q,
                [&](int i) { return (float)(int)obs_b[lane * cells + i]; }, acc);
#pragma unroll
        for (int k = 0; k < 16; ++k) hA[(q * 16 + k) * 64 + lane] = tanh_fast(acc[k]);
    }
    __syncthreads();
    // layer 2: h1 -> h2, tanh
    for (int q = wave; q < h2 / 16; q += POL_WAVES) {
        dense16(wp + a.sh.o_w2, wp + a.sh.o_b2, h1, h2, q, [&](int i) { return hA[i * 64 + lane]; }, acc);
#pragma unroll
        for (int k = 0; k < 16; ++k) hB[(q * 16 + k) * 64 + lane] = tanh_fast(acc[k]);
    }
    __syncthreads();
    // layer 3: h2 -> logits; masked Gumbel-max (or argmax) over this wavefront's cells on the fly
    const unsigned long long ctr = a.state ? a.state[0] : 0ULL;
    const u32 c_lo = (u32)ctr, c_hi = (u32)(ctr >> 32);
    const u32 rkey = mix32(((u32)r + c_lo * 0x85EBCA6BU) * 0x9E3779B9U + a.seed_lo) ^ a.seed_hi ^ (c_hi * 0xC2B2AE35U) ^
                     mix32(a.salt * 0x27D4EB2FU + 0x165667B1U); // = k_sample_masked's row key
    float best = -INFINITY;
    int besti = 0x7fffffff;
    bool bad = false;
    for (int q = wave; q < cp / 16; q += POL_WAVES) {
        dense16(wp + a.sh.o_w3, wp + a.sh.o_b3, h2, cp, q, [&](int i) { return hB[i * 64 + lane]; }, acc);
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            const int c = q * 16 + k;
            if (c < cells) {
                const float z = acc[k];
                if (a.logits_out && r >= 0) a.logits_out[(size_t)r * cells + c] = z;
                const float v = a.deterministic ? z : z + gumbel_from_hash(mix32(rkey + (u32)c * 0x9E3779B9U));
                const bool on = legal_b[lane * cells + c] != 0;
                bad = bad || (on && z != z);
                if (on && v > best) {
                    best = v;
                    besti = c;
                }
            }
        }
    }
    red_v[wave * 64 + lane] = best;
    red_i[wave * 64 + lane] = besti;
    red_b[wave * 64 + lane] = bad ? 1 : 0;
    __syncthreads();
    if (wave == 0) {
        best = red_v[lane];
        besti = red_i[lane];
        bad = red_b[lane] != 0;
#pragma unroll
        for (int w = 1; w < POL_WAVES; ++w) {
            const float ob = red_v[w * 64 + lane];
            const int oi = red_i[w * 64 + lane];
            const bool take = ob > best || (ob == best && oi < besti); // ties go to the lower cell
            best = take ? ob : best;
            besti = take ? oi : besti;
            bad = bad || red_b[w * 64 + lane] != 0;
        }
        const bool degenerate = r >= 0 && (bad || !(best > -INFINITY));
        if (r >= 0) a.actions[r] = degenerate ? 0 : besti;
        const u64 dm = __ballot(degenerate);
        if (dm && a.bad_rows && lane == 0) atomicAdd(a.bad_rows, (unsigned long long)__popcll(dm));
    }
}

// ---- hidden widths 64-64 on the matrix cores ----
typedef float f32x16 __attribute__((ext_vector_type(16)));

struct Pick {
    float best;
    int besti;
    bool bad;
};
// one cell of the masked Gumbel-max (the arithmetic of k_sample_masked)
__device__ __forceinline__ void consider(Pick &p, float z, int c, bool on, int deterministic, u32 rkey)
{
    const float v = deterministic ? z : z + gumbel_from_hash(mix32(rkey + (u32)c * 0x9E3779B9U));
    p.bad = p.bad || (on && z != z);
    if (on && v > p.best) {
        p.best = v;
        p.besti = c;
    }
}

// CNT MFMA operand records (64 floats each; `p` already points at this lane's element)
template <int CNT> __device__ __forceinline__ void wload(float (&w)[CNT], const float *__restrict__ p)
{
#pragma unroll
    for (int i = 0; i < CNT; ++i) w[i] = p[i * 64];
}

// Diagnostic build (-DHV_POL_STAMPS, tools/stamp_policy.py): s_memtime at the phase boundaries of every wavefront
#ifdef HV_POL_STAMPS
__device__ unsigned long long g_pol_stamps[8192][10];
#define POL_MARK(i)                                                                        \
    do {                                                                                   \
        __builtin_amdgcn_sched_barrier(0);                                                 \
        if (lane == 0 && tile < 8192) g_pol_stamps[tile][i] = __builtin_amdgcn_s_memtime(); \
        __builtin_amdgcn_sched_barrier(0);                                                 \
    } while (0)
// (k_mlp64_mfma2: row 2 * tile + wavefront)
#define POL_MARK2(i)                                                                                       \
    do {                                                                                                   \
        __builtin_amdgcn_sched_barrier(0);                                                                 \
        if (lane == 0 && 2 * tile + w < 8192) g_pol_stamps[2 * tile + w][i] = __builtin_amdgcn_s_memtime(); \
        __builtin_amdgcn_sched_barrier(0);                                                                 \
    } while (0)
#else
#define POL_MARK(i) do { } while (0)
#define POL_MARK2(i) do { } while (0)
#endif

constexpr int MF_WAVES = 4; // wavefronts (= tiles of 32 boards) per workgroup; they share nothing

__global__ __launch_bounds__(64 * MF_WAVES) __attribute__((amdgpu_waves_per_eu(2))) void k_mlp64_mfma(MlpArgs a)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int lane = (int)threadIdx.x & 63;
    const int wave = rfl((int)threadIdx.x >> 6);
    const int b = lane & 31, half = lane >> 5; // board of the tile; which input of a k pair / which rows of D
    // (ungrouped launches: consecutive tiles per XCD, the board ranges the env kernels give an XCD -- see k_mlp64_mfma2)
    const int nwg = (int)gridDim.x, wg = (int)blockIdx.x;
    const int tile = ((!a.order && (nwg & 7) == 0) ? (wg & 7) * (nwg >> 3) + (wg >> 3) : wg) * MF_WAVES + wave;
    if (a.ntiles && tile >= 2 * *a.ntiles) return; // (the grouping counts tiles of 64)
    const int cells = a.ms.cells, T1 = a.ms.T1;

    POL_MARK(0);
    // Weights: one coalesced 256-byte record per MFMA, streamed from L2 into registers a whole stage AHEAD of the
    // MFMAs that use them (layer 2's records and layer 1's first chunk are requested here, before the boards'
    // rows; the first output tiles' before layer 2 runs, ...): with one wavefront per SIMD nothing else hides the
    // L2 latency.  With ONE network they are requested before anything else is known about the tile.
    const bool one_net = !(a.slot && a.n_policies > 1);
    const float *__restrict__ wl = a.weights + lane;
    float w2r[66], wa[32], wb[32];
    if (one_net) {
        wload(wa, wl);
        wload(w2r, wl + a.ms.o_w2);
    }
    const int pos = tile * 32 + b;
    int r = a.order ? a.order[pos] : (pos < a.rows ? pos : -1);
    if (!a.order && r >= 0 && a.row_active && a.row_active[r] == 0) { // nobody asked for this board's action
        if (half == 0) a.actions[r] = 0;
        r = -1;
    }
    const u64 vmask = __ballot(r >= 0);
    if (vmask == 0) return; // wavefronts are independent: no workgroup barrier anywhere in this kernel
    if (!one_net) { // grouped launches: one network per tile
        const int r0 = __builtin_amdgcn_readlane(r, (int)__ffsll((long long)vmask) - 1);
        wl += (size_t)rfl(policy_key(a, r0)) * (size_t)a.ms.stride;
        wload(wa, wl);
        wload(w2r, wl + a.ms.o_w2);
    }

    // this wavefront's observation and mask rows -> LDS [32][cells] bytes each (a row is contiguous in HBM)
    const int img = (32 * cells + 15) & ~15;
    signed char *obs_w = reinterpret_cast<signed char *>(smem) + (size_t)wave * 2 * img;
    unsigned char *legal_w = reinterpret_cast<unsigned char *>(obs_w) + img;
    const bool whole = !a.order && tile * 32 + 32 <= a.rows && (((uintptr_t)a.obs | (uintptr_t)a.legal) & 3) == 0;
    if (whole) { // 32 consecutive boards: one contiguous block of 32 * cells bytes, copied as dwords
        const u32 *so = reinterpret_cast<const u32 *>(a.obs + (size_t)tile * 32 * cells);
        const u32 *sl = reinterpret_cast<const u32 *>(a.legal + (size_t)tile * 32 * cells);
        const int nd = 8 * cells;
#pragma unroll 4
        for (int i = lane; i < nd; i += 64) {
            reinterpret_cast<u32 *>(obs_w)[i] = so[i];
            reinterpret_cast<u32 *>(legal_w)[i] = sl[i];
        }
    } else {
        for (int b0 = 0; b0 < 32; b0 += 8) { // eight rows' loads in flight together
            signed char o[8];
            unsigned char l[8];
            for (int c0 = 0; c0 < cells; c0 += 64) {
                const int c = c0 + lane;
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    const int src = __shfl(r, b0 + k);
                    const bool ok = src >= 0 && c < cells;
                    o[k] = ok ? a.obs[(size_t)src * cells + c] : (signed char)0;
                    l[k] = ok ? a.legal[(size_t)src * cells + c] : (unsigned char)0;
                }
#pragma unroll
                for (int k = 0; k < 8; ++k)
                    if (c < cells) {
                        obs_w[(b0 + k) * cells + c] = o[k];
                        legal_w[(b0 + k) * cells + c] = l[k];
                    }
            }
        }
    }
    POL_MARK(1); // rows requested and written to LDS
    // (LDS writes and the reads below are ordered inside one wavefront by s_waitcnt)


    // layer 1: [64 neurons][32 boards] = W1 [64][cells + 1] x (obs | 1) [cells + 1][32], 16 k steps per chunk
    f32x16 h0 = {0}, h1 = {0};
    {
        auto chunk = [&](const float(&w)[32], int t0) {
            float x[16]; // the chunk's 16 inputs first, their LDS reads issued together; steps past T1 (the last
#pragma unroll           // chunk's tail) multiply by zero
            for (int i = 0; i < 16; ++i) {
                const int t = t0 + i;
                const int in = half * T1 + t;
                const float o = (float)(int)obs_w[b * cells + min(in, cells - 1)];
                x[i] = t < T1 ? (in < cells ? o : (in == cells ? 1.0f : 0.0f)) : 0.0f;
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                h0 = __builtin_amdgcn_mfma_f32_32x32x2f32(w[2 * i], x[i], h0, 0, 0, 0);
                h1 = __builtin_amdgcn_mfma_f32_32x32x2f32(w[2 * i + 1], x[i], h1, 0, 0, 0);
            }
        };
        // (a chunk past T1 reads on into the layer-2 / layer-3 records of the same network: in bounds, unused)
        for (int t0 = 0; t0 < T1; t0 += 32) {
            wload(wb, wl + (size_t)(t0 + 16) * 128);
            chunk(wa, t0);
            wload(wa, wl + (size_t)(t0 + 32) * 128);
            chunk(wb, t0 + 16);
        }
    }
    POL_MARK(2); // layer 1
#pragma unroll
    for (int v = 0; v < 16; ++v) {
        h0[v] = tanh_fast(h0[v]);
        h1[v] = tanh_fast(h1[v]);
    }
    POL_MARK(3); // tanh 1
    // layer 2: the accumulators of layer 1 ARE the B operands (k = neuron, n = board)
    float w3p[66];
    const float *__restrict__ w3 = wl + a.ms.o_w3;
    wload(w3p, w3);
    const float one = half == 0 ? 1.0f : 0.0f; // the bias step's input pair {1, 0}
    f32x16 g0 = {0}, g1 = {0};
#pragma unroll
    for (int s = 0; s < 33; ++s) {
        const float x = s < 16 ? h0[s & 15] : s < 32 ? h1[s & 15] : one;
        g0 = __builtin_amdgcn_mfma_f32_32x32x2f32(w2r[2 * s], x, g0, 0, 0, 0);
        g1 = __builtin_amdgcn_mfma_f32_32x32x2f32(w2r[2 * s + 1], x, g1, 0, 0, 0);
    }
    POL_MARK(4); // layer 2
#pragma unroll
    for (int v = 0; v < 16; ++v) {
        g0[v] = tanh_fast(g0[v]);
        g1[v] = tanh_fast(g1[v]);
    }
    POL_MARK(5); // tanh 2
    // layer 3, two tiles of 32 cells at a time; masked Gumbel-max (or arg-max) on the accumulators
    const unsigned long long ctr = a.state ? a.state[0] : 0ULL;
    const u32 c_lo = (u32)ctr, c_hi = (u32)(ctr >> 32);
    const u32 rkey = mix32(((u32)r + c_lo * 0x85EBCA6BU) * 0x9E3779B9U + a.seed_lo) ^ a.seed_hi ^ (c_hi * 0xC2B2AE35U) ^
                     mix32(a.salt * 0x27D4EB2FU + 0x165667B1U); // = k_sample_masked's row key
    Pick pk = {-INFINITY, 0x7fffffff, false};
    // Per pair of output tiles: 66 MFMAs, and -- issued between them, in their shadow -- everything about this
    // lane's 32 cells that does not depend on the logits: the cell's Gumbel noise, or -inf if it is not a legal cell
    // (fp32 MFMAs run on the SIMD's fp32 lanes, so this does not overlap them; it keeps the tail after the last MFMA short).
    const float nscale = a.deterministic ? 0.0f : 1.0f; // arg-max play: no noise
    for (int mt = 0; mt < a.ms.MT3; mt += 2) { // (MT3 is even; the records of a tile pair are 66 * 64 floats)
        f32x16 z0 = {0}, z1 = {0};
        float nz[32];
        u32 lgm = 0; // bit s: this lane's cell s of the pair is on the board and legal (32 LDS byte reads, issued together)
#pragma unroll
        for (int s = 0; s < 32; ++s) {
            const int c = (mt + (s >> 4)) * 32 + mfma_row(s & 15, half);
            lgm |= (u32)((c < cells) & (legal_w[b * cells + min(c, cells - 1)] != 0)) << s;
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int s = 0; s < 33; ++s) {
            const float x = s < 16 ? g0[s & 15] : s < 32 ? g1[s & 15] : one;
            z0 = __builtin_amdgcn_mfma_f32_32x32x2f32(w3p[s], x, z0, 0, 0, 0);
            z1 = __builtin_amdgcn_mfma_f32_32x32x2f32(w3p[33 + s], x, z1, 0, 0, 0);
            if (s < 32) {
                const int c = (mt + (s >> 4)) * 32 + mfma_row(s & 15, half);
                const float g = nscale * gumbel_from_hash(mix32(rkey + (u32)c * 0x9E3779B9U));
                nz[s] = (lgm >> s) & 1u ? g : -INFINITY;
            }
        }
        if (mt + 2 < a.ms.MT3) wload(w3p, w3 + (size_t)(mt + 2) * 33 * 64); // lands while the cells are compared
        if (a.logits_out) { // (diagnostic / test output: one wave-uniform branch, then predicated stores)
#pragma unroll
            for (int s = 0; s < 32; ++s) {
                const int c = (mt + (s >> 4)) * 32 + mfma_row(s & 15, half);
                if (r >= 0 && c < cells) a.logits_out[(size_t)r * cells + c] = s < 16 ? z0[s & 15] : z1[s & 15];
            }
        }
#pragma unroll
        for (int s = 0; s < 32; ++s) { // branch-free: compares and selects only
            const int c = (mt + (s >> 4)) * 32 + mfma_row(s & 15, half);
            const float z = s < 16 ? z0[s & 15] : z1[s & 15];
            const float v = z + nz[s]; // -inf on a cell that is not legal (NaN if z is +inf there: never the maximum)
            pk.bad = pk.bad | ((z != z) & (nz[s] > -INFINITY));
            const bool take = v > pk.best;
            pk.best = take ? v : pk.best;
            pk.besti = take ? c : pk.besti;
        }
    }
    POL_MARK(6); // layer 3 + epilogues
    // the two half-waves hold complementary cells of the same board
    {
        const float ob = __shfl_xor(pk.best, 32);
        const int oi = __shfl_xor(pk.besti, 32);
        const int obad = __shfl_xor(pk.bad ? 1 : 0, 32);
        const bool take = ob > pk.best || (ob == pk.best && oi < pk.besti); // ties go to the lower cell
        pk.best = take ? ob : pk.best;
        pk.besti = take ? oi : pk.besti;
        pk.bad = pk.bad || obad != 0;
    }
    const bool mine = half == 0 && r >= 0;
    const bool degenerate = mine && (pk.bad || !(pk.best > -INFINITY));
    if (mine) a.actions[r] = degenerate ? 0 : pk.besti;
    const u64 dm = __ballot(degenerate);
    if (dm && a.bad_rows && lane == 0) atomicAdd(a.bad_rows, (unsigned long long)__popcll(dm));
    POL_MARK(7);
}

// The same network with TWO wavefronts per tile of 32 boards, for launches that would otherwise leave one lone
// wavefront per SIMD (or fewer): wavefront w owns neuron tile w of both hidden layers and the output tiles
// w, w + 2, ...; the activations cross between the two through LDS ([neuron][board] floats, one workgroup barrier
// per layer).  Per wavefront half the MFMAs, half the tanh and half the sampling epilogue; the weight records are
// the same (each wavefront reads those of its own tiles).  Bit-identical logits to k_mlp64_mfma (same chains).
__global__ __launch_bounds__(128) void k_mlp64_mfma2(MlpArgs a)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int lane = (int)threadIdx.x & 63;
    const int w = rfl((int)threadIdx.x >> 6);
    const int b = lane & 31, half = lane >> 5;
    // Ungrouped launches: consecutive tiles per XCD (workgroups go round-robin over the 8 XCDs), the same board ranges the
    // env kernels give an XCD (hexvec_qpb.h) -- what the env launch before this one wrote and what the one behind it
    // reads (observations, masks, actions) then stays in that XCD's L2.
    const int nwg = (int)gridDim.x, wg = (int)blockIdx.x;
    const int tile = (!a.order && (nwg & 7) == 0) ? (wg & 7) * (nwg >> 3) + (wg >> 3) : wg;
    if (a.ntiles && tile >= 2 * *a.ntiles) return; // (workgroup-uniform, like every exit below)
    const int cells = a.ms.cells, T1 = a.ms.T1;
    POL_MARK2(0);

    // A tile of whole, ungrouped rows: its observation and mask rows are requested FIRST (loads return in order, and the
    // rows are what the first barrier waits for: behind 65 weight records they arrive later), up to ROWK dwords of
    // each array per thread (4: boards up to 8x8; larger ones copy in a loop below -- with 11, up to 13x13, the
    // 13x13 / 16 384 launch stayed at 31.6 us: one wavefront per SIMD there, bound by its own 96-step layer 1)
#ifndef HV_POL_ROWK
#define HV_POL_ROWK 4
#endif
    constexpr int ROWK = HV_POL_ROWK;
    const bool whole = !a.order && tile * 32 + 32 <= a.rows && (((uintptr_t)a.obs | (uintptr_t)a.legal) & 3) == 0;
    const bool whole_pre = whole && 8 * cells <= 128 * ROWK;
    u32 row_o[ROWK], row_l[ROWK];
    if (whole_pre) {
        const u32 *so = reinterpret_cast<const u32 *>(a.obs + (size_t)tile * 32 * cells);
        const u32 *sl = reinterpret_cast<const u32 *>(a.legal + (size_t)tile * 32 * cells);
        const int nd = 8 * cells;
#pragma unroll
        for (int k = 0; k < ROWK; ++k)
            if (128 * k < nd) { // (workgroup-uniform)
                const int i = min((int)threadIdx.x + 128 * k, nd - 1);
                row_o[k] = so[i];
                row_l[k] = sl[i];
            }
    }
    // this wavefront's records: layer 2 (33) and the first 16 steps of layer 1, requested before the boards' rows --
    // with ONE network before anything else is known about the tile
    const bool one_net = !(a.slot && a.n_policies > 1);
    const float *__restrict__ wl = a.weights + lane;
    float w2r[33], wa[16], wb[16];
    auto wload1 = [&](float(&dst)[16], int t0) { // records (t, tile w), t = t0 .. t0 + 15
#pragma unroll
        for (int i = 0; i < 16; ++i) dst[i] = wl[(size_t)((t0 + i) * 2 + w) * 64];
    };
    auto wload2 = [&]() {
#pragma unroll
        for (int s = 0; s < 33; ++s) w2r[s] = wl[a.ms.o_w2 + (size_t)(s * 2 + w) * 64];
    };
    // (up to 32 k steps -- boards up to 7x7 -- the whole of layer 1 is requested here: no record is waited for inside it)
    const bool l1_whole = T1 <= 32;
    if (one_net) {
        wload1(wa, 0);
        if (l1_whole) wload1(wb, 16);
        wload2();
    }
    // the first of this wavefront's output tiles: with one network requested in the opening burst too (its address
    // arithmetic and issue then sit in the shadow of the rows' round trip, not between layer 1 and its barrier).
    // (Measured and dropped: all records through buffer loads with scalar offsets -- 900 address instructions less per
    // wavefront, and no faster: 47.3-47.4 us per collection step at config 3 with everything requested up front, 46.2 with
    // the later layers requested after the first barrier, against 45.6-46.3 as it stands.  The kernel waits for round
    // trips at this batch, not for its instruction stream.)
    float w3p[33];
    auto wload3 = [&]() {
        const float *__restrict__ w3 = wl + a.ms.o_w3;
#pragma unroll
        for (int s = 0; s < 33; ++s) w3p[s] = w3[(size_t)(w * 33 + s) * 64];
    };
    if (one_net) wload3();
    const int pos = tile * 32 + b;
    int r = a.order ? a.order[pos] : (pos < a.rows ? pos : -1);
    if (!a.order && r >= 0 && a.row_active && a.row_active[r] == 0) {
        if (half == 0 && w == 0) a.actions[r] = 0;
        r = -1;
    }
    const u64 vmask = __ballot(r >= 0);
    if (vmask == 0) return;
    if (!one_net) {
        const int r0 = __builtin_amdgcn_readlane(r, (int)__ffsll((long long)vmask) - 1);
        wl += (size_t)rfl(policy_key(a, r0)) * (size_t)a.ms.stride;
        wload1(wa, 0);
        if (l1_whole) wload1(wb, 16);
        wload2();
        wload3();
    }

    // LDS: obs [32][cells] | legal [32][cells] | hA [64][32] floats | hB [64][32] floats | reduction 3 x [32]
    const int img = (32 * cells + 15) & ~15;
    signed char *obs_w = reinterpret_cast<signed char *>(smem);
    unsigned char *legal_w = smem + img;
    float *hA = reinterpret_cast<float *>(smem + 2 * img);
    float *hB = hA + 64 * 32;
    float *red_v = hB + 64 * 32;
    int *red_i = reinterpret_cast<int *>(red_v + 32);
    int *red_b = red_i + 32;
    if (whole_pre) {
        const int nd = 8 * cells;
#pragma unroll
        for (int k = 0; k < ROWK; ++k) {
            const int i = (int)threadIdx.x + 128 * k;
            if (i < nd) {
                reinterpret_cast<u32 *>(obs_w)[i] = row_o[k];
                reinterpret_cast<u32 *>(legal_w)[i] = row_l[k];
            }
        }
    } else if (whole) {
        const u32 *so = reinterpret_cast<const u32 *>(a.obs + (size_t)tile * 32 * cells);
        const u32 *sl = reinterpret_cast<const u32 *>(a.legal + (size_t)tile * 32 * cells);
        const int nd = 8 * cells;
#pragma unroll 4
        for (int i = (int)threadIdx.x; i < nd; i += 128) {
            reinterpret_cast<u32 *>(obs_w)[i] = so[i];
            reinterpret_cast<u32 *>(legal_w)[i] = sl[i];
        }
    } else {
        for (int b0 = w * 16; b0 < w * 16 + 16; b0 += 8) { // wavefront w fetches boards 16 w .. 16 w + 15
            signed char o[8];
            unsigned char l[8];
            for (int c0 = 0; c0 < cells; c0 += 64) {
                const int c = c0 + lane;
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    const int src = __shfl(r, b0 + k);
                    const bool ok = src >= 0 && c < cells;
                    o[k] = ok ? a.obs[(size_t)src * cells + c] : (signed char)0;
                    l[k] = ok ? a.legal[(size_t)src * cells + c] : (unsigned char)0;
                }
#pragma unroll
                for (int k = 0; k < 8; ++k)
                    if (c < cells) {
                        obs_w[(b0 + k) * cells + c] = o[k];
                        legal_w[(b0 + k) * cells + c] = l[k];
                    }
            }
        }
    }
    __syncthreads();
    POL_MARK2(1); // rows in LDS

    // layer 1, neuron tile w
    f32x16 h = {0};
    {
        auto inputs = [&](float(&x)[16], int t0) { // the B operands of k steps t0 .. t0 + 15: two inputs of this lane's board each
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int t = t0 + i;
                const int in = half * T1 + t;
                const float o = (float)(int)obs_w[b * cells + min(in, cells - 1)];
                x[i] = t < T1 ? (in < cells ? o : (in == cells ? 1.0f : 0.0f)) : 0.0f;
            }
        };
        if (l1_whole) {
            // both halves' inputs are read before the first MFMA issues, the k steps past T1 (zero inputs) are not run
            float xa[16], xb[16];
            inputs(xa, 0);
            inputs(xb, 16);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int i = 0; i < 16; ++i)
                if (i < T1) h = __builtin_amdgcn_mfma_f32_32x32x2f32(wa[i], xa[i], h, 0, 0, 0);
#pragma unroll
            for (int i = 0; i < 16; ++i)
                if (16 + i < T1) h = __builtin_amdgcn_mfma_f32_32x32x2f32(wb[i], xb[i], h, 0, 0, 0);
        } else {
            // (the k steps past T1 -- zero inputs -- are not run: 11 of 96 at 13x13.  Requesting every chunk's records two
            // chunks ahead instead of one, three buffers in rotation, changed nothing: the records are not what layer 1
            // waits for at the large boards, its 16 byte-reads-and-converts per chunk are.)
            auto chunk_g = [&](const float(&wr)[16], int t0) {
                float x[16];
                inputs(x, t0);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int i = 0; i < 16; ++i)
                    if (t0 + i < T1) h = __builtin_amdgcn_mfma_f32_32x32x2f32(wr[i], x[i], h, 0, 0, 0);
            };
            for (int t0 = 0; t0 < T1; t0 += 32) {
                wload1(wb, t0 + 16);
                chunk_g(wa, t0);
                wload1(wa, t0 + 32);
                if (t0 + 16 < T1) chunk_g(wb, t0 + 16);
            }
        }
    }
    POL_MARK2(2); // layer 1 MFMAs
#pragma unroll
    for (int v = 0; v < 16; ++v) hA[(w * 32 + mfma_row(v, half)) * 32 + b] = tanh_fast(h[v]);
    __syncthreads();
    POL_MARK2(3); // tanh 1 + barrier
    // layer 2, neuron tile w: all 64 activations of layer 1 come back from LDS in k-step order
    const float one = half == 0 ? 1.0f : 0.0f;
    f32x16 g = {0};
    {
        float x[33];
#pragma unroll
        for (int s = 0; s < 33; ++s) x[s] = s < 32 ? hA[mfma_in(s, half) * 32 + b] : one;
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int s = 0; s < 33; ++s) g = __builtin_amdgcn_mfma_f32_32x32x2f32(w2r[s], x[s], g, 0, 0, 0);
    }
    POL_MARK2(4); // layer 2 MFMAs
#pragma unroll
    for (int v = 0; v < 16; ++v) hB[(w * 32 + mfma_row(v, half)) * 32 + b] = tanh_fast(g[v]);
    __syncthreads();
    POL_MARK2(5); // tanh 2 + barrier
    float xs[33];
#pragma unroll
    for (int s = 0; s < 33; ++s) xs[s] = s < 32 ? hB[mfma_in(s, half) * 32 + b] : one;
    __builtin_amdgcn_sched_barrier(0);
    // layer 3: output tiles w, w + 2, ...
    const unsigned long long ctr = a.state ? a.state[0] : 0ULL;
    const u32 c_lo = (u32)ctr, c_hi = (u32)(ctr >> 32);
    const u32 rkey = mix32(((u32)r + c_lo * 0x85EBCA6BU) * 0x9E3779B9U + a.seed_lo) ^ a.seed_hi ^ (c_hi * 0xC2B2AE35U) ^
                     mix32(a.salt * 0x27D4EB2FU + 0x165667B1U);
    Pick pk = {-INFINITY, 0x7fffffff, false};
    const float nscale = a.deterministic ? 0.0f : 1.0f;
    for (int mt = w; mt < a.ms.MT3; mt += 2) {
        f32x16 z = {0};
        float nz[16];
        u32 lgm = 0;
#pragma unroll
        for (int s = 0; s < 16; ++s) {
            const int c = mt * 32 + mfma_row(s, half);
            lgm |= (u32)((c < cells) & (legal_w[b * cells + min(c, cells - 1)] != 0)) << s;
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int s = 0; s < 33; ++s) {
            z = __builtin_amdgcn_mfma_f32_32x32x2f32(w3p[s], xs[s], z, 0, 0, 0);
            if (s < 16) { // this lane's 16 cells of the tile: noise (or -inf), between the MFMAs
                const int c = mt * 32 + mfma_row(s, half);
                const float gn = nscale * gumbel_from_hash(mix32(rkey + (u32)c * 0x9E3779B9U));
                nz[s] = (lgm >> s) & 1u ? gn : -INFINITY;
            }
        }
        if (mt + 2 < a.ms.MT3) {
#pragma unroll
            for (int s = 0; s < 33; ++s) w3p[s] = (wl + a.ms.o_w3)[(size_t)((mt + 2) * 33 + s) * 64];
        }
        if (a.logits_out) {
#pragma unroll
            for (int s = 0; s < 16; ++s) {
                const int c = mt * 32 + mfma_row(s, half);
                if (r >= 0 && c < cells) a.logits_out[(size_t)r * cells + c] = z[s];
            }
        }
#pragma unroll
        for (int s = 0; s < 16; ++s) {
            const int c = mt * 32 + mfma_row(s, half);
            const float v = z[s] + nz[s];
            pk.bad = pk.bad | ((z[s] != z[s]) & (nz[s] > -INFINITY));
            const bool take = v > pk.best; // (this wavefront's cells come in increasing order)
            pk.best = take ? v : pk.best;
            pk.besti = take ? c : pk.besti;
        }
    }
    {
        const float ob = __shfl_xor(pk.best, 32);
        const int oi = __shfl_xor(pk.besti, 32);
        const int obad = __shfl_xor(pk.bad ? 1 : 0, 32);
        const bool take = ob > pk.best || (ob == pk.best && oi < pk.besti);
        pk.best = take ? ob : pk.best;
        pk.besti = take ? oi : pk.besti;
        pk.bad = pk.bad || obad != 0;
    }
    POL_MARK2(6); // layer 3 + epilogue
    if (w == 1 && half == 0) {
        red_v[b] = pk.best;
        red_i[b] = pk.besti;
        red_b[b] = pk.bad ? 1 : 0;
    }
    __syncthreads();
    if (w == 0) {
        const float ob = red_v[b];
        const int oi = red_i[b];
        const bool take = ob > pk.best || (ob == pk.best && oi < pk.besti);
        pk.best = take ? ob : pk.best;
        pk.besti = take ? oi : pk.besti;
        pk.bad = pk.bad || red_b[b] != 0;
        const bool mine = half == 0 && r >= 0;
        const bool degenerate = mine && (pk.bad || !(pk.best > -INFINITY));
        if (mine) a.actions[r] = degenerate ? 0 : pk.besti;
        const u64 dm = __ballot(degenerate);
        if (dm && a.bad_rows && lane == 0) atomicAdd(a.bad_rows, (unsigned long long)__popcll(dm));
    }
    POL_MARK2(7);
}

// row-major Linear weights ([out][in], torch's layout) -> the packed form of one network
__global__ __launch_bounds__(256) void k_mlp_pack(const float *w1, const float *b1, const float *w2, const float *b2,
                                                  const float *w3, const float *b3, int cells, int h1, int h2, float *dst)
{
    const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
    float val = 0.0f;
    if (mlp_on_mfma(h1, h2)) {
        const MfmaShape m = mfma_shape(cells);
        if (idx >= m.stride) return;
        const int l = (int)(idx & 63), out32 = l & 31, hf = l >> 5;
        if (idx < m.o_w2) {
            const int t = (int)(idx >> 7), out = (int)((idx >> 6) & 1) * 32 + out32, in = hf * m.T1 + t;
            val = in < cells ? w1[(size_t)out * cells + in] : (in == cells ? b1[out] : 0.0f);
        } else if (idx < m.o_w3) {
            const long long j = idx - m.o_w2;
            const int s = (int)(j >> 7), out = (int)((j >> 6) & 1) * 32 + out32;
            val = s < 32 ? w2[(size_t)out * 64 + mfma_in(s, hf)] : (hf == 0 ? b2[out] : 0.0f);
        } else {
            const long long j = idx - m.o_w3;
            const int mt = (int)(j / (33 * 64)), s = (int)((j >> 6) % 33), out = mt * 32 + out32;
            if (out < cells) val = s < 32 ? w3[(size_t)out * 64 + mfma_in(s, hf)] : (hf == 0 ? b3[out] : 0.0f);
        }
    } else { // W1t [cells][h1] | b1 | W2t [h1][h2] | b2 | W3t [h2][cp] | b3 [cp]
        const MlpShape m = mlp_shape(cells, h1, h2);
        if (idx >= m.stride) return;
        if (idx < m.o_b1) {
            val = w1[(size_t)(idx % h1) * cells + idx / h1];
        } else if (idx < m.o_w2) {
            val = b1[idx - m.o_b1];
        } else if (idx < m.o_b2) {
            const long long j = idx - m.o_w2;
            val = w2[(size_t)(j % h2) * h1 + j / h2];
        } else if (idx < m.o_w3) {
            val = b2[idx - m.o_b2];
        } else if (idx < m.o_b3) {
            const long long j = idx - m.o_w3;
            const int out = (int)(j % m.cp);
            val = out < cells ? w3[(size_t)out * h2 + j / m.cp] : 0.0f;
        } else {
            const int out = (int)(idx - m.o_b3);
            val = out < cells ? b3[out] : 0.0f;
        }
    }
    dst[idx] = val;
}

// advances the draw counter behind the sampling workgroups (see k_sample_bump, hexvec_sample.hip)
__global__ void k_policy_bump(unsigned long long *state) { state[0] += 1ULL; }

// ---- grouping the boards by network: counting sort, three launches, no atomics on global memory ----
// scratch (ints): [0] tiles in use | [16 ..) per-workgroup bucket counts / offsets [GRP_BLOCKS][n_policies] |
// then the order: board of every position, buckets padded to whole tiles with -1
__device__ __forceinline__ int grp_chunk(int rows) { return (rows + GRP_BLOCKS - 1) / GRP_BLOCKS; }

__global__ __launch_bounds__(GRP_THREADS) void k_grp_count(MlpArgs a, int *counts)
{
    __shared__ int h[MAX_POLICIES];
    for (int k = (int)threadIdx.x; k < a.n_policies; k += GRP_THREADS) h[k] = 0;
    __syncthreads();
    const int chunk = grp_chunk(a.rows), lo = (int)blockIdx.x * chunk, hi = min(a.rows, lo + chunk);
    for (int r = lo + (int)threadIdx.x; r < hi; r += GRP_THREADS)
        if (!a.row_active || a.row_active[r] != 0) atomicAdd(&h[policy_key(a, r)], 1);
    __syncthreads();
    for (int k = (int)threadIdx.x; k < a.n_policies; k += GRP_THREADS) counts[(int)blockIdx.x * a.n_policies + k] = h[k];
}

// one workgroup of 4 wavefronts; lane g of a wavefront = grouping workgroup g (GRP_BLOCKS == 64), wavefront w takes
// the networks w, w + 4, ...: prefix sums over the workgroups are wavefront scans
__device__ __forceinline__ int wave_incl_scan(int x, int lane)
{
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const int y = __shfl_up(x, d);
        x += lane >= d ? y : 0;
    }
    return x;
}
__global__ __launch_bounds__(256) void k_grp_scan(int n_policies, int *ntiles, int *counts, int *order)
{
    static_assert(GRP_BLOCKS == 64, "one lane per grouping workgroup");
    __shared__ int tot[MAX_POLICIES], start[MAX_POLICIES];
    const int lane = (int)threadIdx.x & 63, wave = (int)threadIdx.x >> 6;
    for (int k = wave; k < n_policies; k += 4) {
        const int incl = wave_incl_scan(counts[lane * n_policies + k], lane);
        if (lane == 63) tot[k] = incl;
    }
    __syncthreads();
    if (wave == 0) { // buckets start on tile boundaries: exclusive scan of the padded sizes, 64 networks per trip
        int run = 0;
        for (int k0 = 0; k0 < n_policies; k0 += 64) {
            const int k = k0 + lane;
            const int padded = k < n_policies ? (tot[k] + 63) & ~63 : 0;
            const int incl = wave_incl_scan(padded, lane);
            if (k < n_policies) start[k] = run + incl - padded;
            run += __shfl(incl, 63);
        }
        if (lane == 0) *ntiles = run >> 6;
    }
    __syncthreads();
    for (int k = wave; k < n_policies; k += 4) {
        const int c = counts[lane * n_policies + k];
        counts[lane * n_policies + k] = start[k] + wave_incl_scan(c, lane) - c; // where workgroup `lane` writes bucket k
        const int total = tot[k], pad = ((total + 63) & ~63) - total;
        if (lane < pad) order[start[k] + total + lane] = -1;                    // holes up to the tile boundary
    }
}

__global__ __launch_bounds__(GRP_THREADS) void k_grp_scatter(MlpArgs a, const int *offsets, int *order)
{
    __shared__ int base[MAX_POLICIES], cnt[MAX_POLICIES];
    for (int k = (int)threadIdx.x; k < a.n_policies; k += GRP_THREADS) {
        base[k] = offsets[(int)blockIdx.x * a.n_policies + k];
        cnt[k] = 0;
    }
    __syncthreads();
    const int chunk = grp_chunk(a.rows), lo = (int)blockIdx.x * chunk, hi = min(a.rows, lo + chunk);
    for (int r = lo + (int)threadIdx.x; r < hi; r += GRP_THREADS) {
        if (!a.row_active || a.row_active[r] != 0) {
            const int key = policy_key(a, r);
            order[base[key] + atomicAdd(&cnt[key], 1)] = r;
        } else {
            a.actions[r] = 0; // nobody asked for this board's action
        }
    }
}

} // namespace hv

#ifdef HV_POL_STAMPS
extern "C" int hexvec_debug_policy_stamps(unsigned long long *host_dst, int ntiles)
{
    return (int)hipMemcpyFromSymbol(host_dst, HIP_SYMBOL(hv::g_pol_stamps), (size_t)ntiles * 10 * sizeof(unsigned long long));
}
#endif

static bool mlp_shape_ok(int cells, int h1, int h2)
{
    return cells >= 1 && cells <= HEXVEC_MAX_BOARD * HEXVEC_MAX_BOARD && h1 >= 16 && h2 >= 16 && !(h1 & 15) && !(h2 & 15) &&
           h1 <= 256 && h2 <= 256;
}

extern "C" size_t hexvec_mlp_weights_floats(int cells, int h1, int h2)
{
    if (!mlp_shape_ok(cells, h1, h2)) return 0;
    return (size_t)(hv::mlp_on_mfma(h1, h2) ? hv::mfma_shape(cells).stride : hv::mlp_shape(cells, h1, h2).stride);
}

extern "C" int hexvec_mlp_pack(const float *w1, const float *b1, const float *w2, const float *b2, const float *w3,
                               const float *b3, int cells, int h1, int h2, float *dst, void *stream)
{
    if (!w1 || !b1 || !w2 || !b2 || !w3 || !b3 || !dst || !mlp_shape_ok(cells, h1, h2)) {
        hv::set_error("hexvec_mlp_pack", "NULL argument, cells out of range or hidden widths not multiples of 16 in [16, 256]");
        return HEXVEC_E_ARG;
    }
    hv::PointerDeviceGuard guard(dst);
    if (!guard.ok) {
        hv::set_error("hexvec_mlp_pack", "dst is not memory of a visible GPU");
        return HEXVEC_E_ARG;
    }
    const size_t n = hexvec_mlp_weights_floats(cells, h1, h2);
    hipLaunchKernelGGL(hv::k_mlp_pack, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, w1, b1, w2, b2,
                       w3, b3, cells, h1, h2, dst);
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        hv::set_error("hexvec_mlp_pack", hipGetErrorString(e));
        return HEXVEC_E_HIP;
    }
    return HEXVEC_OK;
}

extern "C" size_t hexvec_mlp_lds_bytes(int cells, int h1, int h2)
{
    if (!mlp_shape_ok(cells, h1, h2)) return 0;
    if (hv::mlp_on_mfma(h1, h2)) { // the larger of the two MFMA kernels' needs
        const size_t one = (size_t)hv::MF_WAVES * 2 * ((32 * (size_t)cells + 15) & ~(size_t)15);
        const size_t two = 2 * ((32 * (size_t)cells + 15) & ~(size_t)15) + sizeof(float) * (2 * 64 * 32 + 3 * 32);
        return one > two ? one : two;
    }
    return 2 * ((64 * (size_t)cells + 15) & ~(size_t)15) + sizeof(float) * 64 * (size_t)(h1 + h2) + 4 * 64 * (3 * hv::POL_WAVES + 1);
}

extern "C" size_t hexvec_mlp_scratch_bytes(int rows, int n_policies)
{
    if (rows < 0 || n_policies < 1 || n_policies > hv::MAX_POLICIES) return 0;
    return sizeof(int) * (16 + (size_t)hv::GRP_BLOCKS * n_policies + (size_t)rows + 64 * (size_t)n_policies + 64);
}

extern "C" int hexvec_mlp_actions(const int8_t *obs, const uint8_t *legal, const uint8_t *row_active, const int64_t *slot,
                                  int slot_bias, const int32_t *policy_of_slot, int n_slots, int rows, int cells,
                                  const float *weights, int n_policies, int h1, int h2, int deterministic, uint64_t seed,
                                  uint32_t salt, int advance, unsigned long long *state2, int32_t *actions,
                                  float *logits_out, unsigned long long *bad_rows, void *scratch, size_t scratch_bytes,
                                  void *stream)
{
    using namespace hv;
    auto fail = [&](const char *m) {
        set_error("hexvec_mlp_actions", m);
        return HEXVEC_E_ARG;
    };
    if (!obs || !legal || !weights || !actions || rows < 0) return fail("obs / legal / weights / actions NULL or rows < 0");
    if (!mlp_shape_ok(cells, h1, h2)) return fail("cells out of range or hidden widths not multiples of 16 in [16, 256]");
    if (n_policies < 1 || n_policies > MAX_POLICIES) return fail("n_policies out of range");
    if (!deterministic && !state2) return fail("sampling needs the draw counter (state2)");
    if (((uintptr_t)weights & 63) != 0) return fail("weights must be 64-byte aligned");
    const bool grouped = slot != nullptr && n_policies > 1;
    if (slot && n_slots < 1) return fail("n_slots < 1");
    if (grouped && (!scratch || scratch_bytes < hexvec_mlp_scratch_bytes(rows, n_policies)))
        return fail("scratch missing or smaller than hexvec_mlp_scratch_bytes()");
    if (rows == 0) return HEXVEC_OK;
    PointerDeviceGuard guard(actions); // the device that owns the outputs (not the thread's current one)
    if (!guard.ok) return fail("actions is not memory of a visible GPU");

    hipStream_t st = (hipStream_t)stream;
    MlpArgs a;
    a.obs = (const signed char *)obs;
    a.legal = legal;
    a.row_active = row_active;
    a.slot = (const long long *)slot;
    a.policy_of_slot = policy_of_slot;
    a.slot_bias = slot_bias;
    a.n_slots = slot ? n_slots : 1;
    a.n_policies = n_policies;
    a.rows = rows;
    a.sh = mlp_shape(cells, h1, h2);
    a.ms = mfma_shape(cells);
    a.weights = weights;
    a.deterministic = deterministic;
    a.seed_lo = (u32)seed;
    a.seed_hi = (u32)(seed >> 32);
    a.salt = salt;
    a.state = state2;
    a.actions = actions;
    a.logits_out = logits_out;
    a.bad_rows = bad_rows;
    a.order = nullptr;
    a.ntiles = nullptr;
    unsigned tiles = (unsigned)((rows + 63) / 64);
    if (grouped) {
        int *hdr = (int *)scratch, *counts = hdr + 16, *order = counts + (size_t)GRP_BLOCKS * n_policies;
        hipLaunchKernelGGL(k_grp_count, dim3(GRP_BLOCKS), dim3(GRP_THREADS), 0, st, a, counts);
        hipLaunchKernelGGL(k_grp_scan, dim3(1), dim3(MAX_POLICIES), 0, st, n_policies, hdr, counts, order);
        hipLaunchKernelGGL(k_grp_scatter, dim3(GRP_BLOCKS), dim3(GRP_THREADS), 0, st, a, (const int *)counts, order);
        a.order = order;
        a.ntiles = hdr;
        a.row_active = nullptr; // inactive boards are not in the order
        tiles += (unsigned)n_policies;
    }
    const bool mfma = mlp_on_mfma(h1, h2);
    size_t lds;
    const void *fn;
    // few tiles: two wavefronts per tile, so that every SIMD holds two wavefronts where it would hold one (or none)
    static thread_local int simds_of[16] = {};
    int &simds = simds_of[guard.dev & 15];
    if (simds == 0) {
        int cus = 256;
        (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, guard.dev);
        simds = 4 * cus;
    }
    const bool split = mfma && (long long)2 * tiles * 2 <= (long long)3 * simds && getenv("HEXVEC_MLP_NOSPLIT") == nullptr;
    if (mfma && split) {
        lds = 2 * ((32 * (size_t)cells + 15) & ~(size_t)15) + sizeof(float) * (2 * 64 * 32 + 3 * 32);
        fn = (const void *)k_mlp64_mfma2;
    } else if (mfma) {
        lds = (size_t)MF_WAVES * 2 * ((32 * (size_t)cells + 15) & ~(size_t)15);
        fn = (const void *)k_mlp64_mfma;
    } else {
        lds = 2 * ((64 * (size_t)cells + 15) & ~(size_t)15) + sizeof(float) * 64 * (size_t)(h1 + h2) + 4 * 64 * (3 * POL_WAVES + 1);
        fn = (const void *)k_mlp_actions;
    }
    // more dynamic LDS than the default limit must be allowed per kernel AND per device: remembered per
    // (device, kernel) of this thread
    static thread_local size_t lds_allowed[16][3] = {};
    const int which = mfma ? (split ? 2 : 1) : 0;
    size_t &allowed = lds_allowed[guard.dev & 15][which];
    static thread_local size_t lds_limit[16] = {}; // what a workgroup may be given on that device at all
    size_t &limit = lds_limit[guard.dev & 15];
    if (limit == 0) {
        int v = 64 * 1024;
        (void)hipDeviceGetAttribute(&v, hipDeviceAttributeMaxSharedMemoryPerMultiprocessor, guard.dev); // one workgroup may take a CU's whole LDS
        limit = (size_t)v;
    }
    if (lds > limit) return fail("this network shape needs more LDS per workgroup than the device has (use the torch forward)");
    if (lds > 48 * 1024 && lds > allowed) {
        const hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) {
            set_error("hexvec_mlp_actions", hipGetErrorString(e));
            return HEXVEC_E_HIP;
        }
        allowed = lds;
    }
    if (mfma && split) // one tile of 32 boards per workgroup of two wavefronts
        hipLaunchKernelGGL(k_mlp64_mfma2, dim3(2 * tiles), dim3(128), lds, st, a);
    else if (mfma) // tiles of 32 boards, four to a workgroup
        hipLaunchKernelGGL(k_mlp64_mfma, dim3((2 * tiles + MF_WAVES - 1) / MF_WAVES), dim3(64 * MF_WAVES), lds, st, a);
    else
        hipLaunchKernelGGL(k_mlp_actions, dim3(tiles), dim3(64 * POL_WAVES), lds, st, a);
    if (advance && state2) hipLaunchKernelGGL(k_policy_bump, dim3(1), dim3(1), 0, st, state2);
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        set_error("hexvec_mlp_actions", hipGetErrorString(e));
        return HEXVEC_E_HIP;
    }
    return HEXVEC_OK;
}
