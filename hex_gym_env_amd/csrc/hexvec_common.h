// hexvec_common.h -- declarations shared by the translation units of libhexvec.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include "hexvec.h"

namespace hv {

typedef unsigned int u32;
typedef unsigned long long u64;

constexpr int WPB = 4;        // wavefronts (= boards) per workgroup
constexpr int PLANE_WORDS = 3 * 64; // per env: L0, L1, F|FT<<4   (768 B)
constexpr int HOT_WORDS = 8;  // per env: counters touched every step (32 B)
constexpr int STAT_WORDS = 4; // per env: episode sums (64-bit each), touched at episode end

// M_STEP_RNG: hexvec_step on a self-play handle with BOTH sides drawn by the library (actions == opp_actions == NULL, the
// random-policy workload): the whole-step loop compiled without the external-action and split-phase paths (bitboard family)
// M_STEP_AGENT / M_STEP_OPP: the two halves of the split step (hexvec_step_agent / hexvec_step_opponent) as kernels of their
// own instead of the `phase` branch of M_STEP (bitboard family; the other families keep M_STEP + phase)
enum Mode { M_APPLY = 0, M_ROLLOUT2 = 1, M_STEP = 2, M_RESET = 3, M_ROLLOUT1 = 4, M_SETSTATE = 5, M_SETBOARD = 6, M_STEP_RNG = 7,
            M_STEP_AGENT = 8, M_STEP_OPP = 9 };

struct KArgs {
    unsigned char *fast; // lane-per-board state (fast handles)
    u32 *planes;
    u32 *hot;
    long long *stats; // literal-engine handles: [E][STAT_WORDS] episode sums since create (64-bit: sum_len passes 2^31 within hours)
    int E;
    u32 env_offset, seed_lo, seed_hi, flags;
    int T;
    int init;
    const int *actions;
    const int *opp_actions;
    const unsigned char *env_mask;
    signed char *obs;
    unsigned char *legal;
    float *reward;
    unsigned char *done;
    signed char *winner;
    signed char *tobs;
    signed char *rblack;
    int *actions_out;
    const signed char *board_in;   // M_SETSTATE
    const unsigned char *labels_in; // M_SETSTATE (optional)
    int cur_in;
    int phase;              // M_STEP on self-play handles: 0 whole step, 1 agent ply only, 2 opponent plies only
    unsigned char *pending; // split step: 1 = reply owed, 2 = opening owed
    unsigned int *ticket;   // literal-engine handles: completion ticket in host-pinned memory (hexvec_set_ticket)
    // hexvec_attach_pool (split step on bitboard handles): setup_opponents + openings in the launch's tail
    long long *pool_slot;   // NULL = not attached
    unsigned char *pool_ended;
    unsigned long long *pool_ctr; // handle-owned {steps as the agent's launches read it, as the opponent's launches read it}
    unsigned long long *pool_state; // the caller's draw counter: written by the opponent's launch
    const float *pool_cdf;
    const int *pool_pos;
    unsigned long long *pool_bad;
    float pool_prob;
    int pool_size, pool_npol;
    u32 pool_seed_lo, pool_seed_hi;
};


// ------------------------------------------------------------------ helpers
__device__ __forceinline__ u32 mix32(u32 x)
{
    x ^= x >> 16;
    x *= 0x7feb352dU;
    x ^= x >> 15;
    x *= 0x846ca68bU;
    x ^= x >> 16;
    return x;
}
// Gumbel(0, 1) noise from 32 hash bits: G = -ln(-ln(u)), u = (top 24 bits + 0.5) / 2^24 in (0, 1).  Raw v_log_f32
// (log2) twice: neither argument can be subnormal (u >= 2^-25, -log2(u) >= 8.6e-8), so the library log's range
// fix-ups are not needed.  Every sampling kernel uses THIS function: equal hashes give bit-equal noise.
__device__ __forceinline__ float gumbel_from_hash(u32 h)
{
    const float u = ((float)(h >> 8) + 0.5f) * (1.0f / 16777216.0f);
    const float e = -__builtin_amdgcn_logf(u);                            // -log2(u)
    return __builtin_fmaf(-0.69314718f, __builtin_amdgcn_logf(e), 0.36651292f); // -ln(ln2 * e)
}
// the counter hash of hexvec_pick_opponents / hexvec_sample_masked: one key per (seed, salt, counter, row)
__device__ __forceinline__ u32 row_key(u32 row, unsigned long long ctr, u32 seed_lo, u32 seed_hi, u32 salt)
{
    return mix32((row + (u32)ctr * 0x85EBCA6BU) * 0x9E3779B9U + seed_lo) ^ seed_hi ^ ((u32)(ctr >> 32) * 0xC2B2AE35U) ^
           mix32(salt * 0x27D4EB2FU + 0x165667B1U);
}
// SelfPlayEnv.setup_opponents (SelfplayWrapper.py:97-104) for one board: the best model with probability prob_best,
// else a uniformly drawn pool member
__device__ __forceinline__ long long pick_slot(u32 row, unsigned long long ctr, u32 seed_lo, u32 seed_hi, u32 salt,
                                               float prob_best, int pool_size)
{
    const u32 k = row_key(row, ctr, seed_lo, seed_hi, salt);
    const float u = ((float)(mix32(k) >> 8) + 0.5f) * (1.0f / 16777216.0f);
    const u32 member = __umulhi(mix32(k + 0x9E3779B9U), (u32)pool_size);
    return u < prob_best ? -1LL : (long long)member;
}
__device__ __forceinline__ int rfl(int v) { return __builtin_amdgcn_readfirstlane(v); }
__device__ __forceinline__ u32 rflu(u32 v) { return (u32)__builtin_amdgcn_readfirstlane((int)v); }
__device__ __forceinline__ u32 rdlane(u32 v, int l) { return (u32)__builtin_amdgcn_readlane((int)v, l); }
__device__ __forceinline__ u32 mbcnt(u64 m)
{
    return __builtin_amdgcn_mbcnt_hi((u32)(m >> 32), __builtin_amdgcn_mbcnt_lo((u32)m, 0u));
}
// byte `idx` of a striped plane (idx wave-uniform -> result wave-uniform)
__device__ __forceinline__ u32 byte_at(u32 reg, int idx)
{
    return (rdlane(reg, idx & 63) >> ((idx >> 6) * 8)) & 0xffu;
}
__device__ __forceinline__ u32 put_byte(u32 reg, int idx, u32 val, u32 lane)
{
    const int sh = (idx >> 6) * 8;
    const u32 nw = (reg & ~(0xffu << sh)) | (val << sh);
    return lane == (u32)(idx & 63) ? nw : reg;
}


// thread-local text behind hexvec_last_error(), shared by every translation unit (defined in hexvec.hip)
void set_error(const char *what, const char *detail);

// The entry points that take no handle (hexvec_sample_masked, hexvec_pick_opponents, hexvec_mlp_*) work on the device
// that OWNS their output buffer, whatever the calling thread's current device is: launches, per-device function
// attributes and device queries inside the guard go to that device; the caller's current device is restored on return.
struct PointerDeviceGuard {
    int prev = -1, dev = -1;
    bool ok = false;
    explicit PointerDeviceGuard(const void *device_ptr)
    {
        hipPointerAttribute_t at;
        if (hipGetDevice(&prev) != hipSuccess) prev = -1;
        if (device_ptr && hipPointerGetAttributes(&at, device_ptr) == hipSuccess && at.type == hipMemoryTypeDevice) {
            dev = at.device;
            ok = dev == prev || hipSetDevice(dev) == hipSuccess;
        } else {
            (void)hipGetLastError(); // (not a device pointer: the error is reported by the caller, not left pending)
        }
    }
    ~PointerDeviceGuard()
    {
        int cur = -1;
        if (prev >= 0 && hipGetDevice(&cur) == hipSuccess && cur != prev) (void)hipSetDevice(prev);
    }
};

// launchers, one per translation unit (so the kernel families compile in parallel)
hipError_t wpb_launch(int N, int enc, int mode, const KArgs &a, hipStream_t st);
hipError_t wpb_get_state(const u32 *planes, const u32 *hot, int E, int N, int enc, signed char *board,
                         unsigned char *labels, unsigned char *cur, hipStream_t st);
hipError_t wpb_stats(const long long *stats, int E, double *out4, hipStream_t st);
hipError_t wpb_agents(const u32 *hot, int E, unsigned char *out, hipStream_t st);

hipError_t lpb64_launch(int N, int enc, int mode, const KArgs &a, hipStream_t st);
size_t lpb64_block_bytes(int N, int enc, int two);
hipError_t lpb64_get_state(const unsigned char *fast, int N, int enc, int two, int E, signed char *board,
                           unsigned char *cur, hipStream_t st);
hipError_t lpb64_stats(const unsigned char *fast, int N, int enc, int two, int E, double *out4,
                       unsigned char *agents, hipStream_t st);

// lanes-per-board bitboard kernels (hexvec_qpb.h), one translation unit per size range
#define HV_QPB_DECL(x)                                                                                        \
    hipError_t qpb_##x##_launch(int N, int enc, int mode, const KArgs &a, hipStream_t st);                    \
    hipError_t qpb_##x##_get_state(const unsigned char *fast, int N, int enc, int E, signed char *board,      \
                                   unsigned char *cur, hipStream_t st);                                       \
    hipError_t qpb_##x##_stats(const unsigned char *fast, int N, int E, double *out4, unsigned char *agents,  \
                               hipStream_t st);
HV_QPB_DECL(s)
HV_QPB_DECL(m)
HV_QPB_DECL(l)
HV_QPB_DECL(x)
#undef HV_QPB_DECL
// geometry of that family (must match qpb::Geo<N>)
inline int qpb_lanes(int N)
{
    const int jw = (N * N + 31) / 32;
    return jw <= 1 ? 1 : jw <= 2 ? 2 : jw <= 4 ? 4 : jw <= 8 ? 8 : 16;
}
inline size_t qpb_block_bytes(int N)
{
    const int L = qpb_lanes(N);
    return (size_t)(8 + (3 + L - 1) / L) * 64 * 4;
}

} // namespace hv
