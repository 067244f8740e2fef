// hexvec.hip -- host side of libhexvec: the C ABI of include/hexvec.h.
// Kernels: hexvec_wpb.hip (wave-per-board, literal engine), hexvec_qpb_{s,m,l}.hip (lanes-per-board
// bitboards, the throughput path; hexvec_qpb.h), hexvec_lpb64.hip (lane-per-board
// union-find in LDS, the round-1 throughput path, the layout of HEXVEC_F_ROLLOUT_LAYOUT handles).
#include <stdlib.h>

#include "hexvec_common.h"

// =========================================================================
// host side: the C ABI
// =========================================================================
struct hexvec {
    int N, E, device;
    uint32_t flags, env_offset;
    uint64_t seed;
    hv::u32 *planes;
    hv::u32 *hot;
    long long *stats;
    bool fast_mode;      // throughput kernels + fast state (auto-reset handles)
    bool qpb;            // fast_mode: bitboard kernels (hexvec_qpb.h); else the LDS union-find ones (hexvec_lpb.h)
    unsigned char *fast;
    size_t fast_bytes;
    unsigned int *ticket; // hexvec_set_ticket
    // hexvec_attach_pool
    long long *pool_slot;
    unsigned char *pool_ended;
    unsigned long long *pool_state, *pool_bad;
    const float *pool_cdf;
    const int *pool_pos;
    float pool_prob;
    int pool_size, pool_npol;
    uint64_t pool_seed;
};

static thread_local char g_err[512] = "";
static thread_local const char *g_kernel = "";

static int fail(int code, const char *fmt, const char *detail)
{
    snprintf(g_err, sizeof(g_err), fmt, detail);
    return code;
}
// The caller's current device (PyTorch tracks it through the runtime) is restored on return.
struct DeviceGuard {
    int prev = -1;
    bool ok = true;
    explicit DeviceGuard(int dev)
    {
        if (hipGetDevice(&prev) != hipSuccess) prev = -1;
        if (prev != dev) ok = hipSetDevice(dev) == hipSuccess;
    }
    ~DeviceGuard()
    {
        int cur = -1;
        if (prev >= 0 && hipGetDevice(&cur) == hipSuccess && cur != prev) (void)hipSetDevice(prev);
    }
};
#define HV_DEVICE(h)                                                        \
    DeviceGuard dev_guard_((h)->device);                                    \
    if (!dev_guard_.ok) return fail(HEXVEC_E_HIP, "%s", "hipSetDevice failed")

#define HV_HIP(call)                                                              \
    do {                                                                          \
        hipError_t e_ = (call);                                                   \
        if (e_ != hipSuccess) return fail(HEXVEC_E_HIP, #call ": %s", hipGetErrorString(e_)); \
    } while (0)

extern "C" const char *hexvec_last_error(void) { return g_err; }
namespace hv {
void set_error(const char *what, const char *detail) { snprintf(g_err, sizeof(g_err), "%s: %s", what, detail); }
} // namespace hv
extern "C" const char *hexvec_last_kernel(void) { return g_kernel; }

namespace hv {

static KArgs base_args(const hexvec *h)
{
    KArgs a;
    memset(&a, 0, sizeof(a));
    a.fast = h->fast;
    a.planes = h->planes;
    a.hot = h->hot;
    a.stats = h->stats;
    a.E = h->E;
    a.env_offset = h->env_offset;
    a.seed_lo = (u32)h->seed;
    a.seed_hi = (u32)(h->seed >> 32);
    a.flags = h->flags;
    a.ticket = h->ticket;
    a.T = 1;
    if (h->pool_slot) {
        a.pool_slot = h->pool_slot;
        a.pool_ended = h->pool_ended;
        a.pool_ctr = reinterpret_cast<unsigned long long *>(h->fast + h->fast_bytes - 16); // the state's last two words
        a.pool_state = h->pool_state;
        a.pool_cdf = h->pool_cdf;
        a.pool_pos = h->pool_pos;
        a.pool_bad = h->pool_bad;
        a.pool_prob = h->pool_prob;
        a.pool_size = h->pool_size;
        a.pool_npol = h->pool_npol;
        a.pool_seed_lo = (u32)h->pool_seed;
        a.pool_seed_hi = (u32)(h->pool_seed >> 32);
    }
    return a;
}

static const char *kname(int mode)
{
    switch (mode) {
    case M_APPLY: return "k_hex<apply>";
    case M_ROLLOUT2: return "k_hex<rollout2>";
    case M_STEP: return "k_hex<step>";
    case M_ROLLOUT1: return "k_hex<rollout1>";
    case M_SETSTATE: return "k_hex<set_state>";
    default: return "k_hex<reset>";
    }
}

static const char *lpb_kname(int mode)
{
    switch (mode) {
    case M_ROLLOUT2: return "k_lpb<rollout2>";
    case M_STEP: return "k_lpb<step>";
    case M_ROLLOUT1: return "k_lpb<rollout1>";
    case M_SETBOARD: return "k_lpb<set_boards>";
    default: return "k_lpb<reset>";
    }
}

static const char *qpb_kname(int mode)
{
    switch (mode) {
    case M_ROLLOUT2: return "k_qpb<rollout2>";
    case M_STEP:
    case M_STEP_RNG: return "k_qpb<step>";
    case M_STEP_AGENT: return "k_qpb<step_agent>";
    case M_STEP_OPP: return "k_qpb<step_opponent>";
    case M_ROLLOUT1: return "k_qpb<rollout1>";
    case M_SETBOARD: return "k_qpb<set_boards>";
    default: return "k_qpb<reset>";
    }
}
// the translation unit that holds board size N
#define HV_QPB_CALL(N, fn, ...) \
    ((N) <= 8 ? hv::qpb_s_##fn(__VA_ARGS__) : (N) <= 11 ? hv::qpb_m_##fn(__VA_ARGS__)   \
     : (N) <= 14 ? hv::qpb_l_##fn(__VA_ARGS__) : hv::qpb_x_##fn(__VA_ARGS__))

static int run(hexvec *h, int mode, const KArgs &a, void *stream)
{
    HV_DEVICE(h);
    const int enc = (h->flags & HEXVEC_F_ENC_B) ? 1 : 0;
    if (h->fast_mode && h->qpb) {
        g_kernel = qpb_kname(mode);
        HV_HIP(HV_QPB_CALL(h->N, launch, h->N, enc, mode, a, (hipStream_t)stream));
    } else if (h->fast_mode) {
        g_kernel = lpb_kname(mode);
        HV_HIP(lpb64_launch(h->N, enc, mode, a, (hipStream_t)stream));
    } else {
        g_kernel = kname(mode);
        HV_HIP(wpb_launch(h->N, enc, mode, a, (hipStream_t)stream));
    }
    return HEXVEC_OK;
}

} // namespace hv

extern "C" int hexvec_create(hexvec_t **out, int board_size, int num_envs, int device_id,
                             uint64_t seed, uint32_t flags, uint32_t env_offset)
{
    if (!out) return fail(HEXVEC_E_ARG, "%s", "out is NULL");
    *out = nullptr;
    if (board_size < HEXVEC_MIN_BOARD || board_size > HEXVEC_MAX_BOARD)
        return fail(HEXVEC_E_ARG, "%s", "board_size must be in 2..19");
    if (board_size > HEXVEC_MAX_BOARD_LABELS &&
        (flags & (HEXVEC_F_NO_AUTO_RESET | HEXVEC_F_EXACT_LABELS | HEXVEC_F_ROLLOUT_LAYOUT)))
        return fail(HEXVEC_E_ARG, "%s",
                    "board_size 15..19 runs on the default (bitboard) layout of auto-reset handles only: the literal "
                    "engine and the rollout layout keep one-byte labels / connectivity entries (2..14)");
    if (num_envs < 1) return fail(HEXVEC_E_ARG, "%s", "num_envs must be >= 1");
    if (!(flags & HEXVEC_F_ENC_B) && (flags & (HEXVEC_F_AGENT_WHITE | HEXVEC_F_AGENT_RANDOM)))
        return fail(HEXVEC_E_ARG, "%s",
                    "encoding A (hex-v0) supports the BLACK agent only (reference player_color=WHITE is broken)");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1)
        return fail(HEXVEC_E_NOGPU, "%s", "no HIP device visible: libhexvec has no CPU path");
    if (device_id < 0 || device_id >= ndev) return fail(HEXVEC_E_NOGPU, "%s", "device_id out of range");
    DeviceGuard dev_guard_(device_id);
    if (!dev_guard_.ok) return fail(HEXVEC_E_HIP, "%s", "hipSetDevice failed");
    hexvec *h = new hexvec();
    h->N = board_size;
    h->E = num_envs;
    h->device = device_id;
    h->flags = flags;
    h->env_offset = env_offset;
    h->seed = seed;
    h->planes = nullptr;
    h->hot = nullptr;
    h->stats = nullptr;
    h->fast = nullptr;
    h->pool_slot = nullptr;
    // auto-reset handles never see a ply after the end of a game: they run on the
    // lane-per-board kernels unless the caller wants the reference's label planes
    h->fast_mode = !(flags & (HEXVEC_F_NO_AUTO_RESET | HEXVEC_F_EXACT_LABELS)) && board_size >= 4;
    const size_t E = (size_t)num_envs;
    bool ok;
    h->qpb = !(flags & HEXVEC_F_ROLLOUT_LAYOUT);
    if (const char *ev = getenv("HEXVEC_FAMILY")) // A/B experiments
        h->qpb = strcmp(ev, "lpb") == 0 ? false : (strcmp(ev, "qpb") == 0 ? true : h->qpb);
    if (h->fast_mode && h->qpb) {
        const int bpw = 64 / hv::qpb_lanes(board_size);
        // + [E][4] 64-bit episode sums + the two step counters of hexvec_attach_pool (zeroed here, part of the state)
        const size_t bytes = (size_t)((num_envs + bpw - 1) / bpw) * hv::qpb_block_bytes(board_size) + E * 32 + 16;
        h->fast_bytes = bytes;
        ok = hipMalloc(&h->fast, bytes) == hipSuccess && hipMemset(h->fast + bytes - 16, 0, 16) == hipSuccess;
    } else if (h->fast_mode) {
        const int enc = (flags & HEXVEC_F_ENC_B) ? 1 : 0, two = (flags & HEXVEC_F_TWO_SIDED) ? 1 : 0;
        const size_t bytes = (size_t)((num_envs + 63) / 64) * hv::lpb64_block_bytes(board_size, enc, two);
        h->fast_bytes = bytes;
        ok = hipMalloc(&h->fast, bytes) == hipSuccess;
    } else {
        ok = hipMalloc(&h->planes, E * hv::PLANE_WORDS * 4) == hipSuccess &&
             hipMalloc(&h->hot, E * hv::HOT_WORDS * 4) == hipSuccess &&
             hipMalloc(&h->stats, E * hv::STAT_WORDS * 8) == hipSuccess;
    }
    if (!ok) {
        hexvec_destroy(h);
        return fail(HEXVEC_E_HIP, "%s", "hipMalloc of the env state failed");
    }
    hv::KArgs a = hv::base_args(h);
    a.init = 1;
    int rc = hv::run(h, hv::M_RESET, a, nullptr);
    if (rc == HEXVEC_OK && hipStreamSynchronize(nullptr) != hipSuccess)
        rc = fail(HEXVEC_E_HIP, "%s", "initial reset kernel failed");
    if (rc != HEXVEC_OK) {
        hexvec_destroy(h);
        return rc;
    }
    *out = h;
    return HEXVEC_OK;
}

extern "C" int hexvec_destroy(hexvec_t *h)
{
    if (!h) return HEXVEC_OK;
    DeviceGuard dev_guard_(h->device);
    if (h->planes) (void)hipFree(h->planes);
    if (h->hot) (void)hipFree(h->hot);
    if (h->stats) (void)hipFree(h->stats);
    if (h->fast) (void)hipFree(h->fast);
    delete h;
    return HEXVEC_OK;
}

#define HV_HANDLE(h) \
    if (!(h)) return fail(HEXVEC_E_ARG, "%s", "handle is NULL")

extern "C" int hexvec_reset(hexvec_t *h, const uint8_t *env_mask, const int32_t *opp_actions,
                            int8_t *obs, uint8_t *legal, void *stream)
{
    HV_HANDLE(h);
    hv::KArgs a = hv::base_args(h);
    a.env_mask = env_mask;
    a.opp_actions = opp_actions;
    a.obs = (signed char *)obs;
    a.legal = legal;
    return hv::run(h, hv::M_RESET, a, stream);
}

extern "C" int hexvec_step(hexvec_t *h, const int32_t *actions, const int32_t *opp_actions,
                           int8_t *obs, uint8_t *legal, float *reward, uint8_t *done,
                           int8_t *winner, int8_t *terminal_obs, void *stream)
{
    HV_HANDLE(h);
    hv::KArgs a = hv::base_args(h);
    a.actions = actions;
    a.opp_actions = opp_actions;
    a.obs = (signed char *)obs;
    a.legal = legal;
    a.reward = reward;
    a.done = done;
    a.winner = (signed char *)winner;
    a.tobs = (signed char *)terminal_obs;
    // both sides drawn by the library on a self-play handle of the default layout (the random-policy workload): the
    // same step from the kernel compiled without the external-action / split-phase paths (739 -> ~590 vector
    // instructions per wavefront at 11x11, and the launch is bound by exactly those)
    const bool rng_both = !actions && !opp_actions && h->fast_mode && h->qpb && !(h->flags & HEXVEC_F_TWO_SIDED);
    return hv::run(h, rng_both ? hv::M_STEP_RNG : hv::M_STEP, a, stream);
}

extern "C" int hexvec_step_agent(hexvec_t *h, const int32_t *actions, int8_t *obs, uint8_t *legal,
                                 float *reward, uint8_t *done, int8_t *winner, int8_t *terminal_obs,
                                 uint8_t *pending, void *stream)
{
    HV_HANDLE(h);
    if (!h->fast_mode || (h->flags & HEXVEC_F_TWO_SIDED))
        return fail(HEXVEC_E_MODE, "%s", "the split step needs an auto-reset self-play handle");
    hv::KArgs a = hv::base_args(h);
    a.phase = 1;
    a.actions = actions;
    a.obs = (signed char *)obs;
    a.legal = legal;
    a.reward = reward;
    a.done = done;
    a.winner = (signed char *)winner;
    a.tobs = (signed char *)terminal_obs;
    a.pending = pending;
    return hv::run(h, h->qpb ? hv::M_STEP_AGENT : hv::M_STEP, a, stream);
}

extern "C" int hexvec_step_opponent(hexvec_t *h, const int32_t *opp_actions, int8_t *obs, uint8_t *legal,
                                    float *reward, uint8_t *done, int8_t *winner, int8_t *terminal_obs,
                                    uint8_t *pending, void *stream)
{
    HV_HANDLE(h);
    if (!h->fast_mode || (h->flags & HEXVEC_F_TWO_SIDED))
        return fail(HEXVEC_E_MODE, "%s", "the split step needs an auto-reset self-play handle");
    hv::KArgs a = hv::base_args(h);
    a.phase = 2;
    a.opp_actions = opp_actions;
    a.obs = (signed char *)obs;
    a.legal = legal;
    a.reward = reward;
    a.done = done;
    a.winner = (signed char *)winner;
    a.tobs = (signed char *)terminal_obs;
    a.pending = pending;
    return hv::run(h, h->qpb ? hv::M_STEP_OPP : hv::M_STEP, a, stream);
}

extern "C" int hexvec_apply_moves(hexvec_t *h, const int32_t *actions, int T, int8_t *ret_out,
                                  uint8_t *done_out, int8_t *r_black_out, uint8_t *legal_out,
                                  int8_t *obs_out, void *stream)
{
    HV_HANDLE(h);
    if (!(h->flags & HEXVEC_F_TWO_SIDED))
        return fail(HEXVEC_E_MODE, "%s", "hexvec_apply_moves needs a HEXVEC_F_TWO_SIDED handle");
    if (h->fast_mode)
        return fail(HEXVEC_E_MODE, "%s",
                    "hexvec_apply_moves needs HEXVEC_F_NO_AUTO_RESET or HEXVEC_F_EXACT_LABELS (literal plies)");
    if (!actions || T < 0) return fail(HEXVEC_E_ARG, "%s", "actions is NULL or T < 0");
    if (T == 0) return HEXVEC_OK;
    hv::KArgs a = hv::base_args(h);
    a.T = T;
    a.actions = actions;
    a.winner = (signed char *)ret_out;
    a.done = done_out;
    a.rblack = (signed char *)r_black_out;
    a.legal = legal_out;
    a.obs = (signed char *)obs_out;
    return hv::run(h, hv::M_APPLY, a, stream);
}

extern "C" int hexvec_rollout_random(hexvec_t *h, int T, int8_t *obs, uint8_t *legal,
                                     int32_t *actions, float *reward, uint8_t *done, void *stream)
{
    HV_HANDLE(h);
    if (!(h->flags & HEXVEC_F_TWO_SIDED))
        return fail(HEXVEC_E_MODE, "%s", "hexvec_rollout_random needs a HEXVEC_F_TWO_SIDED handle");
    if (h->flags & HEXVEC_F_NO_AUTO_RESET)
        return fail(HEXVEC_E_MODE, "%s", "hexvec_rollout_random needs auto reset");
    if (T < 0) return fail(HEXVEC_E_ARG, "%s", "T < 0");
    if (T == 0) return HEXVEC_OK;
    hv::KArgs a = hv::base_args(h);
    a.T = T;
    a.obs = (signed char *)obs;
    a.legal = legal;
    a.actions_out = actions;
    a.reward = reward;
    a.done = done;
    return hv::run(h, hv::M_ROLLOUT2, a, stream);
}

extern "C" int hexvec_rollout_selfplay(hexvec_t *h, int T, int8_t *obs, uint8_t *legal,
                                       int32_t *actions, float *reward, uint8_t *done,
                                       void *stream)
{
    HV_HANDLE(h);
    if (h->flags & HEXVEC_F_TWO_SIDED)
        return fail(HEXVEC_E_MODE, "%s", "hexvec_rollout_selfplay needs a self-play handle");
    if (h->flags & HEXVEC_F_NO_AUTO_RESET)
        return fail(HEXVEC_E_MODE, "%s", "hexvec_rollout_selfplay needs auto reset");
    if (T < 0) return fail(HEXVEC_E_ARG, "%s", "T < 0");
    if (T == 0) return HEXVEC_OK;
    hv::KArgs a = hv::base_args(h);
    a.T = T;
    a.obs = (signed char *)obs;
    a.legal = legal;
    a.actions_out = actions;
    a.reward = reward;
    a.done = done;
    return hv::run(h, hv::M_ROLLOUT1, a, stream);
}

extern "C" int hexvec_get_state(hexvec_t *h, int8_t *board, uint8_t *labels, uint8_t *cur_player,
                                void *stream)
{
    HV_HANDLE(h);
    HV_DEVICE(h);
    if (h->fast_mode) {
        if (labels)
            return fail(HEXVEC_E_MODE, "%s",
                        "label planes exist only on HEXVEC_F_EXACT_LABELS / NO_AUTO_RESET handles");
        const int enc = (h->flags & HEXVEC_F_ENC_B) ? 1 : 0;
        if (h->qpb) {
            g_kernel = "k_qpb_get_state";
            HV_HIP(HV_QPB_CALL(h->N, get_state, h->fast, h->N, enc, h->E, (signed char *)board, cur_player,
                               (hipStream_t)stream));
            return HEXVEC_OK;
        }
        g_kernel = "k_lpb_get_state";
        HV_HIP(hv::lpb64_get_state(h->fast, h->N, enc, (h->flags & HEXVEC_F_TWO_SIDED) ? 1 : 0, h->E, (signed char *)board,
                                   cur_player, (hipStream_t)stream));
        return HEXVEC_OK;
    }
    g_kernel = "k_get_state";
    HV_HIP(hv::wpb_get_state(h->planes, h->hot, h->E, h->N, (h->flags & HEXVEC_F_ENC_B) ? 1 : 0,
                             (signed char *)board, labels, cur_player, (hipStream_t)stream));
    return HEXVEC_OK;
}

extern "C" int hexvec_set_state(hexvec_t *h, const int8_t *board, const uint8_t *labels, int cur_player,
                                void *stream)
{
    HV_HANDLE(h);
    if (h->fast_mode)
        return fail(HEXVEC_E_MODE, "%s",
                    "hexvec_set_state needs a HEXVEC_F_NO_AUTO_RESET or HEXVEC_F_EXACT_LABELS handle");
    if (!board || (cur_player != 0 && cur_player != 1))
        return fail(HEXVEC_E_ARG, "%s", "board is NULL or cur_player not 0/1");
    hv::KArgs a = hv::base_args(h);
    a.board_in = (const signed char *)board;
    a.labels_in = labels;
    a.cur_in = cur_player;
    return hv::run(h, hv::M_SETSTATE, a, stream);
}

extern "C" int hexvec_set_boards(hexvec_t *h, const uint8_t *env_mask, const int8_t *boards, int8_t *obs,
                                 uint8_t *legal, uint8_t *pending, void *stream)
{
    HV_HANDLE(h);
    if (!h->fast_mode)
        return fail(HEXVEC_E_MODE, "%s", "hexvec_set_boards needs an auto-reset handle (use hexvec_set_state)");
    if (!boards) return fail(HEXVEC_E_ARG, "%s", "boards is NULL");
    hv::KArgs a = hv::base_args(h);
    a.env_mask = env_mask;
    a.board_in = (const signed char *)boards;
    a.obs = (signed char *)obs;
    a.legal = legal;
    a.pending = pending;
    return hv::run(h, hv::M_SETBOARD, a, stream);
}

extern "C" int hexvec_episode_stats(hexvec_t *h, double *out4, void *stream)
{
    HV_HANDLE(h);
    if (!out4) return fail(HEXVEC_E_ARG, "%s", "out4 is NULL");
    HV_DEVICE(h);
    HV_HIP(hipMemsetAsync(out4, 0, 4 * sizeof(double), (hipStream_t)stream)); // the blocks add their partial sums
    if (h->fast_mode && h->qpb) {
        g_kernel = "k_qpb_stats";
        HV_HIP(HV_QPB_CALL(h->N, stats, h->fast, h->N, h->E, out4, nullptr, (hipStream_t)stream));
        return HEXVEC_OK;
    }
    if (h->fast_mode) {
        g_kernel = "k_lpb_stats";
        const int enc = (h->flags & HEXVEC_F_ENC_B) ? 1 : 0;
        HV_HIP(hv::lpb64_stats(h->fast, h->N, enc, (h->flags & HEXVEC_F_TWO_SIDED) ? 1 : 0, h->E, out4, nullptr,
                               (hipStream_t)stream));
        return HEXVEC_OK;
    }
    g_kernel = "k_stats";
    HV_HIP(hv::wpb_stats(h->stats, h->E, out4, (hipStream_t)stream));
    return HEXVEC_OK;
}

extern "C" int hexvec_agent_colours(hexvec_t *h, uint8_t *out, void *stream)
{
    HV_HANDLE(h);
    if (!out) return fail(HEXVEC_E_ARG, "%s", "out is NULL");
    HV_DEVICE(h);
    if (h->fast_mode && h->qpb) {
        g_kernel = "k_qpb_stats";
        HV_HIP(HV_QPB_CALL(h->N, stats, h->fast, h->N, h->E, nullptr, out, (hipStream_t)stream));
        return HEXVEC_OK;
    }
    if (h->fast_mode) {
        g_kernel = "k_lpb_stats";
        const int enc = (h->flags & HEXVEC_F_ENC_B) ? 1 : 0;
        HV_HIP(hv::lpb64_stats(h->fast, h->N, enc, (h->flags & HEXVEC_F_TWO_SIDED) ? 1 : 0, h->E, nullptr, out,
                               (hipStream_t)stream));
        return HEXVEC_OK;
    }
    g_kernel = "k_agents";
    HV_HIP(hv::wpb_agents(h->hot, h->E, out, (hipStream_t)stream));
    return HEXVEC_OK;
}

// ---- checkpoint of the env state (opaque, valid for a handle created with the same arguments) ----
static size_t state_bytes(const hexvec *h)
{
    const size_t E = (size_t)h->E;
    return h->fast_mode ? h->fast_bytes : E * (hv::PLANE_WORDS + hv::HOT_WORDS + 2 * hv::STAT_WORDS) * 4;
}
extern "C" size_t hexvec_state_bytes(const hexvec_t *h) { return h ? state_bytes(h) : 0; }

extern "C" int hexvec_export_state(hexvec_t *h, void *dst, void *stream)
{
    HV_HANDLE(h);
    if (!dst) return fail(HEXVEC_E_ARG, "%s", "dst is NULL");
    HV_DEVICE(h);
    hipStream_t st = (hipStream_t)stream;
    if (h->fast_mode) {
        HV_HIP(hipMemcpyAsync(dst, h->fast, h->fast_bytes, hipMemcpyDeviceToDevice, st));
    } else {
        const size_t E = (size_t)h->E;
        char *d = (char *)dst;
        HV_HIP(hipMemcpyAsync(d, h->planes, E * hv::PLANE_WORDS * 4, hipMemcpyDeviceToDevice, st));
        d += E * hv::PLANE_WORDS * 4;
        HV_HIP(hipMemcpyAsync(d, h->hot, E * hv::HOT_WORDS * 4, hipMemcpyDeviceToDevice, st));
        d += E * hv::HOT_WORDS * 4;
        HV_HIP(hipMemcpyAsync(d, h->stats, E * hv::STAT_WORDS * 8, hipMemcpyDeviceToDevice, st));
    }
    return HEXVEC_OK;
}

extern "C" int hexvec_import_state(hexvec_t *h, const void *src, void *stream)
{
    HV_HANDLE(h);
    if (!src) return fail(HEXVEC_E_ARG, "%s", "src is NULL");
    HV_DEVICE(h);
    hipStream_t st = (hipStream_t)stream;
    if (h->fast_mode) {
        HV_HIP(hipMemcpyAsync(h->fast, src, h->fast_bytes, hipMemcpyDeviceToDevice, st));
    } else {
        const size_t E = (size_t)h->E;
        const char *d = (const char *)src;
        HV_HIP(hipMemcpyAsync(h->planes, d, E * hv::PLANE_WORDS * 4, hipMemcpyDeviceToDevice, st));
        d += E * hv::PLANE_WORDS * 4;
        HV_HIP(hipMemcpyAsync(h->hot, d, E * hv::HOT_WORDS * 4, hipMemcpyDeviceToDevice, st));
        d += E * hv::HOT_WORDS * 4;
        HV_HIP(hipMemcpyAsync(h->stats, d, E * hv::STAT_WORDS * 8, hipMemcpyDeviceToDevice, st));
    }
    return HEXVEC_OK;
}

extern "C" int hexvec_attach_pool(hexvec_t *h, int64_t *slot, uint8_t *ended, float prob_best, int pool_size, uint64_t seed,
                                  unsigned long long *state2, const float *open_cdf, const int32_t *policy_of_slot,
                                  int n_policies, unsigned long long *bad_rows)
{
    HV_HANDLE(h);
    if (!slot) {
        h->pool_slot = nullptr;
        return HEXVEC_OK;
    }
    if (!h->fast_mode || !h->qpb || (h->flags & HEXVEC_F_TWO_SIDED))
        return fail(HEXVEC_E_MODE, "%s", "hexvec_attach_pool needs an auto-reset self-play handle on the default layout");
    if (!ended || !state2 || pool_size < 1 || n_policies < 1 || !(prob_best >= 0.0f && prob_best <= 1.0f))
        return fail(HEXVEC_E_ARG, "%s", "hexvec_attach_pool: ended / state2 NULL, pool_size or n_policies < 1, or prob_best outside [0, 1]");
    h->pool_slot = (long long *)slot;
    h->pool_ended = ended;
    h->pool_state = state2;
    h->pool_bad = bad_rows;
    h->pool_cdf = open_cdf;
    h->pool_pos = policy_of_slot;
    h->pool_prob = prob_best;
    h->pool_size = pool_size;
    h->pool_npol = n_policies;
    h->pool_seed = seed;
    return HEXVEC_OK;
}

extern "C" int hexvec_set_ticket(hexvec_t *h, uint32_t *ticket)
{
    HV_HANDLE(h);
    if (h->fast_mode)
        return fail(HEXVEC_E_MODE, "%s", "hexvec_set_ticket: only the literal engine (NO_AUTO_RESET / EXACT_LABELS handles) signals a ticket");
    h->ticket = ticket;
    return HEXVEC_OK;
}

extern "C" int hexvec_stream_synchronize(hexvec_t *h, void *stream)
{
    HV_HANDLE(h);
    HV_DEVICE(h);
    HV_HIP(hipStreamSynchronize((hipStream_t)stream));
    return HEXVEC_OK;
}

extern "C" int hexvec_board_size(const hexvec_t *h) { return h ? h->N : HEXVEC_E_ARG; }
extern "C" int hexvec_num_envs(const hexvec_t *h) { return h ? h->E : HEXVEC_E_ARG; }
extern "C" uint32_t hexvec_flags(const hexvec_t *h) { return h ? h->flags : 0u; }
