/*
 * hexvec.h -- C ABI of libhexvec: thousands of independent Hex games stepped in
 * lockstep on one MI355X (gfx950) by hand-written HIP kernels (one lane per board
 * on the throughput path, one wavefront per board on the literal path; see below).
 *
 * The reference (MBPrdctns/hex_gym_env) has NO native/FFI interface: its hot
 * path sits behind plain Python classes.  Each entry point below therefore
 * names the reference Python interface it replaces (paths under minihex/).
 * The Python binding a maintainer adds is in INTEGRATION.md (ctypes).
 *
 * Conventions
 *   - every function returns 0 on success, a negative HEXVEC_E_* on failure;
 *     text via hexvec_last_error() (thread-local).  Illegal GAME actions are
 *     not errors: they produce the reference's result (ret 3, HexGame.py:91-92).
 *   - all data pointers are DEVICE pointers owned by the caller (torch
 *     tensors); the library owns only its internal state.  Optional outputs
 *     may be NULL.
 *   - `stream` is a hipStream_t (NULL = default stream).  Calls enqueue and
 *     never synchronise; one handle = one device = one caller thread at a time.
 *   - layouts: obs int8 [E, N*N] row-major board exactly as the reference's
 *     `simulator.board` (encoding A: BLACK 0 / WHITE 1 / EMPTY 2, absolute;
 *     encoding B: -1 / 0 / +1 in the perspective of the player to move);
 *     legal uint8 [E, N*N] (1 = empty); actions int32 [E]; reward float32 [E];
 *     done uint8 [E]; winner/ret int8 [E]: -1 None, 0 BLACK, 1 WHITE, 3 illegal.
 *   - board_size 2..19 on the default layout of auto-reset handles (bitboards: up to 16 lanes x 32
 *     cells per board); 2..14 on NO_AUTO_RESET / EXACT_LABELS / ROLLOUT_LAYOUT handles (labels and
 *     union-find entries are one byte; a padded label plane is one dword per lane); anything else is
 *     rejected with HEXVEC_E_ARG.  (The reference sets no limit, HexSingleGame.py:163-172; its scripts
 *     stop at 11x11.)
 *
 * Three kernel families serve a handle, chosen at create time:
 *   - auto-reset handles (the default) run on the bitboard kernels: 1 / 2 / 4 / 8
 *     lanes own one board (4 at 9x9..11x11), the whole game state is 8 bitboards in
 *     registers (144 B per board in HBM between calls), connectivity by edge-connected
 *     sets grown with a frontier flood.  This is the layout for host-stepped use
 *     (hexvec_step and the split step move the least state) and for any board count.
 *   - auto-reset handles created with HEXVEC_F_ROLLOUT_LAYOUT run on the lane-per-board
 *     kernels: one lane owns one board, connectivity by union-find in LDS (2.3x fewer
 *     vector instructions per ply, 4x the state): the layout for hexvec_rollout_*.
 *     On both, winner / done / reward / obs / legal mask are bit-exact; the reference's
 *     internal label planes are not kept (hexvec_get_state(labels != NULL) and
 *     hexvec_apply_moves return HEXVEC_E_MODE).
 *   - HEXVEC_F_NO_AUTO_RESET or HEXVEC_F_EXACT_LABELS handles run on the
 *     wave-per-board kernels, a literal restatement that also reproduces
 *     `regions`, illegal plies without state change and plies after the end.
 *
 * Random numbers (only when the library picks an action): counter based,
 *   key  = mix32(env_global * 0x9E3779B9 + seed_lo) ^ seed_hi
 *   draw = mix32(key + ctr * 0x9E3779B9), ctr++ per draw, per env
 *   mix32(x): x^=x>>16; x*=0x7feb352d; x^=x>>15; x*=0x846ca68b; x^=x>>16
 *   action = k-th empty cell (row-major, in the mover's view), k = mulhi32(draw, n_empty)
 * so results do not depend on how envs are sharded over GPUs (env_offset).
 */
#ifndef HEXVEC_H
#define HEXVEC_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct hexvec hexvec_t;

#define HEXVEC_MIN_BOARD 2
#define HEXVEC_MAX_BOARD 19        /* default (bitboard) layout of auto-reset handles              */
#define HEXVEC_MAX_BOARD_LABELS 14 /* NO_AUTO_RESET / EXACT_LABELS / ROLLOUT_LAYOUT handles         */

/* error codes */
#define HEXVEC_OK 0
#define HEXVEC_E_ARG (-1)   /* bad size / null handle / bad flag combination */
#define HEXVEC_E_HIP (-2)   /* a HIP runtime call failed                     */
#define HEXVEC_E_NOGPU (-3) /* no gfx950 device / device_id out of range     */
#define HEXVEC_E_MODE (-4)  /* entry point not valid for this handle's mode  */

/* create flags */
#define HEXVEC_F_ENC_B 0x1u         /* HexSingleGame.py encoding (else HexGame.py / hex-v0)      */
#define HEXVEC_F_AGENT_WHITE 0x2u   /* enc B self-play: every agent plays WHITE                  */
#define HEXVEC_F_AGENT_RANDOM 0x4u  /* enc B self-play: colour drawn once per env (SelfplayWrapper.py:72-73) */
#define HEXVEC_F_NO_AUTO_RESET 0x8u /* finished envs are not restarted inside step               */
#define HEXVEC_F_FIRST_WHITE 0x10u  /* enc A: constructor current_player_num = WHITE (HexGame.py:155) */
#define HEXVEC_F_TWO_SIDED 0x20u    /* no in-env opponent: step = one ply of whoever is to move  */
#define HEXVEC_F_EXACT_LABELS 0x40u /* keep the reference's label planes (regions) bit for bit; see below */
#define HEXVEC_F_ROLLOUT_LAYOUT 0x80u /* auto-reset handles: keep the state in the layout the device-resident
                                        rollouts run fastest on (one lane per board, union-find in LDS) instead
                                        of the default bitboards; every entry point works on either layout with
                                        identical results, only the speed differs (see below)                    */

/* Allocates the internal state for num_envs boards on device_id and resets
 * them (replaces HexEnv.__init__ HexGame.py:152-197 / HexSingleGame.py:163-196,
 * SelfPlayEnv.__init__ SelfplayWrapper.py:39-67).  env_offset = global index of
 * env 0 (multi-GPU sharding). */
int hexvec_create(hexvec_t **out, int board_size, int num_envs, int device_id, uint64_t seed,
                  uint32_t flags, uint32_t env_offset);
int hexvec_destroy(hexvec_t *h);

/* reset() of the envs whose env_mask byte is non-zero (NULL = all):
 * HexEnv.reset HexGame.py:206-242 (enc A: the opponent opens when FIRST_WHITE),
 * HexEnv.reset HexSingleGame.py:208-231 + SelfPlayEnv.reset SelfplayWrapper.py:69-89
 * (enc B: the opponent opens when the agent is WHITE).  opp_actions (int32 [E],
 * in the opponent's view) supplies those opening plies; NULL = uniform-legal RNG.
 * obs/legal are written for ALL envs. */
int hexvec_reset(hexvec_t *h, const uint8_t *env_mask, const int32_t *opp_actions, int8_t *obs,
                 uint8_t *legal, void *stream);

/* One env step per board.
 * self-play handles: SelfPlayEnv.step SelfplayWrapper.py:174-199 (enc B) /
 *   HexEnv.step HexGame.py:244-295 (enc A, agent BLACK): agent ply, opponent
 *   reply if the game goes on, agent-indexed reward, then auto reset.
 * TWO_SIDED handles: HexEnv.step HexSingleGame.py:233-263 (enc B) / one
 *   HexGame.make_move HexGame.py:85-111 (enc A): one ply, reward of the mover.
 * actions NULL = uniform-legal RNG agent; opp_actions NULL = RNG opponent.
 * opp_actions, when given, supply the opponent's REPLIES only: the opening of a board that is
 * reset inside this call (the opponent moves first on it) is always drawn by the library RNG --
 * a caller who wants to choose openings too uses the split step below.
 * terminal_obs (int8 [E, N*N], optional) receives the final board of envs that
 * finished in this call (untouched otherwise). */
int hexvec_step(hexvec_t *h, const int32_t *actions, const int32_t *opp_actions, int8_t *obs,
                uint8_t *legal, float *reward, uint8_t *done, int8_t *winner,
                int8_t *terminal_obs, void *stream);

/* Split form of hexvec_step for opponents that are policies on the device (auto-reset
 * self-play handles).  One env step =
 *   hexvec_step_agent(actions)            the agent's ply; boards whose game goes on now show
 *                                         the OPPONENT's view in obs/legal (what
 *                                         OpponentPolicy.choose_action is given,
 *                                         SelfplayWrapper.py:161) and get pending = 1;
 *   hexvec_step_opponent(opp_actions)     the opponent's ply on every pending board.  Boards
 *                                         that finish are reset; if the opponent opens on them
 *                                         (SelfplayWrapper.py:79-80, HexGame.py:224-230) they
 *                                         get pending = 2 and show the opponent's view again;
 *   hexvec_step_opponent(opp_actions)     ... repeated until no board is pending.
 * reward / done / winner / terminal_obs are written only for the boards whose step completes
 * in that call (hexvec_step_agent also writes done = 0 for the boards it leaves in progress, so
 * after the agent's call the done row is current for every board), obs / legal / pending
 * (uint8 [E]) for all boards.  actions / opp_actions may be
 * NULL (uniform-legal RNG).  With opp_actions == NULL throughout, the result (and the RNG stream) equals
 * hexvec_step with the same agent actions; with caller-supplied opp_actions the split step also takes the
 * OPENINGS from the caller (pending == 2 boards) and draws nothing, where hexvec_step draws them itself. */
int hexvec_step_agent(hexvec_t *h, const int32_t *actions, int8_t *obs, uint8_t *legal, float *reward,
                      uint8_t *done, int8_t *winner, int8_t *terminal_obs, uint8_t *pending,
                      void *stream);
int hexvec_step_opponent(hexvec_t *h, const int32_t *opp_actions, int8_t *obs, uint8_t *legal,
                         float *reward, uint8_t *done, int8_t *winner, int8_t *terminal_obs,
                         uint8_t *pending, void *stream);

/* Parity entry (TWO_SIDED handles): T literal plies from actions [T, E], never
 * resets, reproduces the reference on illegal plies and plies after the end.
 * enc A: HexGame.make_move HexGame.py:85-111; enc B: HexEnv.step HexSingleGame.py:233-263.
 * Outputs after each ply: ret [T,E], done [T,E] (simulator.done), r_black [T,E]
 * (reward[0], enc B), legal [T,E,N*N], obs [T,E,N*N]. */
int hexvec_apply_moves(hexvec_t *h, const int32_t *actions, int T, int8_t *ret_out,
                       uint8_t *done_out, int8_t *r_black_out, uint8_t *legal_out,
                       int8_t *obs_out, void *stream);

/* Device-resident random rollout (TWO_SIDED handles): T plies per board with
 * the state held in registers, both sides uniform-legal (minihex/__init__.py:8-12,
 * SelfplayWrapper.py:16-22), auto reset.  Recorded per ply: the obs and mask
 * the mover saw, its action, its reward, done.  All outputs optional. */
int hexvec_rollout_random(hexvec_t *h, int T, int8_t *obs /* [T,E,N*N] */, uint8_t *legal,
                          int32_t *actions /* [T,E] */, float *reward, uint8_t *done,
                          void *stream);

/* Self-play handles: T whole env steps per launch with an RNG agent and RNG
 * opponent (the random-policy throughput workload).  Recorded per step: the
 * agent's action, reward, done and the obs/mask returned by the step. */
int hexvec_rollout_selfplay(hexvec_t *h, int T, int8_t *obs /* [T,E,N*N] */, uint8_t *legal,
                            int32_t *actions /* [T,E] */, float *reward, uint8_t *done,
                            void *stream);

/* Internal state in the reference's own layout: board int8 [E,N*N]
 * (simulator.board), labels uint8 [E,2,(N+2)^2] (simulator.regions), engine
 * current_player_num uint8 [E]. */
int hexvec_get_state(hexvec_t *h, int8_t *board, uint8_t *labels, uint8_t *cur_player,
                     void *stream);

/* Replace the position of every env (exact handles only): HexGame.__init__ on a given
 * board, HexGame.py:21-68 / HexSingleGame.py:26-71.  board int8 [E,N*N] in the handle's
 * encoding (frame of BLACK to move); labels uint8 [E,2,(N+2)^2] are adopted like
 * `connected_stones` (:46-51) or, when NULL, rebuilt by replaying the stones row-major
 * (:53-61); cur_player = constructor current_player_num / active_player. */
int hexvec_set_state(hexvec_t *h, const int8_t *board, const uint8_t *labels, int cur_player,
                     void *stream);

/* Auto-reset handles: give the boards whose env_mask byte is non-zero (NULL = all) the start
 * position boards[e] (int8 [E,N*N], frame of BLACK to move, the handle's encoding): the reset
 * with `sample_board` of HexSingleGame.py:217-222, the stones replayed row-major as the
 * constructor does (:57-65).  Nobody has moved on them yet: on self-play handles where the
 * opponent moves first they get pending = 2 (see hexvec_step_opponent).  obs / legal / pending
 * are written for all boards. */
int hexvec_set_boards(hexvec_t *h, const uint8_t *env_mask, const int8_t *boards, int8_t *obs,
                      uint8_t *legal, uint8_t *pending, void *stream);

/* out4 (device, double[4]) = sum of episode returns, sum of episode lengths,
 * episodes finished, episodes the agent (TWO_SIDED: BLACK) won -- since create. */
int hexvec_episode_stats(hexvec_t *h, double *out4, void *stream);

/* agent colour per env, uint8 [E] (device) */
int hexvec_agent_colours(hexvec_t *h, uint8_t *out, void *stream);

/* Checkpoint / resume of the env state (the reference never saves env state; its
 * scripts checkpoint only the SB3 model).  The blob is opaque: boards, connectivity, players,
 * RNG counters, episode accumulators.  It restores into a handle created with the same
 * board_size / num_envs / flags.  dst / src are device buffers of hexvec_state_bytes(). */
size_t hexvec_state_bytes(const hexvec_t *h);
int hexvec_export_state(hexvec_t *h, void *dst, void *stream);
int hexvec_import_state(hexvec_t *h, const void *src, void *stream);

/* Blocks the calling thread until everything enqueued on `stream` (NULL = default stream) has
 * finished (hipStreamSynchronize on the handle's device): the one blocking call of the ABI, for
 * callers that hand the library HOST-PINNED output buffers (hipHostMalloc / torch pin_memory: the
 * kernels write them directly, no copy) and read them right after -- the single-board mirrors of
 * HexGame.make_move (HexGame.py:85-111) do exactly that, one launch + one wait per ply. */
int hexvec_stream_synchronize(hexvec_t *h, void *stream);

/* Completion ticket for callers with HOST-PINNED outputs (literal-engine handles only, HEXVEC_E_MODE otherwise):
 * once set, every step / apply / reset / set_state launch on the handle ends with each board's wavefront adding 1 to
 * *ticket (uint32 in host-pinned, GPU-visible memory; system-scope release behind a system fence, so all of that
 * board's outputs are visible first).  The host then spins on *ticket == previous + num_envs instead of paying a
 * stream wait: measured on one board, the result is visible 5.6 us after the launch call returns, the stream wait
 * returns 9.3 us later still.  hexvec_get_state does not signal.  NULL switches the ticket off. */
int hexvec_set_ticket(hexvec_t *h, uint32_t *ticket);

/* Batched opponent / agent action sampling on the device (no handle: a stateless utility).
 * One categorical sample per row from softmax(logits[r, :]) restricted to legal[r, c] != 0 -- what
 * OpponentPolicy.choose_action gets from model.predict(board, deterministic=False, action_masks=mask)
 * (SelfplayWrapper.py:26-35); logits == NULL = uniform over the legal cells, BaseRandomPolicy.choose_action
 * (SelfplayWrapper.py:16-24).  logits float32, row r at logits + r * logits_row_stride (elements; 0 = all boards
 * share one row, e.g. the opening on an empty board), or NULL; legal uint8 [rows, cols]; row_active uint8 [rows]
 * or NULL: boards whose byte is 0 get action 0 and are neither sampled nor counted (the boards that owe no
 * opponent ply in this round); actions int32 [rows]; all device pointers.  Random numbers: Gumbel-max with a
 * counter hash of (seed, salt, draw counter, row, column); state2 = two device uint64 words owned by the caller
 * and zeroed once {draw counter, reserved}.  advance != 0: the call advances the counter on the device (a
 * one-thread kernel behind the sampling kernel), so a captured graph draws fresh numbers on every replay;
 * several calls of one step may share a counter with different salts and advance it once, with the last.  A row that cannot be sampled (no legal cell, NaN logit on a legal cell) yields
 * action 0 and adds 1 to *bad_rows (device uint64, may be NULL).  Not the reference's Mersenne-Twister / torch
 * stream: the distribution is the same, the draws are not. */
int hexvec_sample_masked(const float *logits, long long logits_row_stride, const uint8_t *legal,
                         const uint8_t *row_active, int rows, int cols, uint64_t seed, uint32_t salt, int advance,
                         unsigned long long *state2, int32_t *actions, unsigned long long *bad_rows, void *stream);

/* SelfPlayEnv.setup_opponents (SelfplayWrapper.py:97-104) for a batch of boards, outside evaluation: a board that
 * starts a new episode gets opponent slot -1 (the best model) with probability prob_best, else a uniformly drawn
 * pool member 0..pool_size-1; the others keep theirs.  `when` names the point of the vectorised step
 * (hex_gym_env_amd/vec_selfplay.py) and with it which boards start a new episode:
 *   0  after the agent's ply: done[e] && pending[e] == 0 (ended, the agent moves first on the fresh board);
 *      also records ended[e] = done[e] != 0;
 *   1  before an opponent round: pending[e] == 2 (a fresh board on which the opponent is about to open,
 *      SelfplayWrapper.py:75, :79-80);
 *   2  after the opponent's replies: done[e] && !ended[e] && !white[e] (ended on the reply, agent BLACK).
 *   3  = 0 and 1 in one launch (after the agent's ply), 4 = 1 and 2 in one launch (after the replies): a board
 *      qualifies at one of the two points at most; the second point draws under salt + 1, as its own call would.
 * slot int64 [rows], the byte arrays uint8 [rows], all on the device; random numbers as in hexvec_sample_masked
 * (counter hash of seed, salt, draw counter, board; state2 = {draw counter, reserved}, advanced iff advance != 0). */
int hexvec_pick_opponents(int64_t *slot, int when, const uint8_t *done, const uint8_t *pending, uint8_t *ended,
                          const uint8_t *white, int rows, float prob_best, int pool_size, uint64_t seed,
                          uint32_t salt, int advance, unsigned long long *state2, void *stream);

/* SelfPlayEnv.setup_opponents and the opponents' openings INSIDE the split step (SelfplayWrapper.py:75, :79-80,
 * :97-104): after this call hexvec_step_agent / hexvec_step_opponent of the handle do, in the tail of their own
 * launch, what hexvec_pick_opponents does as a launch of its own (same boards, same draws):
 *   hexvec_step_agent     = ... + hexvec_pick_opponents(when = 3, salt 1)
 *   hexvec_step_opponent  = ... + hexvec_pick_opponents(when = 4, salt 4) + the counter moves on
 * and, if open_cdf != NULL, hexvec_step_opponent also PLAYS the opening of every board that ended on the reply and
 * on which the opponent moves first (pending would be 2), drawn from the picked opponent's opening distribution on
 * the empty board -- so one hexvec_step_agent + one hexvec_step_opponent complete a step and no board is left
 * pending.  Bookkeeping, all device memory owned by the caller and alive while attached:
 *   slot   int64 [E]   opponent slot per board (-1 = best model, k = pool member k), read and written;
 *   ended  uint8 [E]   scratch ("ended on the agent's ply");
 *   state2             the caller's draw counter {counter, reserved} (hexvec_sample_masked / hexvec_mlp_actions):
 *                      only WRITTEN here -- hexvec_step_opponent publishes the number of completed steps, so the
 *                      sampling calls of the next step draw under a new counter without a launch of their own; the
 *                      picks read the handle's own copy (part of the exported state);
 *   open_cdf float32 [n_policies, cells] or NULL: per policy, the cumulative distribution of its opening on the
 *                      empty board, non-decreasing, last entry exactly 1.0f (any other last entry marks a policy
 *                      that cannot be sampled: its boards open at cell 0 and are counted in *bad_rows); the opening is
 *                      the first cell whose entry exceeds u, u uniform in (0, 1) from the counter hash of
 *                      (seed, salt 6, steps + 1, board);
 *   policy_of_slot int32 [pool_size + 1] or NULL (= policy 0): the policy index of slot -1, 0, 1, ...;
 *   bad_rows uint64 or NULL.
 * slot == NULL detaches.  Auto-reset self-play handles on the default layout only (HEXVEC_E_MODE otherwise).
 * A host-side setting: nothing is enqueued; launches enqueued earlier keep what they were given. */
int hexvec_attach_pool(hexvec_t *h, int64_t *slot, uint8_t *ended, float prob_best, int pool_size, uint64_t seed,
                       unsigned long long *state2, const float *open_cdf, const int32_t *policy_of_slot,
                       int n_policies, unsigned long long *bad_rows);

/* The opponents' (or the agent's) actor network evaluated, masked and sampled for a batch of boards in one call:
 * OpponentPolicy.choose_action -> model.predict(board, deterministic, action_masks=mask) (SelfplayWrapper.py:26-35)
 * for SB3's default MlpPolicy actor  flatten -> float -> Linear(cells, h1)-Tanh-Linear(h1, h2)-Tanh-Linear(h2, cells)
 * -> logits -> masked categorical sample (or arg-max if deterministic != 0).  fp32 throughout.
 *
 * weights: n_policies networks, network p at weights + p * hexvec_mlp_weights_floats(cells, h1, h2) floats, each in the
 * library's PACKED form written by hexvec_mlp_pack from the six row-major tensors of the three Linear layers
 * (w [out][in], b [out]: torch's layout; device pointers; one small kernel).  The packed form is opaque -- with
 * h1 == h2 == 64 (SB3's default, what the reference's scripts train) it is the operand order of the fp32 MFMA
 * chains, otherwise transposed rows for the scalar-broadcast kernel; h1, h2 multiples of 16 in [16, 256]; the
 * table's base pointer 64-byte aligned.  Re-pack a network whenever its parameters change.
 * Which network plays a board: slot == NULL -> network 0 for every board.  Otherwise board r uses
 *     policy_of_slot[clamp(slot[r] + slot_bias, 0, n_slots - 1)]     (policy_of_slot == NULL: that index itself),
 * i.e. with slot_bias = 1 the vectorised SelfPlayEnv's slot array (-1 = best model, k = pool member k,
 * SelfplayWrapper.py:97-104) indexes a table whose entry 0 is the best model; several slots may name one network
 * (the reference's pool mostly holds the same live model).  With more than one network the boards are grouped by
 * network on the device first (counting sort; `scratch` = hexvec_mlp_scratch_bytes(rows, n_policies) bytes of device
 * memory, contents irrelevant); each workgroup then serves 64 boards of one network.
 * obs int8 [rows, cells] (the mover's view, as the step calls emit it), legal uint8 [rows, cells], row_active uint8
 * [rows] or NULL (boards with 0 get action 0 and are skipped), actions int32 [rows], logits_out float32 [rows, cells] or
 * NULL (the raw logits of the boards that were evaluated; other rows untouched).  Random numbers, state2, salt, advance
 * and bad_rows exactly as in hexvec_sample_masked: the call returns the actions hexvec_sample_masked would return on
 * logits_out.  All pointers are device pointers. */
size_t hexvec_mlp_weights_floats(int cells, int h1, int h2);
size_t hexvec_mlp_scratch_bytes(int rows, int n_policies);
/* LDS bytes per workgroup the call needs for this shape (0: shape not served): a caller falls back to its own forward
 * pass when that exceeds what the GPU gives one workgroup (160 KB on MI355X; hexvec_mlp_actions refuses it otherwise).
 * The handle-less entry points (hexvec_sample_masked, hexvec_pick_opponents, hexvec_mlp_*) run on the GPU that owns
 * their output buffer, whatever the calling thread's current device is. */
size_t hexvec_mlp_lds_bytes(int cells, int h1, int h2);
int hexvec_mlp_pack(const float *w1, const float *b1, const float *w2, const float *b2, const float *w3, const float *b3,
                    int cells, int h1, int h2, float *dst, void *stream);
int hexvec_mlp_actions(const int8_t *obs, const uint8_t *legal, const uint8_t *row_active, const int64_t *slot,
                       int slot_bias, const int32_t *policy_of_slot, int n_slots, int rows, int cells,
                       const float *weights, int n_policies, int h1, int h2, int deterministic, uint64_t seed,
                       uint32_t salt, int advance, unsigned long long *state2, int32_t *actions, float *logits_out,
                       unsigned long long *bad_rows, void *scratch, size_t scratch_bytes, void *stream);

int hexvec_board_size(const hexvec_t *h);
int hexvec_num_envs(const hexvec_t *h);
uint32_t hexvec_flags(const hexvec_t *h);

/* name of the kernel the last call of this thread launched (for profiling) */
const char *hexvec_last_kernel(void);
const char *hexvec_last_error(void);

#ifdef __cplusplus
}
#endif
#endif
