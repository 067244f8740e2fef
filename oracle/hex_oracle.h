/*
 * hex_oracle.h -- CPU ORACLE (test infrastructure, NOT the product).
 *
 * A plain-C restatement of the reference's Hex engine / env / self-play
 * wrapper, function by function, each citing the reference file:line it
 * follows (paths relative to the reference root, minihex/...).  It is pinned
 * against golden fixtures generated from the live reference
 * (tests/golden/make_fixtures.py -> the .npz files under tests/golden; tests/test_oracle_golden.py).
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * load this library.  The product (hex_gym_env_amd + libhexvec.so) never does.
 *
 * Two layers:
 *   1. hexo_game / hexo_envb / hexo_selfplay / hexo_enva : literal, per-env
 *      restatements of the reference classes (arrays laid out as numpy does).
 *   2. hexo_vec_* : the vectorised-env contract of include/hexvec.h (auto
 *      reset, counter RNG, episode statistics) expressed with layer 1, so the
 *      HIP library's whole output tensors can be compared element-wise.
 */
#ifndef HEX_ORACLE_H
#define HEX_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define HEXO_NMAX 19
#define HEXO_PMAX ((HEXO_NMAX + 2) * (HEXO_NMAX + 2))

/* encodings: A = HexGame.py:10-13, B = HexSingleGame.py:15-19 */
#define HEXO_ENC_A 0 /* board values BLACK=0 WHITE=1 EMPTY=2, absolute coordinates  */
#define HEXO_ENC_B 1 /* board values -1/0/+1, stored in the mover's perspective       */

#define HEXO_RET_NONE (-1) /* make_move returned None */
#define HEXO_RET_ILLEGAL 3 /* make_move returned 3 (HexGame.py:91-92)        */

typedef struct {
    int n;
    int enc;
    int8_t board[HEXO_NMAX * HEXO_NMAX]; /* simulator.board, row-major          */
    int32_t regions[2][HEXO_PMAX];       /* simulator.regions [2][N+2][N+2]     */
    int32_t region_counter[2];
    int cur;    /* current_player_num */
    int done;   /* .done   */
    int winner; /* .winner : -1 None, 0, 1 */
    int empty_fields;
} hexo_game;

/* ---- layer 1: engine (HexGame.py:16-142, HexSingleGame.py:21-153) ---- */
void hexo_game_init(hexo_game *g, int n, int enc, int cur, const int8_t *board_or_null);
int hexo_is_valid_move(const hexo_game *g, int action);
int hexo_make_move(hexo_game *g, int action); /* -> HEXO_RET_NONE / 0 / 1 / 3 */
void hexo_flood_fill(hexo_game *g, int y, int x);
void hexo_legal_mask(const hexo_game *g, uint8_t *mask /* [n*n] */);
void hexo_invert_board(hexo_game *g);

/* ---- env B: HexSingleGame.py:156-271 ---- */
typedef struct {
    hexo_game sim;
    int cur_env; /* HexEnv.current_player_num */
    int winner;  /* HexEnv.winner : -1 / 0 / 1 / 3 */
} hexo_envb;
void hexo_envb_reset(hexo_envb *e, int n);
/* returns done; *r_black = reward[0] (reward[1] == -reward[0]) */
int hexo_envb_step(hexo_envb *e, int action, int *r_black);

/* ---- SelfPlayEnv: SelfplayWrapper.py:37-208 (opponent actions supplied) ---- */
typedef struct {
    hexo_envb env;
    int agent; /* agent_player_num */
} hexo_selfplay;
/* opp_action is consumed only if the agent is WHITE (opponent opens, :79-80) */
void hexo_selfplay_reset(hexo_selfplay *s, int n, int agent, int opp_action);
/* returns done; opp_action consumed only if the agent's ply did not end the game */
int hexo_selfplay_step(hexo_selfplay *s, int action, int opp_action, int *agent_reward, int *opp_used);

/* ---- env A "hex-v0": HexGame.py:145-359, player_color = BLACK only (q7) ---- */
typedef struct {
    hexo_game sim;
    int first;  /* constructor current_player_num */
    int winner; /* HexEnv.winner */
} hexo_enva;
/* opponent actions are in the OPPONENT'S view (inverted board, HexGame.py:333-342) */
void hexo_enva_reset(hexo_enva *e, int n, int first, int opp_action, int *last_move_opponent);
int hexo_enva_step(hexo_enva *e, int action, int opp_action, int *reward, int *last_move_opponent,
                   int *opp_used);
/* the board the opponent policy is shown: transpose + colour swap (HexGame.py:297-303) */
void hexo_enva_opponent_view(const hexo_game *g, int8_t *view);

/* ---- counter RNG + uniform-legal policy (the build's own spec, include/hexvec.h) ---- */
uint32_t hexo_rng_key(uint64_t seed, uint32_t env_global);
uint32_t hexo_rng_draw(uint32_t key, uint32_t ctr);
/* k-th empty cell (row-major, in the given view) with k = mulhi(r, n_empty); -1 if none */
int hexo_uniform_legal(const int8_t *view, int ncells, int empty_code, uint32_t r);

/* ---- layer 2: vectorised contract ---- */
typedef struct hexo_vec hexo_vec;

#define HEXO_F_ENC_B 0x1u
#define HEXO_F_AGENT_WHITE 0x2u      /* all agents WHITE (else BLACK unless RANDOM)   */
#define HEXO_F_AGENT_RANDOM 0x4u     /* per-env colour drawn once from the RNG key    */
#define HEXO_F_NO_AUTO_RESET 0x8u
#define HEXO_F_FIRST_WHITE 0x10u     /* enc A only: WHITE (the opponent) opens        */
#define HEXO_F_TWO_SIDED 0x20u       /* no in-env opponent: reset/step are the plain env */

hexo_vec *hexo_vec_create(int n, int num_envs, uint64_t seed, uint32_t flags, uint32_t env_offset);
void hexo_vec_destroy(hexo_vec *v);
/* mask NULL = all envs.  opp_actions NULL = RNG opponent. */
void hexo_vec_reset(hexo_vec *v, const uint8_t *env_mask, const int32_t *opp_actions, int8_t *obs,
                    uint8_t *legal);
/* W1: agent ply (+ opponent reply) (+ auto reset).  actions NULL = RNG agent. */
void hexo_vec_step(hexo_vec *v, const int32_t *actions, const int32_t *opp_actions, int8_t *obs,
                   uint8_t *legal, float *reward, uint8_t *done, int8_t *winner,
                   int8_t *terminal_obs);
/* bench helper: K RNG-agent steps per env inside one call (outputs overwritten) */
void hexo_vec_step_repeat(hexo_vec *v, int K, int8_t *obs, uint8_t *legal, float *reward,
                          uint8_t *done);
/* literal plies, no reset: enc A = HexGame.make_move, enc B = HexEnv.step */
void hexo_vec_apply_moves(hexo_vec *v, const int32_t *actions, int T, int8_t *ret_out,
                          uint8_t *done_out, int8_t *r_black_out, uint8_t *legal_out,
                          int8_t *obs_out);
/* W2: T plies, both sides RNG, every ply recorded (obs/mask BEFORE the ply) */
void hexo_vec_rollout(hexo_vec *v, int T, int8_t *obs, uint8_t *legal, int32_t *actions,
                      float *reward, uint8_t *done);
/* W1: T whole env steps (hexo_vec_step with the RNG agent and opponent), every step recorded: the agent's
 * action, reward, done, obs/mask AFTER the step */
void hexo_vec_rollout_selfplay(hexo_vec *v, int T, int8_t *obs, uint8_t *legal, int32_t *actions,
                               float *reward, uint8_t *done);
void hexo_vec_get_state(hexo_vec *v, int8_t *board, uint8_t *labels, uint8_t *cur_player);
void hexo_vec_episode_stats(hexo_vec *v, double *out4);
void hexo_vec_agent_colours(hexo_vec *v, uint8_t *out);
/* threads > 1 splits the env range over pthreads for the CPU baseline */
void hexo_vec_set_threads(hexo_vec *v, int threads);

#ifdef __cplusplus
}
#endif
#endif
