/*
 * hex_oracle.c -- CPU ORACLE (test infrastructure, NOT the product).
 * See hex_oracle.h.  Pinned by tests/test_oracle_golden.py against fixtures
 * generated from the live reference (tests/golden/make_fixtures.py).
 *
 * Every function below follows the cited reference lines statement by
 * statement; arrays are laid out exactly as the reference's numpy arrays
 * (board [N][N] row-major, regions [2][N+2][N+2]).
 */
#include "hex_oracle.h"

#include <pthread.h>
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------ */
/* layer 1: engine                                                    */
/* ------------------------------------------------------------------ */

static inline int empty_code(int enc) { return enc == HEXO_ENC_A ? 2 : 0; }

/* HexGame.py:124-142 == HexSingleGame.py:135-153 */
void hexo_flood_fill(hexo_game *g, int py, int px)
{
    const int W = g->n + 2;
    int32_t *regions = g->regions[g->cur];
    const int y = py + 1, x = px + 1;
    /* neighbourhood = regions[y-1:y+2, x-1:x+2] with [0,0] and [2,2] zeroed:
     * the six hex neighbours plus the centre (:127-129 / :138-140) */
    const int dy[7] = {-1, -1, 0, 0, 0, 1, 1};
    const int dx[7] = {0, 1, -1, 0, 1, -1, 0};
    int32_t adj[7];
    int na = 0;
    for (int i = 0; i < 7; ++i) {
        int32_t l = regions[(y + dy[i]) * W + (x + dx[i])];
        /* sorted(set(...)) with the 0 label popped (:130-133): keep distinct
         * non-zero labels.  A centre label that is non-zero (cell filled
         * twice, only possible through misuse) stays in the set, as it does
         * in the reference. */
        if (l == 0) continue;
        int dup = 0;
        for (int j = 0; j < na; ++j) dup |= (adj[j] == l);
        if (!dup) adj[na++] = l;
    }
    if (na == 0) { /* :135-137 */
        regions[y * W + x] = g->region_counter[g->cur];
        g->region_counter[g->cur] += 1;
        return;
    }
    int32_t m = adj[0]; /* :139 smallest label */
    for (int j = 1; j < na; ++j)
        if (adj[j] < m) m = adj[j];
    regions[y * W + x] = m; /* :140 */
    /* :141-142 regions[regions == label] = new_region_label for the others.
     * Labels are distinct and all > m, so the order of the rewrites is
     * immaterial. */
    for (int j = 0; j < na; ++j) {
        if (adj[j] == m) continue;
        for (int c = 0; c < W * W; ++c)
            if (regions[c] == adj[j]) regions[c] = m;
    }
}

/* HexGame.py:21-68 / HexSingleGame.py:26-71 */
void hexo_game_init(hexo_game *g, int n, int enc, int cur, const int8_t *board)
{
    const int W = n + 2;
    memset(g, 0, sizeof(*g));
    g->n = n;
    g->enc = enc;
    const int8_t E = (int8_t)empty_code(enc);
    for (int i = 0; i < n * n; ++i) g->board[i] = board ? board[i] : E;
    g->empty_fields = 0; /* :25 / :29 */
    for (int i = 0; i < n * n; ++i) g->empty_fields += (g->board[i] == E);
    /* :38-45 / :42-49  (assignment order matters at the corners) */
    for (int r = 0; r < W; ++r) g->regions[1][r * W + 0] = 1;       /* WHITE [:,0] = 1    */
    for (int c = 0; c < W; ++c) g->regions[0][0 * W + c] = 1;       /* BLACK [0,:] = 1    */
    for (int r = 0; r < W; ++r) g->regions[1][r * W + (n + 1)] = 2; /* WHITE [:,N+1] = 2  */
    for (int c = 0; c < W; ++c) g->regions[0][(n + 1) * W + c] = 2; /* BLACK [N+1,:] = 2  */
    g->region_counter[0] = 3; /* max + 1, :49-51 */
    g->region_counter[1] = 3;
    /* replay pre-placed stones row-major (:53-61 / :57-65) */
    const int black = enc == HEXO_ENC_A ? 0 : -1;
    const int white = 1;
    for (int y = 0; y < n; ++y)
        for (int x = 0; x < n; ++x) {
            int v = g->board[y * n + x];
            if (v == black) {
                g->cur = 0;
                hexo_flood_fill(g, y, x);
            } else if (v == white) {
                g->cur = 1;
                hexo_flood_fill(g, y, x); /* untransposed in both engines (:63-65) */
            }
        }
    g->cur = cur;
    g->done = 0;
    g->winner = HEXO_RET_NONE;
}

/* HexGame.py:74-76 / HexSingleGame.py:77-79 (+ action_to_coordinate :116-119) */
int hexo_is_valid_move(const hexo_game *g, int action)
{
    int y = action / g->n;
    int x = action - g->n * y;
    return g->board[y * g->n + x] == empty_code(g->enc);
}

/* fast_move: HexGame.py:85-111 (A), HexSingleGame.py:88-122 (B) */
int hexo_make_move(hexo_game *g, int action)
{
    if (!hexo_is_valid_move(g, action)) return HEXO_RET_ILLEGAL; /* :91-92 / :95-96 */
    const int n = g->n, W = n + 2;
    int y = action / n;
    int x = action - n * y;
    int winner = HEXO_RET_NONE;
    if (g->enc == HEXO_ENC_A) {
        g->board[y * n + x] = (int8_t)g->cur; /* :95 */
        g->empty_fields -= x;                 /* :96 (sic) */
        hexo_flood_fill(g, y, x);             /* :98 */
        if (g->regions[g->cur][W * W - 1] == 1) { /* :101-105 */
            g->done = 1;
            winner = g->cur;
            g->winner = winner;
        }
    } else {
        g->board[y * n + x] = -1; /* :100 always the BLACK encoding */
        g->empty_fields -= 1;     /* :101 */
        if (g->cur == 1)
            hexo_flood_fill(g, x, y); /* :103-104 switch */
        else
            hexo_flood_fill(g, y, x);
        if (g->regions[g->cur][W * W - 1] == 1) { /* :110-116 */
            g->done = 1;
            winner = g->cur;
            g->winner = winner;
        } else if (g->empty_fields <= 0) { /* :117-119 */
            g->done = 1;
            winner = HEXO_RET_NONE;
        }
    }
    g->cur = (g->cur + 1) % 2; /* :110 / :121 */
    return winner;
}

/* HexGame.py:203-204 / HexSingleGame.py:205-206 */
void hexo_legal_mask(const hexo_game *g, uint8_t *mask)
{
    for (int a = 0; a < g->n * g->n; ++a) mask[a] = (uint8_t)hexo_is_valid_move(g, a);
}

/* HexGame.py:297-303 (A) / HexSingleGame.py:265-271 (B): transpose + colour swap */
void hexo_invert_board(hexo_game *g)
{
    const int n = g->n;
    int8_t t[HEXO_NMAX * HEXO_NMAX];
    const int8_t black = g->enc == HEXO_ENC_A ? 0 : -1, white = 1;
    for (int y = 0; y < n; ++y)
        for (int x = 0; x < n; ++x) {
            int8_t v = g->board[x * n + y]; /* board.T */
            t[y * n + x] = v == black ? white : (v == white ? black : v);
        }
    memcpy(g->board, t, (size_t)(n * n));
}

/* ------------------------------------------------------------------ */
/* env B  (HexSingleGame.py:156-271)                                  */
/* ------------------------------------------------------------------ */

/* :208-231 (sample_board = False; cached regions == fresh regions) */
void hexo_envb_reset(hexo_envb *e, int n)
{
    e->cur_env = 0; /* :209 */
    hexo_game_init(&e->sim, n, HEXO_ENC_B, e->cur_env, NULL);
    /* e->winner is NOT touched by reset() in the reference */
}

/* :233-263 */
int hexo_envb_step(hexo_envb *e, int action, int *r_black)
{
    e->winner = hexo_make_move(&e->sim, action); /* :239 */
    if (e->winner == HEXO_RET_ILLEGAL) e->sim.done = 1; /* :240-242 (reward overwritten) */
    int r;
    if (e->winner == e->cur_env) /* :245-250 ; None never equals an int */
        r = 1;
    else if (e->winner == (e->cur_env + 1) % 2)
        r = -1;
    else
        r = 0;
    int reward[2] = {-r, -r}; /* :252-253 */
    reward[e->cur_env] = r;
    e->cur_env = (e->cur_env + 1) % 2; /* :259 */
    hexo_invert_board(&e->sim);        /* :260 always */
    *r_black = reward[0];
    return e->sim.done;
}

/* ------------------------------------------------------------------ */
/* SelfPlayEnv (SelfplayWrapper.py:37-208), play_gui = False           */
/* ------------------------------------------------------------------ */

/* :69-89 ; agent_player_num given (the :72-73 draw happens once, outside) */
void hexo_selfplay_reset(hexo_selfplay *s, int n, int agent, int opp_action)
{
    hexo_envb_reset(&s->env, n);
    s->agent = agent;
    if (s->env.cur_env != s->agent) { /* :79-80 continue_game() */
        int rb;
        hexo_envb_step(&s->env, opp_action, &rb); /* :161, :170 */
    }
}

/* :174-199 */
int hexo_selfplay_step(hexo_selfplay *s, int action, int opp_action, int *agent_reward,
                       int *opp_used)
{
    int rb;
    int done = hexo_envb_step(&s->env, action, &rb); /* :176 */
    *opp_used = 0;
    if (!done) { /* :187-190 */
        done = hexo_envb_step(&s->env, opp_action, &rb);
        *opp_used = 1;
    }
    *agent_reward = s->agent == 0 ? rb : -rb; /* :192 reward[agent_player_num] */
    return done;
}

/* ------------------------------------------------------------------ */
/* env A "hex-v0" (HexGame.py:145-359), player_color = BLACK           */
/* ------------------------------------------------------------------ */

void hexo_enva_opponent_view(const hexo_game *g, int8_t *view)
{
    hexo_game t = *g;
    hexo_invert_board(&t); /* :333-334 */
    memcpy(view, t.board, (size_t)(g->n * g->n));
}

/* opponent_move, :332-349 : action given in the inverted view, transposed back (:341-342) */
static int enva_opponent_move(hexo_enva *e, int opp_action_view)
{
    const int n = e->sim.n;
    int y = opp_action_view / n, x = opp_action_view - n * y;
    int a = x * n + y;
    e->winner = hexo_make_move(&e->sim, a); /* :346 */
    return a;
}

/* :206-242 */
void hexo_enva_reset(hexo_enva *e, int n, int first, int opp_action, int *last_move_opponent)
{
    e->first = first;
    hexo_game_init(&e->sim, n, HEXO_ENC_A, first, NULL);
    *last_move_opponent = -1;
    if (0 /* player BLACK */ != first) /* :224-230 */
        *last_move_opponent = enva_opponent_move(e, opp_action);
    /* HexEnv.winner is not cleared by reset() unless the opponent moved */
}

/* :244-295 */
int hexo_enva_step(hexo_enva *e, int action, int opp_action, int *reward, int *last_move_opponent,
                   int *opp_used)
{
    *opp_used = 0;
    *last_move_opponent = -1;
    if (!e->sim.done) { /* :250-253 */
        e->winner = hexo_make_move(&e->sim, action);
        if (e->winner == HEXO_RET_ILLEGAL) e->sim.done = 1;
    }
    if (!e->sim.done) { /* :259-265 */
        *last_move_opponent = enva_opponent_move(e, opp_action);
        *opp_used = 1;
    }
    if (e->winner == 0) /* :267-279 ; player == BLACK == 0 */
        *reward = 1;
    else if (e->winner == 1)
        *reward = -1;
    else if (e->winner == HEXO_RET_ILLEGAL)
        *reward = -100;
    else
        *reward = 0;
    return e->sim.done;
}

/* ------------------------------------------------------------------ */
/* counter RNG + uniform-legal policy (build spec, include/hexvec.h)   */
/* ------------------------------------------------------------------ */

static inline uint32_t mix32(uint32_t x)
{
    x ^= x >> 16;
    x *= 0x7feb352dU;
    x ^= x >> 15;
    x *= 0x846ca68bU;
    x ^= x >> 16;
    return x;
}

uint32_t hexo_rng_key(uint64_t seed, uint32_t env_global)
{
    return mix32(env_global * 0x9E3779B9U + (uint32_t)seed) ^ (uint32_t)(seed >> 32);
}

uint32_t hexo_rng_draw(uint32_t key, uint32_t ctr) { return mix32(key + ctr * 0x9E3779B9U); }

/* reference policy shape: SelfplayWrapper.py:16-22 / minihex/__init__.py:8-12
 * (valid = cells == EMPTY in row-major order; choice = floor(u * len)); here
 * u*len is the integer mulhi(r, len). */
int hexo_uniform_legal(const int8_t *view, int ncells, int ecode, uint32_t r)
{
    int cnt = 0;
    for (int i = 0; i < ncells; ++i) cnt += (view[i] == ecode);
    if (cnt == 0) return -1;
    int k = (int)(((uint64_t)r * (uint64_t)cnt) >> 32);
    for (int i = 0; i < ncells; ++i)
        if (view[i] == ecode && k-- == 0) return i;
    return -1;
}

/* ------------------------------------------------------------------ */
/* layer 2: vectorised contract (mirrors include/hexvec.h)             */
/* ------------------------------------------------------------------ */

typedef struct {
    union {
        hexo_selfplay sp; /* enc B (sp.env is also the two-sided env) */
        hexo_enva ea;     /* enc A */
    } u;
    uint32_t key, ctr;
    int agent;
    double ep_ret;
    int ep_len;
    double s_ret, s_len, s_eps, s_wins;
} venv;

struct hexo_vec {
    int n, E, enc, threads;
    uint32_t flags;
    venv *envs;
};

static hexo_game *sim_of(hexo_vec *v, venv *e)
{
    return v->enc == HEXO_ENC_B ? &e->u.sp.env.sim : &e->u.ea.sim;
}

static uint32_t draw(venv *e) { return hexo_rng_draw(e->key, e->ctr++); }

/* uniform-legal action for whoever is to move, in the view that mover is shown */
static int rng_action_stored(hexo_vec *v, venv *e)
{
    hexo_game *g = sim_of(v, e);
    int a = hexo_uniform_legal(g->board, v->n * v->n, empty_code(v->enc), draw(e));
    return a < 0 ? 0 : a;
}

static int rng_action_enva_opponent(hexo_vec *v, venv *e)
{
    int8_t view[HEXO_NMAX * HEXO_NMAX];
    hexo_enva_opponent_view(&e->u.ea.sim, view);
    int a = hexo_uniform_legal(view, v->n * v->n, 2, draw(e));
    return a < 0 ? 0 : a;
}

static void reset_one(hexo_vec *v, venv *e, const int32_t *opp_action)
{
    const int two = (v->flags & HEXO_F_TWO_SIDED) != 0;
    e->ep_ret = 0;
    e->ep_len = 0;
    if (v->enc == HEXO_ENC_B) {
        hexo_selfplay *s = &e->u.sp;
        hexo_envb_reset(&s->env, v->n); /* SelfplayWrapper.py:70 */
        s->agent = e->agent;
        if (!two && s->env.cur_env != s->agent) { /* :79-80 the opponent opens */
            int oa = opp_action ? *opp_action : rng_action_stored(v, e);
            int rb;
            hexo_envb_step(&s->env, oa, &rb);
        }
    } else {
        hexo_enva *ea = &e->u.ea;
        const int first = (v->flags & HEXO_F_FIRST_WHITE) ? 1 : 0;
        ea->first = first;
        hexo_game_init(&ea->sim, v->n, HEXO_ENC_A, first, NULL); /* HexGame.py:207-220 */
        if (!two && first != 0) { /* :224-230 the opponent opens */
            int oa = opp_action ? *opp_action : rng_action_enva_opponent(v, e);
            enva_opponent_move(ea, oa);
        }
    }
}

static void write_obs(hexo_vec *v, venv *e, int8_t *obs, uint8_t *legal)
{
    hexo_game *g = sim_of(v, e);
    if (obs) memcpy(obs, g->board, (size_t)(v->n * v->n));
    if (legal) hexo_legal_mask(g, legal);
}

hexo_vec *hexo_vec_create(int n, int num_envs, uint64_t seed, uint32_t flags, uint32_t env_offset)
{
    if (n < 2 || n > HEXO_NMAX || num_envs < 1) return NULL;
    hexo_vec *v = (hexo_vec *)calloc(1, sizeof(*v));
    v->n = n;
    v->E = num_envs;
    v->flags = flags;
    v->enc = (flags & HEXO_F_ENC_B) ? HEXO_ENC_B : HEXO_ENC_A;
    v->threads = 1;
    v->envs = (venv *)calloc((size_t)num_envs, sizeof(venv));
    for (int i = 0; i < num_envs; ++i) {
        venv *e = &v->envs[i];
        e->key = hexo_rng_key(seed, env_offset + (uint32_t)i);
        e->ctr = 0;
        if (v->enc == HEXO_ENC_A)
            e->agent = 0;
        else if (flags & HEXO_F_AGENT_RANDOM)
            e->agent = (int)(mix32(e->key ^ 0xA5A5A5A5U) & 1U);
        else
            e->agent = (flags & HEXO_F_AGENT_WHITE) ? 1 : 0;
        e->u.sp.env.winner = HEXO_RET_NONE;
        reset_one(v, e, NULL);
    }
    return v;
}

void hexo_vec_destroy(hexo_vec *v)
{
    if (!v) return;
    free(v->envs);
    free(v);
}

void hexo_vec_set_threads(hexo_vec *v, int threads) { v->threads = threads < 1 ? 1 : threads; }

/* ---- tiny parallel-for over env ranges ---- */
typedef struct {
    void (*fn)(hexo_vec *, int, int, void *);
    hexo_vec *v;
    int lo, hi;
    void *arg;
} pf_task;

static void *pf_tramp(void *p)
{
    pf_task *t = (pf_task *)p;
    t->fn(t->v, t->lo, t->hi, t->arg);
    return NULL;
}

static void parallel_for(hexo_vec *v, void (*fn)(hexo_vec *, int, int, void *), void *arg)
{
    int T = v->threads;
    if (T > v->E) T = v->E;
    if (T <= 1) {
        fn(v, 0, v->E, arg);
        return;
    }
    pthread_t th[256];
    pf_task tk[256];
    if (T > 256) T = 256;
    for (int i = 0; i < T; ++i) {
        tk[i].fn = fn;
        tk[i].v = v;
        tk[i].lo = (int)((long long)v->E * i / T);
        tk[i].hi = (int)((long long)v->E * (i + 1) / T);
        tk[i].arg = arg;
        pthread_create(&th[i], NULL, pf_tramp, &tk[i]);
    }
    for (int i = 0; i < T; ++i) pthread_join(th[i], NULL);
}

/* ---- reset ---- */
typedef struct {
    const uint8_t *mask;
    const int32_t *opp;
    int8_t *obs;
    uint8_t *legal;
} reset_args;

static void reset_range(hexo_vec *v, int lo, int hi, void *p)
{
    reset_args *a = (reset_args *)p;
    const int C = v->n * v->n;
    for (int i = lo; i < hi; ++i) {
        venv *e = &v->envs[i];
        if (!a->mask || a->mask[i]) reset_one(v, e, a->opp ? &a->opp[i] : NULL);
        write_obs(v, e, a->obs ? a->obs + (size_t)i * C : NULL,
                  a->legal ? a->legal + (size_t)i * C : NULL);
    }
}

void hexo_vec_reset(hexo_vec *v, const uint8_t *env_mask, const int32_t *opp_actions, int8_t *obs,
                    uint8_t *legal)
{
    reset_args a = {env_mask, opp_actions, obs, legal};
    parallel_for(v, reset_range, &a);
}

/* ---- one W1 step (selfplay) or one ply (two-sided) for env e ---- */
typedef struct {
    const int32_t *actions, *opp;
    int8_t *obs;
    uint8_t *legal;
    float *reward;
    uint8_t *done;
    int8_t *winner;
    int8_t *tobs;
    int repeat; /* bench: run this many steps per env */
} step_args;

static void finish_episode(hexo_vec *v, venv *e, int win)
{
    e->s_ret += e->ep_ret;
    e->s_len += e->ep_len;
    e->s_eps += 1;
    e->s_wins += win;
    (void)v;
}

static void step_one(hexo_vec *v, venv *e, int i, step_args *a)
{
    const int C = v->n * v->n;
    const int two = (v->flags & HEXO_F_TWO_SIDED) != 0;
    int done, rew, win_code, agent_won;
    const int act = a->actions ? a->actions[i] : rng_action_stored(v, e);
    if (v->enc == HEXO_ENC_B) {
        hexo_selfplay *s = &e->u.sp;
        int rb;
        if (two) { /* one HexEnv.step (HexSingleGame.py:233-263) by whoever is to move */
            const int mover = s->env.cur_env;
            done = hexo_envb_step(&s->env, act, &rb);
            rew = mover == 0 ? rb : -rb; /* reward[mover] */
            e->ep_ret += rb;             /* two-sided statistics are BLACK's */
            agent_won = done && s->env.winner == 0;
        } else { /* SelfPlayEnv.step, SelfplayWrapper.py:174-199 */
            done = hexo_envb_step(&s->env, act, &rb); /* :176 */
            if (!done) {                               /* :187-190 */
                int oa = a->opp ? a->opp[i] : rng_action_stored(v, e);
                done = hexo_envb_step(&s->env, oa, &rb);
            }
            rew = s->agent == 0 ? rb : -rb; /* :192 */
            e->ep_ret += rew;
            agent_won = done && rew > 0;
        }
        win_code = s->env.winner;
    } else {
        hexo_enva *ea = &e->u.ea;
        if (two) { /* engine plies (HexGame.py:85-111) with env-A conventions on illegal */
            const int mover = ea->sim.cur;
            const int w = hexo_make_move(&ea->sim, act);
            ea->winner = w;
            if (w == HEXO_RET_ILLEGAL) ea->sim.done = 1;
            rew = w == mover ? 1 : (w == HEXO_RET_ILLEGAL ? -100 : 0);
            done = ea->sim.done;
            e->ep_ret += (w == 0) ? 1 : ((w == 1) ? -1 : 0);
            agent_won = done && w == 0;
        } else { /* HexEnv.step, HexGame.py:244-295, player BLACK */
            if (!ea->sim.done) { /* :250-253 */
                ea->winner = hexo_make_move(&ea->sim, act);
                if (ea->winner == HEXO_RET_ILLEGAL) ea->sim.done = 1;
            }
            if (!ea->sim.done) { /* :259-265 */
                int oa = a->opp ? a->opp[i] : rng_action_enva_opponent(v, e);
                enva_opponent_move(ea, oa);
            }
            rew = ea->winner == 0 ? 1 /* :267-279 */
                  : (ea->winner == 1 ? -1 : (ea->winner == HEXO_RET_ILLEGAL ? -100 : 0));
            done = ea->sim.done;
            e->ep_ret += rew;
            agent_won = done && rew > 0;
        }
        win_code = ea->winner;
    }
    e->ep_len += 1;
    if (a->reward) a->reward[i] = (float)rew;
    if (a->done) a->done[i] = (uint8_t)(done != 0);
    if (a->winner) a->winner[i] = (int8_t)win_code;
    if (done) {
        finish_episode(v, e, agent_won);
        if (a->tobs) memcpy(a->tobs + (size_t)i * C, sim_of(v, e)->board, (size_t)C);
        if (!(v->flags & HEXO_F_NO_AUTO_RESET)) reset_one(v, e, NULL);
    }
    write_obs(v, e, a->obs ? a->obs + (size_t)i * C : NULL,
              a->legal ? a->legal + (size_t)i * C : NULL);
}

static void step_range(hexo_vec *v, int lo, int hi, void *p)
{
    step_args *a = (step_args *)p;
    for (int r = 0; r < a->repeat; ++r)
        for (int i = lo; i < hi; ++i) step_one(v, &v->envs[i], i, a);
}

void hexo_vec_step(hexo_vec *v, const int32_t *actions, const int32_t *opp_actions, int8_t *obs,
                   uint8_t *legal, float *reward, uint8_t *done, int8_t *winner,
                   int8_t *terminal_obs)
{
    step_args a = {actions, opp_actions, obs, legal, reward, done, winner, terminal_obs, 1};
    parallel_for(v, step_range, &a);
}

/* bench helper: K RNG-agent steps per env, outputs overwritten each step */
void hexo_vec_step_repeat(hexo_vec *v, int K, int8_t *obs, uint8_t *legal, float *reward,
                          uint8_t *done)
{
    step_args a = {NULL, NULL, obs, legal, reward, done, NULL, NULL, K};
    parallel_for(v, step_range, &a);
}

/* ---- literal plies ---- */
typedef struct {
    const int32_t *actions;
    int T;
    int8_t *ret;
    uint8_t *done;
    int8_t *rb;
    uint8_t *legal;
    int8_t *obs;
} am_args;

static void am_range(hexo_vec *v, int lo, int hi, void *p)
{
    am_args *a = (am_args *)p;
    const int C = v->n * v->n;
    const size_t E = (size_t)v->E;
    for (int t = 0; t < a->T; ++t)
        for (int i = lo; i < hi; ++i) {
            venv *e = &v->envs[i];
            size_t o = (size_t)t * E + (size_t)i;
            int act = a->actions[o];
            int ret, done, rb = 0;
            if (v->enc == HEXO_ENC_B) {
                done = hexo_envb_step(&e->u.sp.env, act, &rb);
                ret = e->u.sp.env.winner;
            } else {
                ret = hexo_make_move(&e->u.ea.sim, act);
                done = e->u.ea.sim.done;
            }
            if (a->ret) a->ret[o] = (int8_t)ret;
            if (a->done) a->done[o] = (uint8_t)(done != 0);
            if (a->rb) a->rb[o] = (int8_t)rb;
            write_obs(v, e, a->obs ? a->obs + o * C : NULL, a->legal ? a->legal + o * C : NULL);
        }
}

void hexo_vec_apply_moves(hexo_vec *v, const int32_t *actions, int T, int8_t *ret_out,
                          uint8_t *done_out, int8_t *r_black_out, uint8_t *legal_out,
                          int8_t *obs_out)
{
    am_args a = {actions, T, ret_out, done_out, r_black_out, legal_out, obs_out};
    parallel_for(v, am_range, &a);
}

/* ---- W2 rollout ---- */
typedef struct {
    int T;
    int8_t *obs;
    uint8_t *legal;
    int32_t *actions;
    float *reward;
    uint8_t *done;
} ro_args;

static void ro_range(hexo_vec *v, int lo, int hi, void *p)
{
    ro_args *a = (ro_args *)p;
    const int C = v->n * v->n;
    const size_t E = (size_t)v->E;
    for (int t = 0; t < a->T; ++t)
        for (int i = lo; i < hi; ++i) {
            venv *e = &v->envs[i];
            size_t o = (size_t)t * E + (size_t)i;
            write_obs(v, e, a->obs ? a->obs + o * C : NULL, a->legal ? a->legal + o * C : NULL);
            step_args s = {NULL, NULL, NULL, NULL, NULL, NULL, NULL, NULL, 1};
            float rew;
            uint8_t dn;
            s.reward = &rew - i; /* step_one indexes by i */
            s.done = &dn - i;
            /* draw the action here so it can be recorded */
            int act = rng_action_stored(v, e);
            int32_t act32 = act;
            s.actions = &act32 - i;
            step_one(v, e, i, &s);
            if (a->actions) a->actions[o] = act;
            if (a->reward) a->reward[o] = rew;
            if (a->done) a->done[o] = dn;
        }
}

void hexo_vec_rollout(hexo_vec *v, int T, int8_t *obs, uint8_t *legal, int32_t *actions,
                      float *reward, uint8_t *done)
{
    ro_args a = {T, obs, legal, actions, reward, done};
    parallel_for(v, ro_range, &a);
}

/* ---- W1 rollout: T whole env steps with the RNG agent, every step recorded (the agent's action in the
 * view it was shown; obs/mask AFTER the step, as hexo_vec_step returns them) ---- */
static void ro1_range(hexo_vec *v, int lo, int hi, void *p)
{
    ro_args *a = (ro_args *)p;
    const int C = v->n * v->n;
    const size_t E = (size_t)v->E;
    for (int t = 0; t < a->T; ++t)
        for (int i = lo; i < hi; ++i) {
            venv *e = &v->envs[i];
            size_t o = (size_t)t * E + (size_t)i;
            step_args s = {NULL, NULL, NULL, NULL, NULL, NULL, NULL, NULL, 1};
            float rew;
            uint8_t dn;
            int32_t act32 = rng_action_stored(v, e); /* the draw step_one would make itself */
            s.actions = &act32 - i;                  /* step_one indexes by i */
            s.reward = &rew - i;
            s.done = &dn - i;
            s.obs = a->obs ? a->obs + (o - (size_t)i) * C : NULL;
            s.legal = a->legal ? a->legal + (o - (size_t)i) * C : NULL;
            step_one(v, e, i, &s);
            if (a->actions) a->actions[o] = act32;
            if (a->reward) a->reward[o] = rew;
            if (a->done) a->done[o] = dn;
        }
}

void hexo_vec_rollout_selfplay(hexo_vec *v, int T, int8_t *obs, uint8_t *legal, int32_t *actions,
                               float *reward, uint8_t *done)
{
    ro_args a = {T, obs, legal, actions, reward, done};
    parallel_for(v, ro1_range, &a);
}

void hexo_vec_get_state(hexo_vec *v, int8_t *board, uint8_t *labels, uint8_t *cur_player)
{
    const int C = v->n * v->n, P = (v->n + 2) * (v->n + 2);
    for (int i = 0; i < v->E; ++i) {
        hexo_game *g = sim_of(v, &v->envs[i]);
        if (board) memcpy(board + (size_t)i * C, g->board, (size_t)C);
        if (labels)
            for (int p = 0; p < 2; ++p)
                for (int c = 0; c < P; ++c)
                    labels[((size_t)i * 2 + p) * P + c] = (uint8_t)g->regions[p][c];
        if (cur_player) cur_player[i] = (uint8_t)g->cur;
    }
}

void hexo_vec_episode_stats(hexo_vec *v, double *out4)
{
    out4[0] = out4[1] = out4[2] = out4[3] = 0;
    for (int i = 0; i < v->E; ++i) {
        out4[0] += v->envs[i].s_ret;
        out4[1] += v->envs[i].s_len;
        out4[2] += v->envs[i].s_eps;
        out4[3] += v->envs[i].s_wins;
    }
}

void hexo_vec_agent_colours(hexo_vec *v, uint8_t *out)
{
    for (int i = 0; i < v->E; ++i) out[i] = (uint8_t)v->envs[i].agent;
}
