/*
 * mse_oracle.c -- CPU ORACLE (test infrastructure only; see mse_oracle.h header).
 *
 * Build with:  gcc -O2 -ffp-contract=off -fno-fast-math -shared -fPIC   (oracle/Makefile)
 * -ffp-contract=off matters: numpy evaluates `low + range*u`, cumsum and the cdf
 * normalisation with separately rounded IEEE-754 double operations.
 *
 * Every function cites the reference lines it restates (paths relative to the reference
 * checkout).  "numpy:" citations name the numpy 2.2.6 routine whose published algorithm is
 * restated (numpy is a third-party dependency of the reference, not vendored in it).
 */
#include "mse_oracle.h"

#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

typedef unsigned __int128 u128;

/* ====================================================================================== *
 *  numpy.random restatement
 * ====================================================================================== */

/* numpy: _seed_seq.pyx SeedSequence (pool_size=4), constants from the same file */
#define SS_INIT_A 0x43b0d7e5u
#define SS_MULT_A 0x931e8875u
#define SS_INIT_B 0x8b51f9ddu
#define SS_MULT_B 0x58f38dedu
#define SS_MIX_L  0xca01f9ddu
#define SS_MIX_R  0x4973f715u
#define SS_XSHIFT 16

static uint32_t ss_hashmix(uint32_t value, uint32_t *hash_const)
{
    value ^= *hash_const;
    *hash_const *= SS_MULT_A;
    value *= *hash_const;
    value ^= value >> SS_XSHIFT;
    return value;
}

static uint32_t ss_mix(uint32_t x, uint32_t y)
{
    uint32_t r = SS_MIX_L * x - SS_MIX_R * y;
    r ^= r >> SS_XSHIFT;
    return r;
}

/* SeedSequence(entropy).generate_state(4, uint64) for a non-negative python int < 2**64.
 * numpy converts the int to little-endian uint32 words, dropping leading zero words
 * (0 -> one zero word). */
void orc_seed_sequence_u64x4(uint64_t entropy, uint64_t out[4])
{
    uint32_t words[2];
    int n_words = 1;
    words[0] = (uint32_t)entropy;
    words[1] = (uint32_t)(entropy >> 32);
    if (words[1] != 0) n_words = 2;

    uint32_t pool[4];
    uint32_t hc = SS_INIT_A;
    for (int i = 0; i < 4; ++i)
        pool[i] = ss_hashmix(i < n_words ? words[i] : 0u, &hc);
    for (int src = 0; src < 4; ++src)
        for (int dst = 0; dst < 4; ++dst)
            if (src != dst)
                pool[dst] = ss_mix(pool[dst], ss_hashmix(pool[src], &hc));
    /* n_words <= pool size, so there is no "remaining entropy" pass */

    uint32_t st[8];
    uint32_t hb = SS_INIT_B;
    for (int i = 0; i < 8; ++i) {
        uint32_t v = pool[i & 3];
        v ^= hb;
        hb *= SS_MULT_B;
        v *= hb;
        v ^= v >> SS_XSHIFT;
        st[i] = v;
    }
    for (int i = 0; i < 4; ++i)
        out[i] = (uint64_t)st[2 * i] | ((uint64_t)st[2 * i + 1] << 32);
}

/* numpy: pcg64.h  PCG_DEFAULT_MULTIPLIER_128 */
#define PCG_MULT ((((u128)0x2360ED051FC65DA4ull) << 64) | (u128)0x4385DF649FCCF645ull)

static inline u128 g_state(const orc_pcg64 *g) { return ((u128)g->state_hi << 64) | g->state_lo; }
static inline u128 g_inc(const orc_pcg64 *g) { return ((u128)g->inc_hi << 64) | g->inc_lo; }
static inline void g_set_state(orc_pcg64 *g, u128 s)
{
    g->state_hi = (uint64_t)(s >> 64);
    g->state_lo = (uint64_t)s;
}

/* numpy: pcg64.c pcg64_set_seed + pcg_setseq_128_srandom_r; PCG64.__init__ takes
 * generate_state(4, uint64): [0:2] -> initstate (hi, lo), [2:4] -> initseq (hi, lo). */
void orc_pcg64_seed(orc_pcg64 *g, uint64_t seed)
{
    uint64_t w[4];
    orc_seed_sequence_u64x4(seed, w);
    u128 initstate = ((u128)w[0] << 64) | w[1];
    u128 initseq = ((u128)w[2] << 64) | w[3];
    u128 inc = (initseq << 1) | 1u;
    u128 s = 0;
    s = s * PCG_MULT + inc;
    s += initstate;
    s = s * PCG_MULT + inc;
    g->inc_hi = (uint64_t)(inc >> 64);
    g->inc_lo = (uint64_t)inc;
    g_set_state(g, s);
    g->has_uint32 = 0;
    g->uinteger = 0;
}

/* numpy: pcg64.h pcg64_next64 = step, then XSL-RR of the NEW state */
uint64_t orc_pcg64_next64(orc_pcg64 *g)
{
    u128 s = g_state(g) * PCG_MULT + g_inc(g);
    g_set_state(g, s);
    uint64_t hi = (uint64_t)(s >> 64), lo = (uint64_t)s;
    uint64_t x = hi ^ lo;
    unsigned rot = (unsigned)(hi >> 58);
    return (x >> rot) | (x << ((-rot) & 63));
}

/* numpy: pcg64.h pcg64_next32 (lower half first, upper half buffered) */
uint32_t orc_pcg64_next32(orc_pcg64 *g)
{
    if (g->has_uint32) {
        g->has_uint32 = 0;
        return g->uinteger;
    }
    uint64_t next = orc_pcg64_next64(g);
    g->has_uint32 = 1;
    g->uinteger = (uint32_t)(next >> 32);
    return (uint32_t)next;
}

/* numpy: distributions.h next_double */
double orc_pcg64_random(orc_pcg64 *g)
{
    return (double)(orc_pcg64_next64(g) >> 11) * (1.0 / 9007199254740992.0);
}

/* numpy: distributions.c random_uniform(state, lower, range) with range = high - low */
double orc_pcg64_uniform(orc_pcg64 *g, double lo, double hi)
{
    double range = hi - lo;
    double scaled = range * orc_pcg64_random(g);
    return lo + scaled;
}

/* numpy: distributions.c buffered_bounded_lemire_uint32 via random_bounded_uint64_fill,
 * the path taken by Generator.integers(lo, hi) (int64) and Generator.choice(n) / choice(arr)
 * when hi-lo-1 < 2**32 - 1.  No draw at all when the range is a single value. */
int64_t orc_pcg64_integers(orc_pcg64 *g, int64_t lo, int64_t hi_excl)
{
    uint32_t rng = (uint32_t)(hi_excl - lo - 1);
    if (rng == 0) return lo;
    uint32_t rng_excl = rng + 1u;
    uint64_t m = (uint64_t)orc_pcg64_next32(g) * rng_excl;
    uint32_t leftover = (uint32_t)m;
    if (leftover < rng_excl) {
        uint32_t threshold = (0xFFFFFFFFu - rng) % rng_excl;
        while (leftover < threshold) {
            m = (uint64_t)orc_pcg64_next32(g) * rng_excl;
            leftover = (uint32_t)m;
        }
    }
    return lo + (int64_t)(m >> 32);
}

/* numpy: _generator.pyx Generator.choice(n, p=p), replace=True, size=None:
 *   cdf = p.cumsum(); cdf /= cdf[-1]; u = random(); idx = cdf.searchsorted(u, 'right') */
int orc_pcg64_choice_p(orc_pcg64 *g, const double *p, int n)
{
    double cdf[16];
    double acc = p[0];
    cdf[0] = acc;
    for (int k = 1; k < n; ++k) {
        acc = acc + p[k];
        cdf[k] = acc;
    }
    double last = cdf[n - 1];
    for (int k = 0; k < n; ++k) cdf[k] = cdf[k] / last;
    double u = orc_pcg64_random(g);
    int idx = 0; /* side='right': number of entries <= u */
    while (idx < n && cdf[idx] <= u) ++idx;
    return idx;
}

/* numpy: Generator.permutation([1,2]) -> shuffle of 2 items: one iteration i=1,
 * j = random_interval(1) which draws one buffered uint32 and masks it with 1; swap iff j==0.
 * utils/input_generator.py:28-30 */
int orc_permutation12_first(uint64_t seed)
{
    orc_pcg64 g;
    orc_pcg64_seed(&g, seed);
    uint32_t j = orc_pcg64_next32(&g) & 1u;
    return j == 0 ? 2 : 1;
}

/* round(np.float64, 2) == rint(x*100)/100 (numpy multiplies, rints, divides) */
double orc_round2(double x)
{
    double scaled = x * 100.0;
    double r = nearbyint(scaled); /* default rounding mode: half to even */
    return r / 100.0;
}

/* Python's round(float, 2): correctly rounded on the exact binary value, exact ties to even - what the reference gets
 * where the operand is a plain Python float (a threshold read from config.yml), unlike round(np.float64, 2) above.
 * glibc's printf rounds the exact value the same way. */
double orc_round2_py(double x)
{
    char buf[64];
    snprintf(buf, sizeof buf, "%.2f", x);
    return strtod(buf, NULL);
}

/* int(round(np.float64)) : half to even */
int64_t orc_rint_i64(double x) { return (int64_t)nearbyint(x); }

/* ====================================================================================== *
 *  configuration (config.yml) and construction
 * ====================================================================================== */

void orc_config_default(orc_config *c)
{
    memset(c, 0, sizeof(*c));
    c->input_occupancy_min = 60;  /* config.yml:5 */
    c->input_occupancy_max = 80;  /* config.yml:6 */
    c->input_batch_size = 100;    /* config.yml:7 */
    c->steps_per_pattern = 20;    /* config.yml:8 (reset() uses the generator's default 20, env_super.py:375) */
    for (int i = 0; i < 4; ++i) c->baseline_accuracy[i] = 0.75; /* config.yml:13 */
    c->boost = 0.5;               /* config.yml:14 */
    c->noise = 0.05;              /* config.yml:17; ctor arg noise_sorting overrides (env_super.py:71) */
    c->stage_capacity = 100;      /* config.yml:18 */
    c->press_time[0] = 12;        /* config.yml:23 */
    c->press_time[1] = 15;        /* config.yml:24 */
    c->container_capacity = 700;  /* config.yml:25 */
    c->bale_standard_size = 200;  /* config.yml:26; ctor arg balesize overrides (env_super.py:87) */
    c->bale_remainder_threshold = 0.5; /* config.yml:27 */
    for (int i = 0; i < 4; ++i) c->quality_threshold[i] = 0.9; /* config.yml:29-32 */
    c->purity_threshold_theta = 0.80;  /* config.yml:39 */
    c->tanh_temperature = 0.5;         /* config.yml:45 */
    c->overflow_penalty_catastrophic = -1.0; /* config.yml:50 */
    c->overflow_penalty_severe = -0.5;       /* config.yml:51 */
    c->overflow_penalty_mild = -0.2;         /* config.yml:52 */
    c->bale_efficiency_factor = 1.0;         /* config.yml:54 */
    c->max_state_reward = 0.5;               /* config.yml:56 */
    c->overflow_termination_penalty = -10.0; /* config.yml:59 */
    /* utils/input_generator.py:17-20, order A,B,C,D */
    const double p1[4] = {0.40, 0.15, 0.35, 0.10};
    const double p2[4] = {0.15, 0.40, 0.10, 0.35};
    memcpy(c->pattern_ratio[0], p1, sizeof(p1));
    memcpy(c->pattern_ratio[1], p2, sizeof(p2));
    c->env_kind = ORC_ENV_MONO;
    c->max_steps = 50;            /* env_monolith.py:22 */
}

/* The build's own rule for reset(seed=None): the reference seeds a fresh generator from OS
 * entropy there (env_super.py:375), which is not reproducible and therefore excluded from
 * parity.  Rule (shared with the HIP path): first pattern = 1 + (splitmix64_finalizer(
 * inc_lo(rng) ^ episode * 0x9E3779B97F4A7C15) & 1), episode = number of resets since (and
 * including) the last seeded one. */
static uint64_t mix64(uint64_t z)
{
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

static int unseeded_first_pattern(const orc_env *e)
{
    uint64_t h = mix64(e->rng.inc_lo ^ ((uint64_t)e->episode * 0x9E3779B97F4A7C15ull));
    return 1 + (int)(h & 1u);
}

/* env_super.py:165-184 set_seed: five streams at seed+1/+2/+3/+4/+99 */
static void env_set_seed(orc_env *e, uint64_t seed)
{
    orc_pcg64_seed(&e->rng_input, seed + 1);
    orc_pcg64_seed(&e->rng_sorting, seed + 2);
    orc_pcg64_seed(&e->rng_pressing, seed + 3);
    orc_pcg64_seed(&e->rng_noise, seed + 4);
    orc_pcg64_seed(&e->rng, seed + 99);
}

static void bales_clear(orc_env *e)
{
    for (int m = 0; m < 5; ++m) e->n_bales[m] = 0;
}

static void bales_push(orc_env *e, int m, int64_t size, int32_t q)
{
    if (e->n_bales[m] == e->cap_bales[m]) {
        int32_t cap = e->cap_bales[m] ? 2 * e->cap_bales[m] : 16;
        e->bales[m] = (orc_bale *)realloc(e->bales[m], (size_t)cap * sizeof(orc_bale));
        e->cap_bales[m] = cap;
    }
    e->bales[m][e->n_bales[m]].size = size;
    e->bales[m][e->n_bales[m]].q = q;
    e->n_bales[m]++;
}

/* env_super.py:25-137 __init__ : set_seed(seed) with `seed or 0`, generator seeded with
 * `seed` (None -> unseeded rule). */
orc_env *orc_env_create(const orc_config *cfg, int has_seed, uint64_t seed)
{
    orc_env *e = (orc_env *)calloc(1, sizeof(orc_env));
    if (!e) return NULL;
    e->cfg = *cfg;
    env_set_seed(e, has_seed ? seed : 0);
    e->episode = 0;
    float obs[32];
    /* __init__ leaves the same state a reset would, without re-seeding the streams */
    orc_env_reset(e, has_seed, seed, obs);
    /* __init__'s set_seed already happened above; reset(seed) re-seeds identically */
    return e;
}

void orc_env_destroy(orc_env *e)
{
    if (!e) return;
    for (int m = 0; m < 5; ++m) free(e->bales[m]);
    free(e);
}

int orc_env_obs_dim(const orc_env *e)
{
    switch (e->cfg.env_kind) {
    case ORC_ENV_SORT: return 13;  /* env_1_sort.py:71 */
    case ORC_ENV_PRESS: return 16; /* env_2_press.py:62 */
    default: return 29;            /* env_monolith.py:72-76 */
    }
}

int orc_env_num_actions(const orc_env *e)
{
    switch (e->cfg.env_kind) {
    case ORC_ENV_SORT: return 2;   /* env_1_sort.py:72 */
    case ORC_ENV_PRESS: return 11; /* env_2_press.py:64 */
    default: return 22;            /* env_monolith.py:79 */
    }
}

/* ====================================================================================== *
 *  observation, purity, masks
 * ====================================================================================== */

static int64_t level_of(const orc_env *e, int m)
{
    return m < 4 ? e->cont_true[m] + e->cont_false[m] : e->cont_e;
}

/* env_super.py:771-791 get_container_purity (per-material part) */
static void container_purity(const orc_env *e, double purity[4])
{
    for (int m = 0; m < 4; ++m) {
        int64_t total = e->cont_true[m] + e->cont_false[m];
        if (total > 0) /* an np.float64 quotient: numpy's round */
            purity[m] = orc_round2((double)e->cont_true[m] / (double)total);
        else           /* :786-789 the threshold itself, a Python float from config.yml: Python's round */
            purity[m] = orc_round2_py(e->cfg.quality_threshold[m]);
    }
}

static float clipf(float v, float lo, float hi) { return v < lo ? lo : (v > hi ? hi : v); }

/* env_super.py:306-325 get_sort_obs (+ :199-210 belt proportions, :212-227 purity diffs) */
static void sort_obs(const orc_env *e, float *o)
{
    int total = e->belt[0] + e->belt[1] + e->belt[2] + e->belt[3];
    o[0] = (float)e->belt_occupancy;
    for (int m = 0; m < 4; ++m)
        o[1 + m] = total > 0 ? (float)((double)e->belt[m] / (double)total) : 0.0f;
    for (int m = 0; m < 4; ++m) o[5 + m] = (float)e->acc_belt[m];
    double purity[4];
    container_purity(e, purity);
    for (int m = 0; m < 4; ++m) { /* :212-227: round(diff, 2) rounds as its operand's type does (see container_purity) */
        double diff = purity[m] - e->cfg.quality_threshold[m];
        int empty = e->cont_true[m] + e->cont_false[m] == 0;
        o[9 + m] = (float)(empty ? orc_round2_py(diff) : orc_round2(diff));
    }
    for (int i = 0; i < 13; ++i) o[i] = clipf(o[i], -1.0f, 1.0f);
}

/* env_super.py:327-359 get_press_obs */
static void press_obs(const orc_env *e, float *o)
{
    for (int m = 0; m < 5; ++m) {
        float v = (float)((double)level_of(e, m) / (double)e->cfg.container_capacity);
        o[m] = v;
        o[5 + m] = v;
    }
    for (int m = 0; m < 4; ++m)
        o[10 + m] = (float)((double)e->sorting[m] / (double)e->cfg.stage_capacity);
    for (int p = 0; p < 2; ++p)
        o[14 + p] = (float)((double)e->press_timer[p] / (double)e->cfg.press_time[p]);
    for (int i = 0; i < 16; ++i) o[i] = clipf(o[i], 0.0f, 1.0f);
}

/* env_1_sort.py:90-91, env_2_press.py:80-81, env_monolith.py:98-104 */
void orc_env_obs(const orc_env *e, float *o)
{
    switch (e->cfg.env_kind) {
    case ORC_ENV_SORT: sort_obs(e, o); break;
    case ORC_ENV_PRESS: press_obs(e, o); break;
    default:
        sort_obs(e, o);
        press_obs(e, o + 13);
        break;
    }
}

/* env_super.py:869-885 press_action_masks */
static void press_mask(const orc_env *e, uint8_t m11[11])
{
    memset(m11, 0, 11);
    m11[0] = 1;
    int p1_ready = e->press_timer[0] == 0, p2_ready = e->press_timer[1] == 0;
    for (int m = 0; m < 5; ++m) {
        if (level_of(e, m) >= e->cfg.bale_standard_size) {
            if (p1_ready) m11[1 + m] = 1;
            if (p2_ready) m11[6 + m] = 1;
        }
    }
}

/* env_1_sort.py:74-76, env_2_press.py:66-67, env_super.py:887-898 */
void orc_env_action_mask(const orc_env *e, uint8_t *out)
{
    uint8_t m11[11];
    switch (e->cfg.env_kind) {
    case ORC_ENV_SORT:
        out[0] = 1;
        out[1] = 1;
        break;
    case ORC_ENV_PRESS:
        press_mask(e, out);
        break;
    default:
        press_mask(e, m11);
        memcpy(out, m11, 11);
        memcpy(out + 11, m11, 11);
        break;
    }
}

/* ====================================================================================== *
 *  reset
 * ====================================================================================== */

/* env_super.py:365-420 reset + the variants' overrides (env_1_sort.py:81-85,
 * env_2_press.py:73-75, env_monolith.py:91-93) */
void orc_env_reset(orc_env *e, int has_seed, uint64_t seed, float *obs_out)
{
    for (int m = 0; m < 4; ++m) {
        e->cont_true[m] = 0;
        e->cont_false[m] = 0;
        e->input[m] = e->belt[m] = e->sorting[m] = 0;
        e->acc_belt[m] = e->cfg.baseline_accuracy[m];
        e->acc_sorter[m] = e->cfg.baseline_accuracy[m];
    }
    e->cont_e = 0;
    e->current_step = 0;

    /* new SeasonalInputGenerator(seed=seed) with the default steps_per_pattern (env_super.py:375) */
    if (has_seed) {
        /* utils/input_generator.py:26-30: default_rng(seed), then permutation([1, 2]) = one next_uint32 */
        orc_pcg64_seed(&e->rng_gen, seed);
        e->gen_first = (orc_pcg64_next32(&e->rng_gen) & 1u) ? 1 : 2;
        env_set_seed(e, seed); /* env_super.py:377-378 */
        e->episode = 1;        /* episodes are counted from the last seeded reset */
    } else {
        /* the build's rule for the OS-entropy generator of reset(seed=None): pattern order and stream from one hash */
        e->gen_first = unseeded_first_pattern(e);
        orc_pcg64_seed(&e->rng_gen, mix64(e->rng.inc_lo ^ ((uint64_t)e->episode * 0x9E3779B97F4A7C15ull)));
        e->episode++;
    }
    e->gen_idx = 0;
    e->gen_counter = 0;

    for (int p = 0; p < 2; ++p) {
        e->press_timer[p] = 0;
        e->press_mat[p] = -1;
        e->press_n[p] = 0;
        e->press_q[p] = 0.0;
    }
    e->last_press_started = 0;
    e->last_press_amount = 0;
    e->sensor_mode = 0;
    e->input_occupancy = 0.0;
    e->belt_occupancy = 0.0;
    bales_clear(e);
    e->last_log_code = -1;
    e->last_log_mat = -1;
    e->last_internal_press_action = 0;
    e->draws_this_step = 0;
    if (obs_out) orc_env_obs(e, obs_out);
}

/* ====================================================================================== *
 *  step pieces
 * ====================================================================================== */

/* utils/input_generator.py:37-64 generate_input, reduced to the material counts that update_environment keeps
 * (env_super.py:448-453).  When floor(ratio * batch) leaves units over, each goes to rng.choice(material_names)
 * (:49-55) and the batch list is shuffled (:58-61): Generator.shuffle of a Python list draws random_interval(i) for
 * i = n-1 .. 1 (masked rejection on next_uint32) - its result is never read, but the draws move the generator's private
 * stream.  With a remainder-free batch (config.yml's 100) nothing of that stream is ever observed and it is left alone. */
static int gen_has_remainder(const orc_config *c)
{
    for (int key = 0; key < 2; ++key) {
        int sum = 0;
        for (int m = 0; m < 4; ++m) sum += (int)floor(c->pattern_ratio[key][m] * (double)c->input_batch_size);
        if (sum != c->input_batch_size) return 1;
    }
    return 0;
}

static int generate_counts(orc_env *e, int32_t counts[4])
{
    if (e->gen_counter >= 20) { /* reset() builds the generator with the default 20 */
        e->gen_idx = (e->gen_idx + 1) % 2;
        e->gen_counter = 0;
    }
    int key = e->gen_idx == 0 ? e->gen_first : 3 - e->gen_first;
    int sum = 0;
    for (int m = 0; m < 4; ++m) {
        counts[m] = (int32_t)floor(e->cfg.pattern_ratio[key - 1][m] * (double)e->cfg.input_batch_size);
        sum += counts[m];
    }
    if (gen_has_remainder(&e->cfg)) {
        for (int r = sum; r < e->cfg.input_batch_size; ++r) /* :53-55 */
            counts[orc_pcg64_integers(&e->rng_gen, 0, 4)] += 1;
        for (uint32_t i = (uint32_t)e->cfg.input_batch_size - 1u; i > 0u; --i) { /* :61 shuffle -> random_interval(i) */
            uint32_t mask = i;
            mask |= mask >> 1;
            mask |= mask >> 2;
            mask |= mask >> 4;
            mask |= mask >> 8;
            mask |= mask >> 16;
            while ((orc_pcg64_next32(&e->rng_gen) & mask) > i) {
            }
        }
    }
    e->gen_counter++;
    return 0;
}

/* env_super.py:433-461 update_environment */
static int update_environment(orc_env *e)
{
    for (int m = 0; m < 4; ++m) {
        e->sorting[m] = e->belt[m];
        e->belt[m] = e->input[m];
    }
    e->belt_occupancy = e->input_occupancy;
    int32_t counts[4];
    if (generate_counts(e, counts) != 0) return -1;
    int sum = 0;
    for (int m = 0; m < 4; ++m) {
        e->input[m] = counts[m];
        sum += counts[m];
    }
    /* python round(int/100, 2) of a two-decimal quotient is the quotient itself */
    e->input_occupancy = (double)sum / 100.0;
    for (int m = 0; m < 4; ++m) e->acc_sorter[m] = e->acc_belt[m];
    return 0;
}

/* What Env_2_Pressing.step hands its sorting agent (env_2_press.py:95-104): get_sort_obs() taken AFTER this
 * step's flow update and before the sensor setting.  A preview: the env is not changed (the flow update is a
 * function of the state; the occupancy draw it discards comes from an unobserved stream). */
void orc_env_sort_agent_obs(const orc_env *e, float *o)
{
    orc_env tmp = *e; /* shallow copy: sort_obs reads no bale list */
    update_environment(&tmp);
    sort_obs(&tmp, o);
}

/* What Env_3_Monolith.step hands a press_agent in mode='model' (env_monolith.py:198-210) and, concatenated behind
 * the sorting view, a stored mono_agent (env_monolith.py:113-114,144-146 -> :98-104): get_press_obs() after this
 * step's flow update.  The same preview as above: the env is not changed. */
void orc_env_press_agent_obs(const orc_env *e, float *o)
{
    orc_env tmp = *e; /* shallow copy: press_obs reads no bale list */
    update_environment(&tmp);
    press_obs(&tmp, o);
}

/* env_super.py:484-509 set_multisensor_mode + update_accuracy */
static void update_accuracy(orc_env *e, int mode)
{
    e->sensor_mode = mode;
    double acc[4];
    for (int m = 0; m < 4; ++m) acc[m] = e->cfg.baseline_accuracy[m];
    if (mode == 0) {
        acc[0] += e->cfg.boost;
        acc[2] += e->cfg.boost;
    } else if (mode == 1) {
        acc[1] += e->cfg.boost;
        acc[3] += e->cfg.boost;
    }
    double n = e->cfg.noise;
    for (int m = 0; m < 4; ++m) {
        double v = acc[m] + orc_pcg64_uniform(&e->rng_noise, -n, n);
        e->acc_belt[m] = v < 0.0 ? 0.0 : (v > 1.0 ? 1.0 : v);
    }
}

/* env_super.py:511-609 sort_material */
static void sort_material(orc_env *e)
{
    int64_t leftover[4], true_arr[4], false_arr[4];
    for (int m = 0; m < 4; ++m) leftover[m] = e->sorting[m];
    e->draws_this_step = 0;
    for (int i = 0; i < 4; ++i) {
        int64_t target = leftover[i];
        int64_t t = orc_rint_i64((double)target * e->acc_sorter[i]); /* :539 */
        int64_t f = target - t;
        true_arr[i] = t;
        false_arr[i] = f;
        leftover[i] = f; /* :546 */
        for (int64_t d = 0; d < f; ++d) { /* :553-571 */
            int64_t total = leftover[0] + leftover[1] + leftover[2] + leftover[3];
            if (total == 0) break;
            double p[4];
            for (int k = 0; k < 4; ++k) p[k] = (double)leftover[k] / (double)total;
            int sel = orc_pcg64_choice_p(&e->rng, p, 4);
            e->draws_this_step++;
            leftover[sel] -= 1;
        }
    }
    e->cont_e += leftover[0] + leftover[1] + leftover[2] + leftover[3]; /* :579,597 */
    for (int m = 0; m < 4; ++m) { /* :600-602 */
        e->cont_true[m] += true_arr[m];
        e->cont_false[m] += false_arr[m];
    }
}

static void log_entry(orc_env *e, int code, int mat)
{
    e->last_log_code = code;
    e->last_log_mat = mat;
    if (e->step_n_log < 2) {
        e->step_log_code[e->step_n_log] = code;
        e->step_log_mat[e->step_n_log] = mat;
    }
    e->step_n_log++;
}

/* env_super.py:661-687 press_bale */
static void press_bale(orc_env *e, int mat, int64_t n, double q)
{
    int32_t qi = (int32_t)(q * 100.0); /* int(q*100): truncation */
    if (e->step_n_bale < 2) {
        e->step_bale_mat[e->step_n_bale] = mat;
        e->step_bale_n[e->step_n_bale] = n;
        e->step_bale_q[e->step_n_bale] = qi;
    }
    e->step_n_bale++;
    int64_t S = e->cfg.bale_standard_size;
    int64_t full = n / S, rem = n % S;
    for (int64_t k = 0; k < full; ++k) bales_push(e, mat, S, qi);
    if (rem > 0) {
        if ((double)rem > (double)S * e->cfg.bale_remainder_threshold) {
            bales_push(e, mat, rem, qi);
        } else if (e->n_bales[mat] > 0) {
            e->bales[mat][e->n_bales[mat] - 1].size += rem;
        } else {
            bales_push(e, mat, rem, qi);
        }
    }
}

/* env_super.py:642-659 check_press_status */
static void check_press_status(orc_env *e)
{
    for (int p = 0; p < 2; ++p) {
        if (e->press_timer[p] > 0) {
            e->press_timer[p] -= 1;
            if (e->press_timer[p] == 0) {
                press_bale(e, e->press_mat[p], e->press_n[p], e->press_q[p]);
                e->press_mat[p] = -1;
                e->press_n[p] = 0;
                e->press_q[p] = 0.0;
            }
        }
    }
}

/* env_super.py:722-769 use_press; press in {1,2}, mat in 0..4 */
static void use_press(orc_env *e, int press, int mat)
{
    int p = press - 1;
    if (e->press_timer[p] > 0) { /* :725-733 busy */
        log_entry(e, press == 1 ? 111 : 222, mat);
        return;
    }
    log_entry(e, press, mat); /* :736 */
    int64_t total = level_of(e, mat);
    e->last_press_started = 1;
    e->last_press_amount = total;
    double quality = 0.0;
    if (mat < 4) {
        if (total > 0) quality = orc_round2((double)e->cont_true[mat] / (double)total);
        e->cont_true[mat] = 0;
        e->cont_false[mat] = 0;
    } else {
        e->cont_e = 0;
    }
    e->press_timer[p] = e->cfg.press_time[p];
    e->press_mat[p] = mat;
    e->press_n[p] = total;
    e->press_q[p] = quality;
}

/* env_super.py:626-640 press_action_rules; press_action 0..10 (0 = (None,None)/(0,None)) */
static void press_action_rules(orc_env *e, int press_action)
{
    check_press_status(e);
    if (press_action == 0) {
        log_entry(e, 0, -1);
        return;
    }
    /* env_super.py:804-809 press_discrete_to_action */
    int press = press_action <= 5 ? 1 : 2;
    int mat = (press_action - 1) % 5;
    use_press(e, press, mat);
}

/* env_super.py:811-836 validate_press_action */
static int validate_press_action(const orc_env *e, int press_action)
{
    if (press_action == 0) return 1;
    int press = press_action <= 5 ? 1 : 2;
    int mat = (press_action - 1) % 5;
    if (e->press_timer[press - 1] > 0) return 0;
    if (level_of(e, mat) < e->cfg.bale_standard_size) return 0;
    return 1;
}

/* env_super.py:469-482 sorting_rules */
static int sorting_rules(const orc_env *e)
{
    int total = e->belt[0] + e->belt[1] + e->belt[2] + e->belt[3];
    double pr[4];
    for (int m = 0; m < 4; ++m) pr[m] = total > 0 ? (double)e->belt[m] / (double)total : 0.0;
    return (pr[0] + pr[2] > pr[1] + pr[3]) ? 0 : 1;
}

/* env_super.py:291-300 sample_masked_press_action */
static int sample_masked_press_action(orc_env *e)
{
    uint8_t m11[11];
    int valid[11], n = 0;
    press_mask(e, m11);
    for (int a = 0; a < 11; ++a)
        if (m11[a]) valid[n++] = a;
    int idx = (int)orc_pcg64_integers(&e->rng_pressing, 0, n);
    return valid[idx];
}

/* env_super.py:963-1003 calculate_sorting_reward */
static double sorting_reward(const orc_env *e)
{
    double purity[4];
    container_purity(e, purity);
    double total = 0.0;
    for (int m = 0; m < 4; ++m) total += purity[m] - e->cfg.purity_threshold_theta;
    double state_based = (total / 4.0) * 2.0; /* purity_scaling_factor hard-coded :971 */
    return tanh(state_based / e->cfg.tanh_temperature);
}

/* env_super.py:1006-1080 calculate_press_reward */
static double press_reward(orc_env *e)
{
    double max_penalty = 0.0;
    double cap = (double)e->cfg.container_capacity;
    for (int m = 0; m < 5; ++m) {
        double fill = (double)level_of(e, m) / cap;
        if (fill > 1.0)
            return e->cfg.overflow_penalty_catastrophic;
        else if (fill > 0.95)
            max_penalty = fmin(max_penalty, e->cfg.overflow_penalty_severe);
        else if (fill > 0.90)
            max_penalty = fmin(max_penalty, e->cfg.overflow_penalty_mild);
    }
    if (max_penalty < 0.0) return max_penalty;

    int64_t total_level = 0, total_cap = 0;
    for (int m = 0; m < 5; ++m) {
        total_level += level_of(e, m);
        total_cap += e->cfg.container_capacity;
    }
    double state_reward = 0.0;
    if (total_cap > 0) state_reward = ((double)total_level / (double)total_cap) * e->cfg.max_state_reward;

    double action_reward = 0.0;
    if (e->last_press_started) {
        int64_t S = e->cfg.bale_standard_size;
        int64_t amount = e->last_press_amount;
        int64_t num_bales = amount / S, rem = amount % S;
        int64_t dist = rem < S - rem ? rem : S - rem;
        double bef = e->cfg.bale_efficiency_factor;
        double eff = (1.0 - 4.0 * ((double)dist / (double)S)) * bef;
        const double peaks[4] = {0.0, 1.0 / 3.0, 2.0 / 3.0, 1.0};
        int bi = num_bales < 3 ? (int)num_bales : 3;
        double bonus = peaks[bi] - bef;
        action_reward = eff + bonus;
        e->last_press_started = 0;
        e->last_press_amount = 0;
    }
    double r = state_reward + action_reward;
    return r < -1.0 ? -1.0 : (r > 1.0 ? 1.0 : r);
}

/* env_super.py:900-905 detect_overflow */
static int detect_overflow(const orc_env *e)
{
    for (int m = 0; m < 5; ++m)
        if (level_of(e, m) > e->cfg.container_capacity) return m + 1; /* (True, material m) */
    return 0;
}

/* ====================================================================================== *
 *  step (env_1_sort.py:97-154, env_2_press.py:88-165, env_monolith.py:109-284)
 * ====================================================================================== */

int orc_env_step(orc_env *e, int32_t action, int32_t sort_mode_in, uint32_t flags, float *obs_out,
                 double *reward_out, int32_t *terminated_out)
{
    const int kind = e->cfg.env_kind;
    const int unmasked = (flags & ORC_STEP_UNMASKED) != 0;
    const int late = (flags & ORC_STEP_SANITIZE_LATE) != 0; /* Env_3 mode='random' without masking */
    if (action < 0 || action >= orc_env_num_actions(e)) return -2;

    e->last_log_code = -1;
    e->last_log_mat = -1;
    e->last_internal_press_action = 0;
    e->step_n_log = e->step_n_bale = 0;
    for (int q = 0; q < 2; ++q) {
        e->step_log_code[q] = e->step_log_mat[q] = e->step_bale_mat[q] = e->step_bale_q[q] = -1;
        e->step_bale_n[q] = -1;
    }
    e->step_action = action;

    /* input_action_rules: draws rng_input.integers(60, 81); the value is discarded downstream
     * (env_super.py:911-922, :433,445) */
    (void)orc_pcg64_integers(&e->rng_input, e->cfg.input_occupancy_min, e->cfg.input_occupancy_max + 1);
    if (update_environment(e) != 0) return -3;

    int sort_mode, press_action = 0, run_press_rules = 1;
    if (kind == ORC_ENV_SORT) {
        sort_mode = action; /* env_1_sort.py:116 */
    } else if (kind == ORC_ENV_PRESS) {
        sort_mode = sort_mode_in >= 0 ? sort_mode_in : sorting_rules(e); /* env_2_press.py:106-112 */
        press_action = action; /* sanitised AFTER sort_material, see below */
    } else {
        sort_mode = action / 11; /* env_monolith.py:127-129 */
        press_action = action % 11;
        if (unmasked && !late && !validate_press_action(e, press_action)) { /* env_monolith.py:132-138 */
            log_entry(e, press_action <= 5 ? 111 : 222, (press_action - 1) % 5);
            run_press_rules = 0; /* press_action_tuple=None: no tick this step (:237-243) */
        }
    }

    update_accuracy(e, sort_mode);
    sort_material(e);

    if (kind == ORC_ENV_SORT) {
        press_action = sample_masked_press_action(e); /* env_1_sort.py:125-126 */
        e->last_internal_press_action = press_action;
    } else if (kind == ORC_ENV_PRESS && unmasked && !validate_press_action(e, press_action)) {
        /* env_2_press.py:125-131: Env_2 validates against the POST-sort levels (Env_3 validates
         * at decode time, before the sort: env_monolith.py:132) and still ticks the timers:
         * press_action_rules((None,None)) :138 */
        log_entry(e, press_action <= 5 ? 111 : 222, (press_action - 1) % 5);
        press_action = 0;
    }
    else if (kind == ORC_ENV_MONO && unmasked && late && !validate_press_action(e, press_action)) {
        /* env_monolith.py:245-253: mode='random' without masking sanitises in the "apply" section, after
         * sort_material; an invalid action is logged and press_action_rules is not called (no tick) */
        log_entry(e, press_action <= 5 ? 111 : 222, (press_action - 1) % 5);
        run_press_rules = 0;
    }
    /* Env_2 unmasked-invalid: the ledger gets the invalid entry first and the (0,None)
     * no-op entry after it, so the last entry reads as a no-op there. */
    if (run_press_rules) press_action_rules(e, press_action);

    e->step_overflow = (flags & ORC_STEP_CHECK_OVERFLOW) ? detect_overflow(e) : 0;
    if (e->step_overflow) {
        /* env_monolith.py:265-272 and the variants' equivalents */
        e->current_step += 1;
        if (obs_out) orc_env_obs(e, obs_out);
        *reward_out = e->cfg.overflow_termination_penalty;
        *terminated_out = 1;
        /* _log_step_data: env_monolith.py:271 (reward/2 each), env_1_sort.py:141 and env_2_press.py:152 (0, reward) */
        e->step_reward = *reward_out;
        e->step_r_sort = kind == ORC_ENV_MONO ? *reward_out / 2 : 0.0;
        e->step_r_press = kind == ORC_ENV_MONO ? *reward_out / 2 : *reward_out;
        e->step_done = 1;
        return 0;
    }

    double reward;
    e->step_r_sort = e->step_r_press = 0.0; /* env_1_sort.py:151, env_2_press.py:162, env_monolith.py:282 */
    if (kind == ORC_ENV_SORT)
        reward = e->step_r_sort = sorting_reward(e);
    else if (kind == ORC_ENV_PRESS)
        reward = e->step_r_press = press_reward(e);
    else {
        double rs = sorting_reward(e);
        double rp = press_reward(e);
        reward = rs + rp; /* env_monolith.py:274-276 */
        e->step_r_sort = rs;
        e->step_r_press = rp;
    }
    if (obs_out) orc_env_obs(e, obs_out);
    e->current_step += 1;
    *terminated_out = e->current_step >= e->cfg.max_steps;
    *reward_out = reward;
    e->step_reward = reward;
    e->step_done = *terminated_out;
    return 0;
}

/* ====================================================================================== *
 *  snapshot
 * ====================================================================================== */

static void pack_rng(const orc_pcg64 *g, uint64_t *w)
{
    w[0] = g->state_hi;
    w[1] = g->state_lo;
    w[2] = g->inc_hi;
    w[3] = g->inc_lo;
    w[4] = (uint64_t)g->has_uint32;
    w[5] = g->uinteger;
}

void orc_env_snapshot(const orc_env *e, int64_t *I, double *D, uint64_t *R)
{
    memset(I, 0, ORC_SNAP_INTS * sizeof(int64_t));
    for (int m = 0; m < 4; ++m) {
        I[0 + m] = e->input[m];
        I[4 + m] = e->belt[m];
        I[8 + m] = e->sorting[m];
        I[12 + m] = e->cont_true[m];
        I[16 + m] = e->cont_false[m];
        D[m] = e->acc_belt[m];
        D[4 + m] = e->acc_sorter[m];
    }
    I[20] = e->cont_e;
    for (int p = 0; p < 2; ++p) {
        I[21 + p] = e->press_timer[p];
        I[23 + p] = e->press_mat[p];
        I[25 + p] = e->press_n[p];
        I[27 + p] = (int64_t)nearbyint(e->press_q[p] * 100.0);
    }
    I[29] = e->sensor_mode;
    I[30] = e->last_press_started;
    I[31] = e->last_press_amount;
    I[32] = e->current_step;
    I[33] = e->gen_first;
    I[34] = e->gen_idx;
    I[35] = e->gen_counter;
    for (int m = 0; m < 5; ++m) {
        int64_t sum = 0;
        for (int k = 0; k < e->n_bales[m]; ++k) sum += e->bales[m][k].size;
        I[36 + m] = e->n_bales[m];
        I[41 + m] = sum;
        I[46 + m] = e->n_bales[m] ? e->bales[m][e->n_bales[m] - 1].size : 0;
        I[51 + m] = e->n_bales[m] ? e->bales[m][e->n_bales[m] - 1].q : 0;
    }
    I[56] = e->last_log_code;
    I[57] = e->last_log_mat;
    I[58] = e->last_internal_press_action;
    I[59] = e->draws_this_step;
    I[60] = e->episode;
    pack_rng(&e->rng, R);
    pack_rng(&e->rng_noise, R + 6);
    pack_rng(&e->rng_pressing, R + 12);
    pack_rng(&e->rng_sorting, R + 18);
    pack_rng(&e->rng_gen, R + 24);
}

/* ====================================================================================== *
 *  mode='model' without agents, trace record, bale lists
 * ====================================================================================== */

/* env_monolith.py:186-221: each part of the action is the env's own draw unless an agent decides it (draw_* = 0:
 * no draw from that stream, the part comes back as 0) */
int32_t orc_env_model_action(orc_env *e, int use_action_masking, int draw_sort, int draw_press)
{
    /* :195 rng_sorting.choice([0, 1]): an index draw over two entries */
    int sort_mode = draw_sort ? (int)orc_pcg64_integers(&e->rng_sorting, 0, 2) : 0;
    int press_action;
    if (!draw_press) {
        press_action = 0;
    } else if (use_action_masking) { /* :213-217 rng_pressing.choice(flatnonzero(press_action_masks())) */
        uint8_t m11[11];
        int valid[11], n = 0;
        press_mask(e, m11);
        for (int a = 0; a < 11; ++a)
            if (m11[a]) valid[n++] = a;
        press_action = n > 0 ? valid[(int)orc_pcg64_integers(&e->rng_pressing, 0, n)] : 0;
    } else { /* :219 rng_pressing.choice(11) */
        press_action = (int)orc_pcg64_integers(&e->rng_pressing, 0, 11);
    }
    return sort_mode * 11 + press_action; /* :221 */
}

/* the same with sort_agent = press_agent = None */
int32_t orc_env_model_fallback_action(orc_env *e, int use_action_masking)
{
    return orc_env_model_action(e, use_action_masking, 1, 1);
}

void orc_env_trace_record(const orc_env *e, double *t)
{
    for (int c = 0; c < 40; ++c) t[c] = 0.0;
    t[0] = e->step_action;
    t[1] = e->step_r_sort;
    t[2] = e->step_r_press;
    t[3] = e->sensor_mode;
    for (int m = 0; m < 4; ++m) {
        t[4 + m] = e->belt[m];
        t[8 + m] = (double)e->cont_true[m];
        t[12 + m] = (double)e->cont_false[m];
    }
    t[16] = (double)e->cont_e;
    t[17] = e->step_n_log;
    t[22] = e->step_n_bale;
    for (int q = 0; q < 2; ++q) {
        t[18 + 2 * q] = e->step_log_code[q];
        t[19 + 2 * q] = e->step_log_mat[q];
        t[23 + 3 * q] = e->step_bale_mat[q];
        t[24 + 3 * q] = (double)e->step_bale_n[q];
        t[25 + 3 * q] = e->step_bale_q[q];
    }
    t[29] = e->step_done;
    t[30] = e->current_step;
    t[31] = e->last_internal_press_action;
    t[32] = e->step_reward;
    for (int m = 0; m < 4; ++m) t[33 + m] = e->acc_belt[m];
    t[37] = e->step_overflow; /* info["overflow_material"] + 1 (env_monolith.py:264-268) */
}

int32_t orc_env_bales(const orc_env *e, int m, int64_t *sizes, int32_t *qs, int32_t cap)
{
    for (int k = 0; k < e->n_bales[m] && k < cap; ++k) {
        sizes[k] = e->bales[m][k].size;
        qs[k] = e->bales[m][k].q;
    }
    return e->n_bales[m];
}

/* ====================================================================================== *
 *  masked-uniform random rollout (bench.py cpu_baseline leg)
 * ====================================================================================== */

/* the workload's policy stream (same shape as the HIP rollout kernel's: key by (seed, env), murmur3 finaliser of
 * (t * odd) ^ key); only the bench's cpu_baseline leg uses it, parity never depends on it */
static uint32_t fmix32(uint32_t h)
{
    h ^= h >> 16;
    h *= 0x85EBCA6Bu;
    h ^= h >> 13;
    h *= 0xC2B2AE35u;
    h ^= h >> 16;
    return h;
}
static uint32_t policy_u32(uint64_t seed, uint64_t env_index, uint64_t t)
{
    const uint32_t s = fmix32((uint32_t)seed ^ fmix32((uint32_t)(seed >> 32) + 0x9E3779B9u));
    const uint32_t g = (uint32_t)env_index * 0x9E3779B1u + (uint32_t)(env_index >> 32) * 0xC2B2AE3Du;
    const uint32_t key = fmix32(s + g);
    const uint32_t c = (uint32_t)t * 0x85EBCA77u + (uint32_t)(t >> 32) * 0x27D4EB2Fu;
    return fmix32(c ^ key);
}

double orc_env_random_rollout(orc_env *e, int64_t n_steps, uint64_t policy_seed)
{
    float obs[32];
    uint8_t mask[32];
    double acc = 0.0;
    int A = orc_env_num_actions(e);
    for (int64_t t = 0; t < n_steps; ++t) {
        orc_env_action_mask(e, mask);
        int valid[32], n = 0;
        for (int a = 0; a < A; ++a)
            if (mask[a]) valid[n++] = a;
        uint32_t r = policy_u32(policy_seed, 0, (uint64_t)t);
        int action = valid[(int)(((uint64_t)r * (uint64_t)n) >> 32)];
        double reward;
        int32_t term;
        orc_env_step(e, action, -1, 0, obs, &reward, &term);
        acc += reward + obs[0];
        if (term) orc_env_reset(e, 0, 0, obs);
    }
    return acc;
}
