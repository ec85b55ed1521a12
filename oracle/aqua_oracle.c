/*
 * aqua_oracle.c -- CPU restatement of the reference's AquaEnv.step() hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under aquaticgymenv_amd/ or gym_aqua/ may
 * include, link, import or call this file; only tests/, __graft_entry__.smoke()
 * and bench.py's cpu_baseline leg use it, and only as the checker / the timed
 * CPU baseline -- never as the product path.
 *
 * Parity status: PINNED.  tests/test_oracle_golden.py checks every function
 * here against tests/golden/step_golden.npz and traj_golden.npz, which were
 * produced by running the reference's own gym_aqua/envs/aqua.py in the build
 * container (tests/golden/make_golden.py).  reset() is the exception: the
 * reference samples with gym's Box.sample() (un-pinned third-party RNG), so the
 * reset here is this build's own float32 specification and only its
 * DISTRIBUTION is compared with the reference (tests/golden/reset_golden.npz):
 * "reset parity unpinned".
 *
 * step(): literal float64 transcription, same operation order as the reference
 *   gym_aqua/envs/aqua.py:135-213 (step), :128-133 (normalize_angle),
 *   :373-390 (distances), :392-402,421-439 (predicates), :13-98 (constants).
 * noise: the reference draws 2 uniforms from numpy's global MT19937
 *   (aqua.py:188).  That stream cannot exist on a GPU; here the two draws are
 *   either INJECTED (noise_u, in [-1,1), multiplied by sigma) or taken from
 *   Philox4x32-10 (Salmon, Moraes, Dror, Shaw, SC'11; Random123 v1.09 known
 *   answers are checked in tests/test_oracle_golden.py).
 */
#define _GNU_SOURCE
#include <math.h>
#include <stdint.h>
#include <stddef.h>
#include <stdlib.h>
#include <string.h>

#ifdef _OPENMP
#include <omp.h>
#endif

/* ------------------------------------------------------------------ Philox */
static inline void philox4x32_10(uint32_t k0, uint32_t k1, const uint32_t ctr[4], uint32_t out[4])
{
    uint32_t c0 = ctr[0], c1 = ctr[1], c2 = ctr[2], c3 = ctr[3];
    for (int r = 0; r < 10; ++r) {
        uint64_t p0 = (uint64_t)0xD2511F53u * c0;
        uint64_t p1 = (uint64_t)0xCD9E8D57u * c2;
        uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
        uint32_t n1 = (uint32_t)p1;
        uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
        uint32_t n3 = (uint32_t)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

void aqua_oracle_philox(uint32_t k0, uint32_t k1, const uint32_t ctr[4], uint32_t out[4])
{
    philox4x32_10(k0, k1, ctr, out);
}

/* stream ids of this build's RNG specification (DESIGN.md "RNG") */
/* 0: step noise (per pair of worlds); 1: placement attempts of reset (words 0,1 = goal candidate of attempt a,
 * words 2,3 = boat candidate of attempt a); 3: heading + wave of reset; 4: sampled actions (per pair) */
enum { STREAM_STEP = 0, STREAM_PLACE = 1, STREAM_POSE = 3, STREAM_ACT = 4 };

static inline void aqua_draw(uint64_t seed, uint64_t env, uint64_t tick, uint32_t stream, uint32_t attempt,
                             uint32_t out[4])
{
    uint32_t ctr[4];
    ctr[0] = (uint32_t)env;
    ctr[1] = (uint32_t)(env >> 32);
    ctr[2] = (uint32_t)tick;
    ctr[3] = ((uint32_t)(tick >> 32) & 0xFFFFu) | ((attempt & 0xFFu) << 16) | (stream << 24);
    philox4x32_10((uint32_t)seed, (uint32_t)(seed >> 32), ctr, out);
}

/* 24-bit uniforms: exactly representable in float32 and float64 */
static inline double u_pm1(uint32_t r) { return (double)(r >> 8) * 0x1p-23 - 1.0; }   /* [-1, 1) */
static inline float u01f(uint32_t r) { return (float)(r >> 8) * 0x1p-24f; }            /* [0, 1)  */

/*
 * Step noise of world `env` at `tick`: one Philox call serves the PAIR of worlds (env >> 1); the even
 * world takes words 0,1 and the odd world words 2,3 (a lane that advances two worlds makes one call).
 * raw[] returns the two words of this world (raw[0], raw[1]) for inspection.
 */
void aqua_oracle_step_noise(uint64_t seed, uint64_t env, uint64_t tick, double out_u[2], uint32_t raw[4])
{
    uint32_t r[4];
    aqua_draw(seed, env >> 1, tick, STREAM_STEP, 0, r);
    const int h = (int)(env & 1u) * 2;
    raw[0] = r[h]; raw[1] = r[h + 1]; raw[2] = 0; raw[3] = 0;
    out_u[0] = u_pm1(r[h]);
    out_u[1] = u_pm1(r[h + 1]);
}

/* the two action words of world `env` at `tick` (device-sampled actions of the synthetic rollouts) */
static void action_words(uint64_t seed, uint64_t env, uint64_t tick, uint32_t out[2])
{
    uint32_t r[4];
    aqua_draw(seed, env >> 1, tick, STREAM_ACT, 0, r);
    const int h = (int)(env & 1u) * 2;
    out[0] = r[h]; out[1] = r[h + 1];
}

/* ------------------------------------------------------- geometry, float64 */
/* aqua.py:373-377 */
static double dist_circle_circle(double ax, double ay, double ar, double bx, double by, double br)
{
    double dx = ax - bx, dy = ay - by;
    double centers = sqrt(dx * dx + dy * dy);
    double total = ar + br;
    return centers - total;
}

static double clipd(double v, double lo, double hi) { return v < lo ? lo : (v > hi ? hi : v); }

/* aqua.py:379-390 */
static double dist_rect_circle(double rx, double ry, double rw, double rh, double cx, double cy, double cr)
{
    double l = rx - rw / 2, r = rx + rw / 2, b = ry - rh / 2, t = ry + rh / 2;
    double dx = cx - clipd(cx, l, r);
    double dy = cy - clipd(cy, b, t);
    return sqrt(dx * dx + dy * dy) - cr;
}

#define BOAT_RADIUS 2.5
#define GOAL_RADIUS 2.5
#define AXLE 2.5
#define WORLD 100.0
#define THRUST_EPS 1e-8
#define TIME_LIMIT 1000

/* obstacle rows: cx, cy, kind (0 circle, 1 rectangle), a (radius | width), b (unused | height) */
static double obstacle_distance(const double* o, double px, double py, double pr)
{
    if (o[2] == 0.0)
        return dist_circle_circle(o[0], o[1], o[3], px, py, pr);
    return dist_rect_circle(o[0], o[1], o[3], o[4], px, py, pr);
}

/* aqua.py:429-439 (first hit wins; the result is an OR so order does not matter) */
static int collided_obstacle(int K, const double* obst, double px, double py, double pr, double* min_dist)
{
    int hit = 0;
    double m = INFINITY;
    for (int k = 0; k < K; ++k) {
        double d = obstacle_distance(obst + 5 * k, px, py, pr);
        if (d < m) m = d;
        if (d <= 0) hit = 1;
    }
    if (min_dist) *min_dist = m;
    return hit;
}

/* aqua.py:424-427: strict comparisons */
static int collided_border(double px, double py, double pr)
{
    return (px - pr < 0.0) || (py - pr < 0.0) || (px + pr > WORLD) || (py + pr > WORLD);
}

/* aqua.py:128-133 with range [-pi, pi] */
static double wrap_angle(double value)
{
    double start = -M_PI, end = M_PI;
    double width = end - start;
    double off = value - start;
    return (off - (floor(off / width) * width)) + start;
}

/* one env, one step.  s = x, y, theta, gx, gy, wx, wy (updated in place). */
static void step_one(int K, const double* obst, int waves, double s[7], int32_t* time, double vl, double vr,
                     const double u[2], double* reward, uint8_t* term, double margins[3])
{
    const double wave_lo = -0.05 * waves, wave_hi = 0.05 * waves, sigma = 0.001 * waves;
    double px = s[0], py = s[1], th = s[2];
    const double gx = s[3], gy = s[4];
    const double prev_x = px, prev_y = py;
    *time += 1;                                                          /* aqua.py:141 */

    double diff = vr - vl;                                               /* aqua.py:159 */
    diff = copysign(fmax(fabs(diff), THRUST_EPS), diff);                 /* aqua.py:160 */
    double r = AXLE / 2 * (vr + vl) / diff;                              /* aqua.py:166 */
    double w = diff / AXLE;                                              /* aqua.py:168 */
    double angle = M_PI / 2 + th;                                        /* aqua.py:173 */
    double icc_x = px + r * (-sin(angle));                               /* aqua.py:174 */
    double icc_y = py + r * cos(angle);
    double c = cos(w), sn = sin(w);                                      /* aqua.py:176 (tau = 1) */
    double qx = px - icc_x, qy = py - icc_y;
    double nx = (c * qx + (-sn) * qy) + icc_x + s[5];                    /* aqua.py:180-181 */
    double ny = (sn * qx + c * qy) + icc_y + s[6];
    px = nx; py = ny;
    th = wrap_angle(th + w);                                             /* aqua.py:183 */

    s[5] = clipd(s[5] + u[0] * sigma, wave_lo, wave_hi);                 /* aqua.py:188-191 */
    s[6] = clipd(s[6] + u[1] * sigma, wave_lo, wave_hi);

    double m_obst;
    int hit_o = collided_obstacle(K, obst, px, py, BOAT_RADIUS, &m_obst);
    int hit_b = collided_border(px, py, BOAT_RADIUS);
    double d_cur = dist_circle_circle(gx, gy, GOAL_RADIUS, px, py, BOAT_RADIUS);       /* aqua.py:392-402 */
    double d_prev = dist_circle_circle(gx, gy, GOAL_RADIUS, prev_x, prev_y, BOAT_RADIUS);
    if (hit_o || hit_b) { *term = 1; *reward = -10.0; }                  /* aqua.py:200-202 */
    else if (*time > TIME_LIMIT) { *term = 2; *reward = -10.0; }         /* aqua.py:203-205 */
    else if (d_cur <= 0) { *term = 3; *reward = 10.0; }                  /* aqua.py:206-208 */
    else { *term = 0; *reward = (d_prev - d_cur) * 0.7; }                /* aqua.py:89-90,210 */
    s[0] = px; s[1] = py; s[2] = th;
    if (margins) {
        double bl = fmin(px - BOAT_RADIUS, py - BOAT_RADIUS);            /* aqua.py:404-407 */
        double tr = fmin(WORLD - (px + BOAT_RADIUS), WORLD - (py + BOAT_RADIUS));
        margins[0] = fmin(bl, tr);
        margins[1] = m_obst;
        margins[2] = d_cur;
    }
}

/* action decode: aqua.py:33-43,154 (discrete) and aqua.py:144-151 (continuous clip) */
static void decode_action(int kind, const void* action, int64_t i, int64_t n, double* vl, double* vr)
{
    static const double tab[3][2] = {{0.2, 0.5}, {0.5, 0.2}, {0.5, 0.5}};
    if (kind == 3) {
        const float* a = (const float*)action;
        double l = (double)a[i], r = (double)a[n + i];
        int inside = (l >= 0.2 && l <= 0.5 && r >= 0.2 && r <= 0.5);
        if (!inside) { l = clipd(l, 0.2, 0.5); r = clipd(r, 0.2, 0.5); }
        *vl = l; *vr = r;
        return;
    }
    int64_t a;
    if (kind == 0) a = ((const uint8_t*)action)[i];
    else if (kind == 1) a = ((const int32_t*)action)[i];
    else a = ((const int64_t*)action)[i];
    if (a < 0) a += 3;                    /* Python list index wrap-around */
    if (a < 0) a = 0;
    if (a > 2) a = 2;                     /* the reference raises IndexError; the batched build clamps */
    *vl = tab[a][0]; *vr = tab[a][1];
}

/*
 * Batched step, float64 state, SoA [7][n].  noise_u == NULL -> Philox(seed, env_offset + i, tick).
 * margins (optional) [3][n]: border, nearest obstacle, goal distance (all "<= 0 / < 0 means hit").
 */
void aqua_oracle_step(int64_t n, int K, const double* obst, int waves, double* state, int32_t* time,
                      int action_kind, const void* action, const double* noise_u, uint64_t seed, uint64_t tick,
                      int64_t env_offset, double* reward, uint8_t* term, double* margins)
{
/* (a team of threads for a handful of worlds costs a GPU box that shows 100+ cores ~0.1 s per call) */
#pragma omp parallel for schedule(static) if (n > 1024)
    for (int64_t i = 0; i < n; ++i) {
        double s[7], u[2], m[3], vl, vr;
        for (int j = 0; j < 7; ++j) s[j] = state[j * n + i];
        if (noise_u) { u[0] = noise_u[i]; u[1] = noise_u[n + i]; }
        else { uint32_t raw[4]; aqua_oracle_step_noise(seed, (uint64_t)(env_offset + i), tick, u, raw); }
        decode_action(action_kind, action, i, n, &vl, &vr);
        step_one(K, obst, waves, s, &time[i], vl, vr, u, &reward[i], &term[i], margins ? m : NULL);
        for (int j = 0; j < 7; ++j) state[j * n + i] = s[j];
        if (margins) { margins[i] = m[0]; margins[n + i] = m[1]; margins[2 * n + i] = m[2]; }
    }
}

/*
 * The same with one obstacle list PER WORLD (each env object of the reference has its own, aqua.py:13,56-68):
 * obst_all [n][K][5], rows with kind < 0 are absent.
 */
void aqua_oracle_step_tables(int64_t n, int K, const double* obst_all, int waves, double* state, int32_t* time,
                             int action_kind, const void* action, const double* noise_u, uint64_t seed, uint64_t tick,
                             int64_t env_offset, double* reward, uint8_t* term, double* margins)
{
/* (a team of threads for a handful of worlds costs a GPU box that shows 100+ cores ~0.1 s per call) */
#pragma omp parallel for schedule(static) if (n > 1024)
    for (int64_t i = 0; i < n; ++i) {
        double s[7], u[2], m[3], vl, vr, mine[64 * 5];
        int Ki = 0;
        for (int k = 0; k < K && k < 64; ++k) {
            const double* o = obst_all + (i * K + k) * 5;
            if (o[2] < 0.0) continue;
            for (int c = 0; c < 5; ++c) mine[5 * Ki + c] = o[c];
            ++Ki;
        }
        for (int j = 0; j < 7; ++j) s[j] = state[j * n + i];
        if (noise_u) { u[0] = noise_u[i]; u[1] = noise_u[n + i]; }
        else { uint32_t raw[4]; aqua_oracle_step_noise(seed, (uint64_t)(env_offset + i), tick, u, raw); }
        decode_action(action_kind, action, i, n, &vl, &vr);
        step_one(Ki, mine, waves, s, &time[i], vl, vr, u, &reward[i], &term[i], margins ? m : NULL);
        for (int j = 0; j < 7; ++j) state[j * n + i] = s[j];
        if (margins) { margins[i] = m[0]; margins[n + i] = m[1]; margins[2 * n + i] = m[2]; }
    }
}

/* ----------------------------------------------------------- reset, float32 */
/*
 * This build's reset specification (reference behaviour: aqua.py:100-126,442-455 -- rejection
 * sampling of a goal clear of border+obstacles, then of a boat pose clear of goal, border and
 * obstacles, then a wave in the wave box; time = 0).  All arithmetic is float32 with every
 * rounding explicit (fmaf or a single operation per statement) so a device implementation can be
 * bit-identical.  Positions are drawn directly in [2.5, 97.5): rejecting border hits of a draw in
 * [0, 100] gives the same distribution.  After 64 rejected attempts the fixed pose of
 * aqua.py:107,117 is used.
 */
typedef struct { float cx, cy, hx, hy, r2; } ObstF;     /* box centre + half extents (0 for a circle) */

static void obst_to_f32(int K, const double* obst, double other_radius, ObstF* t)
{
    for (int k = 0; k < K; ++k) {
        const double* o = obst + 5 * k;
        double hx = 0, hy = 0, rs = other_radius;
        if (o[2] == 0.0) rs += o[3]; else { hx = o[3] / 2; hy = o[4] / 2; }
        t[k].cx = (float)o[0]; t[k].cy = (float)o[1];
        t[k].hx = (float)hx; t[k].hy = (float)hy;
        t[k].r2 = (float)(rs * rs);
    }
}

/* distance from the point to the box, per axis max(|p - c| - h, 0); one rounding per operation */
static int hit_f32(int K, const ObstF* t, float px, float py)
{
    int hit = 0;
    for (int k = 0; k < K; ++k) {
        float ax = fabsf(px - t[k].cx), ay = fabsf(py - t[k].cy);
        float dx = fmaxf(ax - t[k].hx, 0.0f), dy = fmaxf(ay - t[k].hy, 0.0f);
        float dy2 = dy * dy;
        float d2 = fmaf(dx, dx, dy2);
        hit |= (d2 <= t[k].r2);
    }
    return hit;
}

#define RESET_TRIES 64
#define KMAX 64

static void reset_world(int K, const ObstF* t, int waves, int random_boat, int random_goal, uint64_t seed, uint64_t tick,
                        uint64_t env, float* state, int64_t ld, int64_t i, int32_t* time)
{
    const float PI_F = 3.14159274101257324f, TWO_PI_F = 6.28318548202514648f;
    uint32_t r[4];
    float gx = 25.0f, gy = 80.0f;
    if (random_goal) {
        for (uint32_t a = 0; a < RESET_TRIES; ++a) {
            aqua_draw(seed, env, tick, STREAM_PLACE, a, r);
            float cx = fmaf(95.0f, u01f(r[0]), 2.5f);
            float cy = fmaf(95.0f, u01f(r[1]), 2.5f);
            if (!hit_f32(K, t, cx, cy)) { gx = cx; gy = cy; break; }
        }
    }
    aqua_draw(seed, env, tick, STREAM_POSE, 0, r);          /* heading and wave: independent of acceptance */
    float W = 0.05f * (float)waves;
    float heading = fmaf(TWO_PI_F, u01f(r[0]), -PI_F);
    float wx = W * (float)u_pm1(r[1]);
    float wy = W * (float)u_pm1(r[2]);
    float bx = 85.0f, by = 45.0f, bt = 0.0f;
    if (random_boat) {
        for (uint32_t a = 0; a < RESET_TRIES; ++a) {
            aqua_draw(seed, env, tick, STREAM_PLACE, a, r);
            float cx = fmaf(95.0f, u01f(r[2]), 2.5f);
            float cy = fmaf(95.0f, u01f(r[3]), 2.5f);
            float ex = gx - cx, ey = gy - cy;
            float ey2 = ey * ey;
            float g2 = fmaf(ex, ex, ey2);
            if (g2 <= 25.0f) continue;                       /* on the goal: aqua.py:112 */
            if (hit_f32(K, t, cx, cy)) continue;
            bx = cx; by = cy; bt = heading;
            break;
        }
    }
    state[0 * ld + i] = bx; state[1 * ld + i] = by; state[2 * ld + i] = bt;
    state[3 * ld + i] = gx; state[4 * ld + i] = gy;
    state[5 * ld + i] = wx; state[6 * ld + i] = wy;
    time[i] = 0;
}

void aqua_oracle_reset(int64_t n, int K, const double* obst, int waves, int random_boat, int random_goal,
                       uint64_t seed, uint64_t tick, int64_t env_offset, const uint8_t* mask, float* state,
                       int64_t ld, int32_t* time)
{
    ObstF t[KMAX];
    if (K > KMAX) K = KMAX;
    obst_to_f32(K, obst, 2.5, t);           /* goal radius == boat radius == 2.5 (aqua.py:72,75) */
#pragma omp parallel for schedule(static) if (n > 4096)
    for (int64_t i = 0; i < n; ++i) {
        if (mask && !mask[i]) continue;
        reset_world(K, t, waves, random_boat, random_goal, seed, tick, (uint64_t)(env_offset + i), state, ld, i, time);
    }
}

/* one obstacle list per world: obst_all [n][K][5], rows with kind < 0 are absent */
void aqua_oracle_reset_tables(int64_t n, int K, const double* obst_all, int waves, int random_boat, int random_goal,
                              uint64_t seed, uint64_t tick, int64_t env_offset, const uint8_t* mask, float* state,
                              int64_t ld, int32_t* time)
{
    if (K > KMAX) K = KMAX;
#pragma omp parallel for schedule(static) if (n > 4096)
    for (int64_t i = 0; i < n; ++i) {
        if (mask && !mask[i]) continue;
        double mine[KMAX * 5];
        ObstF t[KMAX];
        int Ki = 0;
        for (int k = 0; k < K; ++k) {
            const double* o = obst_all + (i * K + k) * 5;
            if (o[2] < 0.0) continue;
            for (int c = 0; c < 5; ++c) mine[5 * Ki + c] = o[c];
            ++Ki;
        }
        obst_to_f32(Ki, mine, 2.5, t);
        reset_world(Ki, t, waves, random_boat, random_goal, seed, tick, (uint64_t)(env_offset + i), state, ld, i, time);
    }
}

/* ------------------------------------------------- float32-state rollout */
/*
 * The batched build stores its state in float32.  This drives step_one() from float32 storage
 * (state widened exactly to float64, result rounded to float32) with Philox noise, on-spec action
 * sampling and auto-reset; it is the CPU baseline timed by bench.py and the multi-step checker.
 * actions == NULL -> actions are sampled from Philox stream 4 (action_words) as the device
 * rollout does.  Returns the number of finished episodes.
 */
int64_t aqua_oracle_rollout_f32(int64_t n, int K, const double* obst, int waves, int continuous, float* state,
                                int64_t ld, int32_t* time, int64_t steps, const void* actions, int action_kind,
                                uint64_t seed, uint64_t tick0, int64_t env_offset, int auto_reset, float* reward,
                                uint8_t* term, int64_t* term_counts)
{
    int64_t episodes = 0, c1 = 0, c2 = 0, c3 = 0;
    for (int64_t t = 0; t < steps; ++t) {
        uint64_t tick = tick0 + (uint64_t)t;
        /* auto_reset == 2 ("next-step").  Markers in the time row, with the parity of the tick that wrote them:
         *   -1 - (t & 1)  finished at tick t, waits for its restart;  -3 - (t & 1)  restarted during tick t.
         * At this tick: worlds marked "finished at tick - 1" are re-initialised and marked "restarted at tick";
         * worlds marked "restarted at tick - 1" step from time 0; every other marker does not move.  Marked
         * worlds that do not step report reward 0, term 0. */
        uint8_t* pending = NULL;
        uint8_t* restart = NULL;
        if (auto_reset == 2) {
            const int32_t fresh = -3 - (int32_t)((tick - 1) & 1u), finished = -1 - (int32_t)((tick - 1) & 1u);
            pending = (uint8_t*)malloc((size_t)n);
            restart = (uint8_t*)malloc((size_t)n);
            for (int64_t i = 0; i < n; ++i) {
                if (time[i] == fresh) time[i] = 0;
                restart[i] = time[i] == finished;
                pending[i] = time[i] < 0;
            }
        }
#pragma omp parallel for schedule(static) reduction(+ : episodes, c1, c2, c3)
        for (int64_t i = 0; i < n; ++i) {
            double s[7], u[2], vl, vr, rew;
            uint32_t raw[4], aw[2];
            uint8_t code;
            if (pending && pending[i]) { reward[i] = 0.0f; term[i] = 0; continue; }
            for (int j = 0; j < 7; ++j) s[j] = (double)state[j * ld + i];
            aqua_oracle_step_noise(seed, (uint64_t)(env_offset + i), tick, u, raw);
            if (actions) {
                size_t esz = action_kind == 0 ? 1 : action_kind == 1 ? 4 : action_kind == 2 ? 8 : 8;
                const char* base = (const char*)actions + (size_t)t * (size_t)n * esz;
                decode_action(action_kind, base, i, n, &vl, &vr);
            } else if (continuous) {
                action_words(seed, (uint64_t)(env_offset + i), tick, aw);
                float fl = fmaf(0.3f, u01f(aw[0]), 0.2f), fr = fmaf(0.3f, u01f(aw[1]), 0.2f);
                vl = clipd((double)fl, 0.2, 0.5); vr = clipd((double)fr, 0.2, 0.5);
            } else {
                static const double tab[3][2] = {{0.2, 0.5}, {0.5, 0.2}, {0.5, 0.5}};
                action_words(seed, (uint64_t)(env_offset + i), tick, aw);
                uint32_t a = (uint32_t)(((uint64_t)(aw[0] >> 8) * 3u) >> 24);
                vl = tab[a][0]; vr = tab[a][1];
            }
            step_one(K, obst, waves, s, &time[i], vl, vr, u, &rew, &code, NULL);
            for (int j = 0; j < 7; ++j) state[j * ld + i] = (float)s[j];
            reward[i] = (float)rew;
            term[i] = code;
            if (code) { episodes++; c1 += code == 1; c2 += code == 2; c3 += code == 3; }
            if (code && auto_reset == 2) {
                /* pending: goal stays, pose/wave are the terminal ones.  The marker carries the parity of the
                 * tick that finished the world (-1 even, -2 odd), as the device writes it. */
                time[i] = -1 - (int32_t)(tick & 1u);
            }
        }
        if (auto_reset == 1)
            aqua_oracle_reset(n, K, obst, waves, 1, 1, seed, tick, env_offset, term, state, ld, time);
        if (auto_reset == 2) {
            aqua_oracle_reset(n, K, obst, waves, 1, 1, seed, tick, env_offset, restart, state, ld, time);
            for (int64_t i = 0; i < n; ++i)
                if (restart[i]) time[i] = -3 - (int32_t)(tick & 1u);
            free(pending);
            free(restart);
        }
    }
    if (term_counts) { term_counts[0] = c1; term_counts[1] = c2; term_counts[2] = c3; }
    return episodes;
}

int aqua_oracle_threads(void)
{
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}
