// aqua_device.hpp -- device-side arithmetic of the batched AquaEnv step()/reset() for gfx950.
//
// Two paths per world-step (DESIGN.md "Numerics"):
//   fast  : float32, chord form of the differential-drive update (algebraically identical to the
//           reference's rotate-about-ICC form, gym_aqua/envs/aqua.py:159-183, but free of its
//           1.25e8-radius cancellation), squared-distance collision tests against thresholds that
//           bracket each decision by +-BAND.
//   exact : float64, the reference's own operation order (aqua.py:159-211), taken only by worlds
//           whose fast-path margin to ANY threshold (border, obstacle, goal radius) is inside the
//           band, so that every boolean the kernel reports is the one the reference would report.
// Nothing here is shared with oracle/: the oracle is an independent restatement.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace aqua {

constexpr float BAND = 1.0e-4f;          // half-width of the knife-edge band, in world units
constexpr int RESET_TRIES = 64;
constexpr int MAX_OBST = 64;

// one obstacle in fast-path form: a box (a circle is a box with zero half extents) inflated by R
struct ObstF {
    float cx, cy, hx, hy;       // box centre and half extents (aqua.py:381-384); circle: hx = hy = 0
    float a, b;                 // a = 1 / (2R), b = -R / 2 with R = obstacle radius + 2.5 (circle) or 2.5 (rect):
                                //   m = a * d^2 + b = (d^2 - R^2) / (2R) has the sign of (d - R) and ~ its size near 0
    float r2;                   // R^2, used by the float32 reset specification
    float pad;
};
static_assert(sizeof(ObstF) == 32, "ObstF is two float4");

// Where the per-batch obstacle table is read from.  Default: straight from the packed blob through the
// CONSTANT address space -- every lane reads the same row, so the loads are scalar (s_load_dwordx4 into
// SGPRs, served by the scalar cache) and the rows cost no VGPRs, no LDS round trip and no barrier.
// -DAQUA_OBST_LDS=1 builds the variant that stages the table into LDS once per workgroup instead
// (kept for A/B measurements and as the base of per-world tables; DESIGN.md "Obstacle table").
#ifndef AQUA_OBST_LDS
#define AQUA_OBST_LDS 0
#endif
#if AQUA_OBST_LDS
using ObstPtr = const ObstF*;
#else
using ObstPtr = const ObstF __attribute__((address_space(4)))*;
#endif

struct EnvState {               // registers of one world
    float x, y, th, gx, gy, wx, wy;
    int t;
};

struct StepConst {              // wave-uniform
    float W, sigma;             // wave bound 0.05*waves, wave step 0.001*waves (aqua.py:23-25)
    int waves;
    int time_limit;             // aqua.py:91
    int K;
    ObstPtr obst;               // float32 table (scalar-loaded from the blob, or the LDS copy)
    const double* obst64;       // global float64 rows [K][5] (exact path)
};

// Discrete action table (aqua.py:33-42) folded through aqua.py:159-170 in double and rounded once:
//   action 0 (0.2, 0.5): w = +0.12, 1 (0.5, 0.2): w = -0.12, 2 (0.5, 0.5): w = 1e-8 / 2.5 (the epsilon
//   sentinel of aqua.py:160); h = w / 2; chord = v * sin(h) / h.  Literals (not kernel arguments) so the
//   per-lane choice is three v_cndmask on immediates; aqua_discrete_constants() exports them and
//   tests/test_capi_cpu.py re-derives them.
constexpr float ACT_H_TURN = 0x1.eb851ep-5f, ACT_W_TURN = 0x1.eb851ep-4f, ACT_C_TURN = 0x1.662f5cp-2f;
constexpr float ACT_H_LINE = 0x1.12e0bep-29f, ACT_W_LINE = 0x1.12e0bep-28f, ACT_C_LINE = 0.5f;

// ------------------------------------------------------------------------------------ Philox
// Philox4x32-10 (Salmon et al., SC'11).  key = seed; counter = (env lo, env hi, tick lo,
// tick hi[15:0] | attempt << 16 | stream << 24).
// Streams 0 (step noise) and 4 (sampled actions) are drawn per PAIR of worlds: counter env = world >> 1,
// the even world uses words 0,1 and the odd world words 2,3 -- a lane that owns two worlds makes one call.
enum : uint32_t { STREAM_STEP = 0, STREAM_GOAL = 1, STREAM_BOAT = 2, STREAM_WAVE = 3, STREAM_ACT = 4 };

template <bool SCALAR_KEY = false>
__device__ __forceinline__ void philox4x32_10(uint32_t k0, uint32_t k1, uint32_t c0, uint32_t c1, uint32_t c2,
                                               uint32_t c3, uint32_t (&out)[4])
{
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const uint32_t hi0 = __umulhi(0xD2511F53u, c0), lo0 = 0xD2511F53u * c0;
        const uint32_t hi1 = __umulhi(0xCD9E8D57u, c2), lo1 = 0xCD9E8D57u * c2;
        const uint32_t n0 = hi1 ^ c1 ^ k0;
        const uint32_t n2 = hi0 ^ c3 ^ k1;
        c0 = n0; c1 = lo1; c2 = n2; c3 = lo0;
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
        // keep the key schedule a running pair of scalars: without this the ten round keys are
        // hoisted into twenty long-lived SGPRs and spill into VGPR lanes (v_readlane per round)
        if constexpr (SCALAR_KEY) asm volatile("" : "+s"(k0), "+s"(k1));
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

// SCALAR_KEY: the caller guarantees `seed` is wave-uniform and lives in SGPRs (kernel arguments)
template <bool SCALAR_KEY = false>
__device__ __forceinline__ void draw(uint64_t seed, uint64_t env, uint64_t tick, uint32_t stream, uint32_t attempt,
                                      uint32_t (&out)[4])
{
    const uint32_t c3 = (static_cast<uint32_t>(tick >> 32) & 0xFFFFu) | ((attempt & 0xFFu) << 16) | (stream << 24);
    philox4x32_10<SCALAR_KEY>(static_cast<uint32_t>(seed), static_cast<uint32_t>(seed >> 32), static_cast<uint32_t>(env),
                  static_cast<uint32_t>(env >> 32), static_cast<uint32_t>(tick), c3, out);
}

// 24-bit uniforms, exact in float32
__device__ __forceinline__ float u_pm1(uint32_t r)     // [-1, 1)
{
    return static_cast<float>(static_cast<int>(r >> 8) - 0x800000) * 0x1p-23f;
}
__device__ __forceinline__ float u_01(uint32_t r)      // [0, 1)
{
    return static_cast<float>(r >> 8) * 0x1p-24f;
}

// ------------------------------------------------------------------------------------ math
// sin and cos of a bounded argument (|a| <= pi + 0.06 in normal operation): Cody-Waite reduction
// by pi/2 and degree-7/8 minimax polynomials; abs error < 1.2e-7.  Huge arguments (only possible
// when a caller writes a wild theta into the state) go to the library routine.
__device__ __forceinline__ void sincos_bounded(float a, float& s, float& c)
{
    if (__builtin_expect(fabsf(a) > 64.0f, 0)) { s = sinf(a); c = cosf(a); return; }
    const float k = rintf(a * 0.636619772367581343f);
    float r = fmaf(k, -1.5707962512969971f, a);            // pi/2 hi
    r = fmaf(k, -7.5497894158615964e-08f, r);              // pi/2 lo
    const float r2 = r * r;
    float sp = fmaf(r2, -1.9515295891e-4f, 8.3321608736e-3f);
    sp = fmaf(sp, r2, -1.6666654611e-1f);
    const float sr = fmaf(r * r2, sp, r);
    float cp = fmaf(r2, 2.443315711809948e-5f, -1.388731625493765e-3f);
    cp = fmaf(cp, r2, 4.166664568298827e-2f);
    const float cr = fmaf(r2 * r2, cp, fmaf(r2, -0.5f, 1.0f));
    const int q = static_cast<int>(k) & 3;
    const float s0 = (q & 1) ? cr : sr;
    const float c0 = (q & 1) ? sr : cr;
    s = (q & 2) ? -s0 : s0;
    c = ((q + 1) & 2) ? -c0 : c0;
}

// theta <- theta + w folded into [-pi, pi) (aqua.py:128-133).  The float32 neighbours of +-pi are
// classified exactly as the float64 reference classifies them.
__device__ __forceinline__ float wrap_add(float th, float w)
{
    constexpr float PI_F = 3.14159274101257324f;
    constexpr float TWO_PI_HI = 6.28318548202514648f, TWO_PI_LO = -1.74845553146951715e-07f;
    float s = th + w;
    const float turns = s >= PI_F ? -1.0f : (s <= -PI_F ? 1.0f : 0.0f);
    s = fmaf(turns, TWO_PI_HI, s);           // exact (Sterbenz range), then the low part of 2 pi
    return fmaf(turns, TWO_PI_LO, s);
}

// continuous thrusts -> (h, w, chord) (aqua.py:159-170 with r eliminated: chord = 2 r sin(w/2) = v sinc(w/2))
__device__ __forceinline__ void thrust_to_motion(float vl, float vr, float& h, float& w, float& chord)
{
    float d = vr - vl;
    d = copysignf(fmaxf(fabsf(d), 1.0e-8f), d);            // aqua.py:160 (+0.0 keeps the + sign)
    w = d * 0.4f;                                          // d / 2.5
    h = 0.5f * w;
    const float v = 0.5f * (vl + vr);
    const float h2 = h * h;
    const float sinc = fmaf(h2, fmaf(h2, fmaf(h2, -1.984126984e-4f, 8.333333333e-3f), -1.666666667e-1f), 1.0f);
    chord = v * sinc;
}

// ------------------------------------------------------------------------------------ exact path
// float64, operation order of the reference (aqua.py:159-211).  Contraction is off so that
// products and sums round where the reference's do.
struct ExactOut { float x, y, th, reward; uint32_t term; };

__device__ __noinline__ ExactOut exact_step(float fx, float fy, float fth, float fgx, float fgy, float fwx,
                                            float fwy, int t_new, double vl, double vr, int K,
                                            const double* __restrict__ obst64, int time_limit)
{
#pragma clang fp contract(off)
    constexpr double PI_D = 3.141592653589793;
    const double px = fx, py = fy, th = fth, gx = fgx, gy = fgy;
    double diff = vr - vl;
    diff = copysign(fmax(fabs(diff), 1e-8), diff);
    const double r = 2.5 / 2 * (vr + vl) / diff;
    const double w = diff / 2.5;
    const double angle = PI_D / 2 + th;
    const double icc_x = px + r * (-sin(angle));
    const double icc_y = py + r * cos(angle);
    const double c = cos(w), s = sin(w);
    const double qx = px - icc_x, qy = py - icc_y;
    const double nx = (c * qx + (-s) * qy) + icc_x + static_cast<double>(fwx);
    const double ny = (s * qx + c * qy) + icc_y + static_cast<double>(fwy);
    const double width = PI_D - (-PI_D);
    const double off = (th + w) - (-PI_D);
    const double nth = (off - (floor(off / width) * width)) + (-PI_D);

    bool hit = (nx - 2.5 < 0.0) || (ny - 2.5 < 0.0) || (nx + 2.5 > 100.0) || (ny + 2.5 > 100.0);
    for (int k = 0; k < K; ++k) {
        const double* o = obst64 + 5 * k;
        double dist;
        if (o[2] == 0.0) {
            const double dx = o[0] - nx, dy = o[1] - ny;
            dist = sqrt(dx * dx + dy * dy) - (o[3] + 2.5);
        } else {
            const double l = o[0] - o[3] / 2, rr = o[0] + o[3] / 2, b = o[1] - o[4] / 2, tt = o[1] + o[4] / 2;
            const double cx = nx < l ? l : (nx > rr ? rr : nx);
            const double cy = ny < b ? b : (ny > tt ? tt : ny);
            const double dx = nx - cx, dy = ny - cy;
            dist = sqrt(dx * dx + dy * dy) - 2.5;
        }
        hit = hit || (dist <= 0.0);
    }
    const double ex = gx - nx, ey = gy - ny;
    const double d_cur = sqrt(ex * ex + ey * ey) - (2.5 + 2.5);
    const double ox = gx - px, oy = gy - py;
    const double d_prev = sqrt(ox * ox + oy * oy) - (2.5 + 2.5);
    ExactOut out;
    if (hit) { out.term = 1; out.reward = -10.0f; }
    else if (t_new > time_limit) { out.term = 2; out.reward = -10.0f; }
    else if (d_cur <= 0.0) { out.term = 3; out.reward = 10.0f; }
    else { out.term = 0; out.reward = static_cast<float>((d_prev - d_cur) * 0.7); }
    out.x = static_cast<float>(nx);
    out.y = static_cast<float>(ny);
    out.th = static_cast<float>(nth);
    return out;
}

// ------------------------------------------------------------------------------------ fast path
// Advances one world in float32.  Returns true when a margin falls inside the knife-edge band
// (the caller then overrides pose/reward/term with exact_step()).
//
// Collision is decided from ONE running minimum of signed margins: the border margin
// min(x, y) - 2.5, 97.5 - max(x, y) (aqua.py:424-427) and, per obstacle, m = (d^2 - R^2) / (2R) with d the
// distance from the boat centre to the obstacle's box (aqua.py:373-390, 429-439).  m has exactly the sign
// of d - R and |m| >= |d - R| / 2, so outside the band the sign of the minimum IS the reference's
// OR-of-tests, and the strict/non-strict difference of the reference's comparisons only matters at
// margin == 0, which is inside the band by construction.  ~10 VALU operations per obstacle, no compares.
__device__ __forceinline__ bool fast_step(EnvState& e, float h, float w, float chord, float u0, float u1,
                                          const StepConst& k, float& reward, uint32_t& term)
{
    float s, c;
    sincos_bounded(e.th + h, s, c);
    const float ddx = fmaf(-chord, s, e.wx);               // displacement incl. wave drift (old wave, aqua.py:180-181)
    const float ddy = fmaf(chord, c, e.wy);
    const float xn = e.x + ddx, yn = e.y + ddy;
    const float thn = wrap_add(e.th, w);
    // wave random walk (aqua.py:188-191): drawn after the move
    const float wxn = __builtin_amdgcn_fmed3f(fmaf(u0, k.sigma, e.wx), -k.W, k.W);
    const float wyn = __builtin_amdgcn_fmed3f(fmaf(u1, k.sigma, e.wy), -k.W, k.W);
    const int tn = e.t + 1;                                // aqua.py:141

    float mc = fminf(fminf(xn, yn) - 2.5f, 97.5f - fmaxf(xn, yn));
#pragma unroll 2
    for (int j = 0; j < k.K; ++j) {
        const float dx = fmaxf(fabsf(xn - k.obst[j].cx) - k.obst[j].hx, 0.0f);
        const float dy = fmaxf(fabsf(yn - k.obst[j].cy) - k.obst[j].hy, 0.0f);
        const float d2 = fmaf(dx, dx, dy * dy);
        mc = fminf(mc, fmaf(d2, k.obst[j].a, k.obst[j].b));
    }
    // goal distance and shaped reward (aqua.py:89-90, 392-402, 421-422).  prev - cur is formed
    // from the displacement, (|a|^2 - |b|^2) / (|a| + |b|), not as a difference of two norms.
    const float ex = e.gx - e.x, ey = e.gy - e.y;
    const float fx = ex - ddx, fy = ey - ddy;
    const float dprev = __builtin_amdgcn_sqrtf(fmaf(ex, ex, ey * ey));       // v_sqrt_f32, 1 ulp
    const float dcur = __builtin_amdgcn_sqrtf(fmaf(fx, fx, fy * fy));
    const float mg = dcur - 5.0f;
    const float dsum = dprev + dcur;
    const float num = fmaf(ddx, ex + fx, ddy * (ey + fy));
    const float shaped = dsum > 0.0f ? 0.7f * (num * __builtin_amdgcn_rcpf(dsum)) : 0.0f;

    const bool knife = fminf(fabsf(mc), fabsf(mg)) < BAND;
    term = mc < 0.0f ? 1u : (tn > k.time_limit ? 2u : (mg <= 0.0f ? 3u : 0u));   // aqua.py:200-211
    reward = term == 0u ? shaped : (term == 3u ? 10.0f : -10.0f);
    e.x = xn; e.y = yn; e.th = thn; e.wx = wxn; e.wy = wyn; e.t = tn;
    return knife;
}

// ------------------------------------------------------------------------------------ reset
// float32 specification shared with nothing: oracle/aqua_oracle.c restates it independently and the
// two are compared bit for bit.  Every rounding is explicit (fmaf or one operation per statement).
__device__ __forceinline__ bool reset_hit(int K, ObstPtr t, float px, float py)
{
#pragma clang fp contract(off)
    bool hit = false;
#pragma unroll 4
    for (int j = 0; j < K; ++j) {
        const float ax = fabsf(px - t[j].cx), ay = fabsf(py - t[j].cy);
        const float dx = fmaxf(ax - t[j].hx, 0.0f), dy = fmaxf(ay - t[j].hy, 0.0f);
        const float dy2 = dy * dy;
        const float d2 = fmaf(dx, dx, dy2);
        hit |= d2 <= t[j].r2;
    }
    return hit;
}

__device__ __noinline__ EnvState reset_env(uint64_t seed, uint64_t env, uint64_t tick, int waves, int random_boat,
                                           int random_goal, int K, ObstPtr t)
{
#pragma clang fp contract(off)
    EnvState e;
    constexpr float PI_F = 3.14159274101257324f, TWO_PI_F = 6.28318548202514648f;
    uint32_t r[4];
    float gx = 25.0f, gy = 80.0f;                          // aqua.py:107
    if (random_goal) {
        for (uint32_t a = 0; a < RESET_TRIES; ++a) {       // aqua.py:103-105
            draw(seed, env, tick, STREAM_GOAL, a, r);
            const float cx = fmaf(95.0f, u_01(r[0]), 2.5f);
            const float cy = fmaf(95.0f, u_01(r[1]), 2.5f);
            if (!reset_hit(K, t, cx, cy)) { gx = cx; gy = cy; break; }
        }
    }
    float bx = 85.0f, by = 45.0f, bt = 0.0f;               // aqua.py:117
    if (random_boat) {
        for (uint32_t a = 0; a < RESET_TRIES; ++a) {       // aqua.py:111-115
            draw(seed, env, tick, STREAM_BOAT, a, r);
            const float cx = fmaf(95.0f, u_01(r[0]), 2.5f);
            const float cy = fmaf(95.0f, u_01(r[1]), 2.5f);
            const float ct = fmaf(TWO_PI_F, u_01(r[2]), -PI_F);
            const float ex = gx - cx, ey = gy - cy;
            const float ey2 = ey * ey;
            const float g2 = fmaf(ex, ex, ey2);
            if (g2 <= 25.0f) continue;
            if (reset_hit(K, t, cx, cy)) continue;
            bx = cx; by = cy; bt = ct;
            break;
        }
    }
    draw(seed, env, tick, STREAM_WAVE, 0, r);              // aqua.py:124
    const float W = 0.05f * static_cast<float>(waves);
    e.x = bx; e.y = by; e.th = bt; e.gx = gx; e.gy = gy;
    e.wx = W * u_pm1(r[0]);
    e.wy = W * u_pm1(r[1]);
    e.t = 0;                                               // aqua.py:125
    return e;
}

// The same specification with the attempts of ONE world spread over a group of G adjacent lanes:
// lane `sub` of the group evaluates attempts sub, sub + G, ... and the group keeps the lowest accepted
// attempt (ballot + find-first), which is exactly the attempt the serial loop above stops at.  The
// three Philox chains (goal, boat, wave) of the first round are independent and interleave.  Must be
// called by all 64 lanes of a wavefront together; `active` says whether this lane's group has a world.
// Every lane of a group returns the group's result.
template <int G>
__device__ __noinline__ EnvState reset_env_group(bool active, uint64_t seed, uint64_t env, uint64_t tick, int waves,
                                                     int random_boat, int random_goal, int K, ObstPtr t)
{
#pragma clang fp contract(off)
    static_assert(G >= 2 && G <= 64 && (G & (G - 1)) == 0, "group size");
    constexpr float PI_F = 3.14159274101257324f, TWO_PI_F = 6.28318548202514648f;
    const int lane = static_cast<int>(threadIdx.x) & 63;
    const int sub = lane & (G - 1), gbase = lane & ~(G - 1);
    const uint64_t gmask = (G == 64) ? ~0ull : ((1ull << G) - 1ull);
    uint32_t rg[4], rb[4], rw[4];
    draw(seed, env, tick, STREAM_GOAL, static_cast<uint32_t>(sub), rg);
    draw(seed, env, tick, STREAM_BOAT, static_cast<uint32_t>(sub), rb);
    draw(seed, env, tick, STREAM_WAVE, 0, rw);

    float gx = 25.0f, gy = 80.0f;
    if (random_goal) {
        bool found = !active;
        for (uint32_t base = 0; base < RESET_TRIES; base += G) {
            if (base != 0) draw(seed, env, tick, STREAM_GOAL, base + sub, rg);
            const float cx = fmaf(95.0f, u_01(rg[0]), 2.5f);
            const float cy = fmaf(95.0f, u_01(rg[1]), 2.5f);
            const bool ok = !found && !reset_hit(K, t, cx, cy);
            const uint64_t mine = (__ballot(ok) >> gbase) & gmask;
            const int src = mine ? gbase + __builtin_ctzll(mine) : lane;
            const float sx = __shfl(cx, src), sy = __shfl(cy, src);
            if (!found && mine) { gx = sx; gy = sy; found = true; }
            if (!__any(!found)) break;
        }
    }
    float bx = 85.0f, by = 45.0f, bt = 0.0f;
    if (random_boat) {
        bool found = !active;
        for (uint32_t base = 0; base < RESET_TRIES; base += G) {
            if (base != 0) draw(seed, env, tick, STREAM_BOAT, base + sub, rb);
            const float cx = fmaf(95.0f, u_01(rb[0]), 2.5f);
            const float cy = fmaf(95.0f, u_01(rb[1]), 2.5f);
            const float ct = fmaf(TWO_PI_F, u_01(rb[2]), -PI_F);
            const float ex = gx - cx, ey = gy - cy;
            const float ey2 = ey * ey;
            const float g2 = fmaf(ex, ex, ey2);
            const bool ok = !found && !(g2 <= 25.0f) && !reset_hit(K, t, cx, cy);
            const uint64_t mine = (__ballot(ok) >> gbase) & gmask;
            const int src = mine ? gbase + __builtin_ctzll(mine) : lane;
            const float sx = __shfl(cx, src), sy = __shfl(cy, src), st = __shfl(ct, src);
            if (!found && mine) { bx = sx; by = sy; bt = st; found = true; }
            if (!__any(!found)) break;
        }
    }
    const float W = 0.05f * static_cast<float>(waves);
    EnvState e;
    e.x = bx; e.y = by; e.th = bt; e.gx = gx; e.gy = gy;
    e.wx = W * u_pm1(rw[0]);
    e.wy = W * u_pm1(rw[1]);
    e.t = 0;
    return e;
}

}  // namespace aqua
