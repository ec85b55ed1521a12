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

constexpr float BAND = 1.0e-4f;          // half-width of the knife-edge band of the plain float32 margins, in world units
// ... of the error-compensated margins (second look, see fast_step).  Border and obstacles: what is left after the
// compensation is the error of the displacement (<= 3.3e-7: hardware sin/cos 3.5e-7 x chord 0.5, the rounded
// chord and angle, one fma, and the reference's own float64 cancellation on straight moves) plus, for obstacles,
// the rounding of d^2 (covered by the 4 ulp(R^2) term of the per-obstacle band).  The goal distance goes through
// two more roundings at magnitude 5 and a hardware sqrt (bound 1.2e-6, largest seen 1.3e-7): it keeps 4e-6.
constexpr float BAND_TIGHT = 2.0e-6f;
constexpr float BAND_TIGHT_GOAL = 4.0e-6f;
constexpr int RESET_TRIES = 64;
constexpr int MAX_OBST = 64;

// one obstacle in fast-path form: a box (a circle is a box with zero half extents) inflated by R.
// The packed table lists the CIRCLES FIRST (header.n_circles of them): their per-axis distance needs no
// |.| - h, max(., 0) (the generic formula with h = 0 gives the same bits).
struct ObstF {
    float cx, cy, hx, hy;       // box centre and half extents (aqua.py:381-384); circle: hx = hy = 0
    float r2;                   // R^2 with R = obstacle radius + 2.5 (circle) or 2.5 (rect)
    float w;                    // band2_tight(R_max) / band2_tight(R): scales this obstacle's compensated margin so
                                // that ONE threshold (the header's) gives every obstacle the band of its own radius
    float pad[2];
};
static_assert(sizeof(ObstF) == 32, "ObstF is two float4");

// first 32 bytes of the packed blob
struct ObstHeader {
    int32_t n_obstacles, n_circles;
    float band2;                // knife-edge band for the SQUARED margin d^2 - R^2: 2.5 * R_max * BAND
    float r_max;
    float band2_tight;          // the same for the compensated margins: 2.5 R_max BAND_TIGHT + 4 ulp(R_max^2)
    int32_t reserved[3];
};
static_assert(sizeof(ObstHeader) == 32, "ObstHeader");

// Quick table (tables of at most QUICK_MAX rows; behind the float64 rows, 64-byte aligned): the operands of the first
// look once more, struct-of-arrays in groups of four obstacles -- circles {cx[4] cy[4] -r2[4]}, rectangles
// {cx[4] cy[4] hx[4] hy[4] -r2[4]} -- so that a group is three (five) s_load_dwordx4 at constant offsets from the blob
// and its SGPR pairs feed v_pk_* operations as they are.  Walking the ObstF rows instead costs a 64-bit address per
// row, three loads per row and SGPR shuffles to pair the operands: ~110 scalar instructions and four or five
// scalar-memory round trips per wavefront for 4 circles + 4 rectangles, on a path whose scalar issue is as loaded as
// its vector issue.  Unused slots hold -r2 = 1e30: their margin d^2 + 1e30 never is the
// minimum, is never inside a band and is never negative.  Same operations in the same order per obstacle as the row
// loops: the first look's margins have the same bits either way.
constexpr int QUICK_MAX = 8;
constexpr float QUICK_EMPTY_NR2 = 1.0e30f;
typedef float f32x4 __attribute__((ext_vector_type(4)));
using QuickPtr = const f32x4 __attribute__((address_space(4)))*;
// in f32x4 units from the start of the quick table: circle groups c0 c1 (64 B each, 3 vectors used), rectangle
// groups r0 r1 (128 B each, 5 vectors used)
constexpr int QUICK_C0 = 0, QUICK_C1 = 4, QUICK_R0 = 8, QUICK_R1 = 16, QUICK_VECS = 24;
__host__ __device__ constexpr size_t quick_offset(int K)
{
    return (32u + 72u * static_cast<size_t>(K) + 63u) & ~static_cast<size_t>(63);
}
struct QuickCircles { f32x4 cx, cy, nr2; };                 // nr2 = -R^2: the fma's addend as it is
struct QuickRects { f32x4 cx, cy, hx, hy, nr2; };
__device__ __forceinline__ QuickCircles quick_circles(QuickPtr q, int at) { return QuickCircles{q[at], q[at + 1], q[at + 2]}; }
__device__ __forceinline__ QuickRects quick_rects(QuickPtr q, int at)
{
    return QuickRects{q[at], q[at + 1], q[at + 2], q[at + 3], q[at + 4]};
}

// The per-batch obstacle table is read straight from the packed blob through the CONSTANT address space: every lane
// reads the same row, so the loads are scalar (s_load_dwordx4 into SGPRs, served by the scalar cache) and the rows cost
// no VGPRs, no LDS round trip and no barrier (staging the table in LDS per workgroup measured 0.6-1.5 us per step
// slower; the re-seeding blocks of the next-step kernel, which wait on LDS anyway, do use an LDS copy).
using ObstPtr = const ObstF __attribute__((address_space(4)))*;

struct EnvState {               // registers of one world
    float x, y, th, gx, gy, wx, wy;
    int t;
};

struct StepConst {              // wave-uniform
    float W, sigma;             // wave bound 0.05*waves, wave step 0.001*waves (aqua.py:23-25)
    int waves;
    int time_limit;             // aqua.py:91
    int K, Kc;                  // obstacles, of which the first Kc are circles
    float band2, band2_tight;
    uint32_t touch[4];          // one word per table cache line (see make_const): dependencies, not data
    ObstPtr obst;               // float32 table (scalar-loaded from the blob, or the LDS copy)
    QuickPtr quick;             // quick table (K <= QUICK_MAX), else unused
    QuickCircles qc0;           // its first circle / rectangle group, loaded with the header (fast_step<.., QUICK>)
    QuickRects qr0;
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
// Stream 1 holds the placement attempts of a reset: words 0,1 = goal candidate of attempt a, words 2,3 = boat
// candidate of attempt a.  Stream 3 (attempt 0): heading, wave x, wave y.
enum : uint32_t { STREAM_STEP = 0, STREAM_PLACE = 1, STREAM_POSE = 3, STREAM_ACT = 4 };
constexpr int PHILOX_ROUNDS = 10;

template <bool SCALAR_KEY = false>
__device__ __forceinline__ void philox4x32_10(uint32_t k0, uint32_t k1, uint32_t c0, uint32_t c1, uint32_t c2,
                                               uint32_t c3, uint32_t (&out)[4])
{
    // fully unrolled: as a loop of 1 / 2 / 5 round bodies the launch is 0.2-0.5 us slower (independent chains no longer
    // interleave)
#pragma unroll
    for (int r = 0; r < PHILOX_ROUNDS; ++r) {
        // one 32x32->64 product per half round (v_mad_u64_u32) instead of v_mul_hi_u32 + v_mul_lo_u32:
        // integer multiplies issue at a quarter of the VALU rate and are the bulk of this routine
        const uint64_t p0 = static_cast<uint64_t>(c0) * 0xD2511F53ull;
        const uint64_t p1 = static_cast<uint64_t>(c2) * 0xCD9E8D57ull;
        const uint32_t hi0 = static_cast<uint32_t>(p0 >> 32), lo0 = static_cast<uint32_t>(p0);
        const uint32_t hi1 = static_cast<uint32_t>(p1 >> 32), lo1 = static_cast<uint32_t>(p1);
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

// The same ten rounds computed ONCE per pair of adjacent lanes (2i, 2i + 1) that share a counter: the even
// lane carries (c0, c1) and forms the M0 product, the odd lane carries (c2, c3) and forms the M1 product; a
// round swaps the two products between the lanes (DPP quad_perm [1, 0, 3, 2], no LDS).  One quarter-rate
// 32x32->64 multiply per lane and round instead of two.  Returns the lane's half of the block: words (0, 1)
// on the even lane, (2, 3) on the odd lane -- bit for bit philox4x32_10's.  All 64 lanes must be active.
__device__ __forceinline__ uint32_t swap_pair(uint32_t v)
{
    return static_cast<uint32_t>(__builtin_amdgcn_update_dpp(0, static_cast<int>(v), 0xB1, 0xF, 0xF, true));
}

__device__ __forceinline__ void philox4x32_10_pair(uint32_t k0, uint32_t k1, uint32_t c0, uint32_t c1, uint32_t c2,
                                                   uint32_t c3, bool odd, uint32_t& w0, uint32_t& w1)
{
    // m: this lane's multiplicand (c0 | c2).  p: the PARTNER's other word (even lane: c3, odd lane: c1) -- it is only
    // ever combined with this lane's next high word, so it stays here and one value crosses per round, not two.
    uint32_t m = odd ? c2 : c0, p = odd ? c1 : c3, key = odd ? k1 : k0;
    const uint32_t mult = odd ? 0xCD9E8D57u : 0xD2511F53u;
    const uint32_t bump = odd ? 0xBB67AE85u : 0x9E3779B9u;
#pragma unroll
    for (int r = 0; r < PHILOX_ROUNDS; ++r) {
        const uint64_t prod = static_cast<uint64_t>(m) * mult;
        const uint32_t t = static_cast<uint32_t>(prod >> 32) ^ p;   // odd: hi(M1 c2) ^ c1     even: hi(M0 c0) ^ c3
        m = swap_pair(t) ^ key;                                     // even: .. ^ k0 = c0'     odd: .. ^ k1 = c2'
        p = static_cast<uint32_t>(prod);                            // odd: lo(M1 c2) = c1'    even: lo(M0 c0) = c3'
        key += bump;
    }
    w0 = m; w1 = swap_pair(p);
}

__device__ __forceinline__ void draw_pair(uint64_t seed, uint64_t env, uint64_t tick, uint32_t stream, uint32_t attempt,
                                          bool odd, uint32_t& w0, uint32_t& w1)
{
    const uint32_t c3 = (static_cast<uint32_t>(tick >> 32) & 0xFFFFu) | ((attempt & 0xFFu) << 16) | (stream << 24);
    philox4x32_10_pair(static_cast<uint32_t>(seed), static_cast<uint32_t>(seed >> 32), static_cast<uint32_t>(env),
                       static_cast<uint32_t>(env >> 32), static_cast<uint32_t>(tick), c3, odd, w0, w1);
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
// sin and cos on the transcendental unit: v_sin_f32 / v_cos_f32 take their argument in revolutions.
// Measured on MI355X over [-3.3, 3.3] rad (tools/micro/vsin_acc.hip): max abs error 3.5e-7 / 2.5e-7, i.e.
// < 2e-7 world units on the chord -- well inside the 1e-5 budget whose bulk is the float32 rounding of x', y'
// themselves (3.8e-6).  v_fract first keeps any theta a caller may have written inside the unit's domain.
__device__ __forceinline__ void sincos_bounded(float a, float& s, float& c)
{
    const float rev = __builtin_amdgcn_fractf(a * 0.15915494309189535f);
    s = __builtin_amdgcn_sinf(rev);
    c = __builtin_amdgcn_cosf(rev);
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

// The hand-coded bearing policy of main/testing/test_optimal.py:8-28: turn towards the goal while the
// bearing error exceeds 8 degrees, else full throttle.  Like the reference it does NOT wrap the difference of
// the two [0, 2 pi) angles.  float32; worlds within ~1e-6 rad of the threshold may pick the other action
// than a float64 evaluation would.
__device__ __forceinline__ int bearing_action(float x, float y, float th, float gx, float gy)
{
    constexpr float TWO_PI = 6.28318530717958647692f, HALF_PI = 1.57079632679489661923f;
    constexpr float THRESHOLD = 0.13962634015954636f;          // 8 / 180 * pi
    float boat = (th + HALF_PI) + TWO_PI;                       // in [pi/2 + pi, 7 pi / 2): one conditional subtraction
    boat = boat >= TWO_PI ? boat - TWO_PI : boat;               // == (theta + pi/2 + 2 pi) % (2 pi)
    float goal = atan2f(gy - y, gx - x);
    goal = goal < 0.0f ? goal + TWO_PI : goal;                  // == (atan2 + 2 pi) % (2 pi)
    const float diff = goal - boat;
    return fabsf(diff) > THRESHOLD ? (diff > 0.0f ? 0 : 1) : 2;
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

// Arguments of an out-of-line device function arrive in VGPRs, so the compiler no longer knows that the
// table pointer, K, the seed ... are wave-uniform and would read the obstacle rows with per-lane vector
// loads.  These put them back into SGPRs (v_readfirstlane), which restores scalar loads and SGPR operands.
__device__ __forceinline__ uint32_t uni(uint32_t v) { return __builtin_amdgcn_readfirstlane(v); }
__device__ __forceinline__ int uni(int v) { return __builtin_amdgcn_readfirstlane(v); }
__device__ __forceinline__ float uni(float v) { return __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, v))); }
__device__ __forceinline__ uint64_t uni(uint64_t v)
{
    return (static_cast<uint64_t>(uni(static_cast<uint32_t>(v >> 32))) << 32) | uni(static_cast<uint32_t>(v));
}
// "does any active lane ...": the ballot is the compare's own SGPR pair and the test a scalar compare.  (__any()
// goes through a 0/1 value per lane: v_cndmask + v_cmp on top of the compare.)
__device__ __forceinline__ bool any_lane(bool pred) { return __builtin_amdgcn_ballot_w64(pred) != 0; }

template <typename P> __device__ __forceinline__ P uni_ptr(P p) { return (P)uni((uint64_t)(uintptr_t)p); }

// ------------------------------------------------------------------------------------ exact path
// float64, operation order of the reference (aqua.py:159-211).  Contraction is off so that products and
// sums round where the reference's do.  The path is rare (~2e-5 of world-steps) but whatever it costs is
// added to the launch (some wavefront in a 262 144-world batch takes it in practically every step), so
// it is written for latency: inputs come from registers, sin/cos are a compact bounded-argument
// routine instead of the library's generic one, the discrete-action quantities are literals, and only
// the obstacles whose float32 margin is itself inside the band are re-evaluated in float64.
struct ExactOut { float x, y, th, reward; uint32_t term; };

// sin and cos of |x| <~ 8 in float64, < 1 ulp: three-stage Cody-Waite reduction by pi/2 (the constants and
// the two kernels are fdlibm's __ieee754_rem_pio2 medium path, __kernel_sin and __kernel_cos).  It returns
// the same bits as libm for the arguments the reference's special cases hit (sin(fl(pi/2)) == 1,
// cos(fl(pi/2)) == 6.123233995736766e-17, sin(4e-9) == 4e-9, cos(4e-9) == 1).
__device__ __forceinline__ void sincos_f64(double x, double& s, double& c)
{
#pragma clang fp contract(off)
    constexpr double INVPIO2 = 6.36619772367581382433e-01;
    constexpr double P1 = 0x1.921fb54400000p+0, P1T = 0x1.0b4611a626331p-34;
    constexpr double P2 = 0x1.0b4611a600000p-34, P2T = 0x1.3198a2e037073p-69;
    constexpr double P3 = 0x1.3198a2e000000p-69, P3T = 0x1.b839a252049c1p-104;
    const double fn = rint(x * INVPIO2);
    double r = x - fn * P1, w = fn * P1T, t;
    t = r; w = fn * P2; r = t - w; w = fn * P2T - ((t - r) - w);
    t = r; w = fn * P3; r = t - w; w = fn * P3T - ((t - r) - w);
    const double y0 = r - w, y1 = (r - y0) - w;
    const double z = y0 * y0;
    // __kernel_sin(y0, y1)
    const double v = z * y0;
    const double rs = 8.33333333332248946124e-03 + z * (-1.98412698298579493134e-04 + z * (2.75573137070700676789e-06 +
                      z * (-2.50507602534068634195e-08 + z * 1.58969099521155010221e-10)));
    const double ks = y0 - ((z * (0.5 * y1 - v * rs) - y1) - v * -1.66666666666666324348e-01);
    // __kernel_cos(y0, y1)
    const double rc = z * (4.16666666666666019037e-02 + z * (-1.38888888888741095749e-03 + z * (2.48015872894767294178e-05 +
                      z * (-2.75573143513906633035e-07 + z * (2.08757232129817482790e-09 + z * -1.13596475577881948265e-11)))));
    const double hz = 0.5 * z, wc = 1.0 - hz;
    const double kc = wc + (((1.0 - wc) - hz) + (z * rc - y0 * y1));
    const int q = static_cast<int>(fn) & 3;
    const double s0 = (q & 1) ? kc : ks, c0 = (q & 1) ? ks : kc;
    s = (q & 2) ? -s0 : s0;
    c = ((q + 1) & 2) ? -c0 : c0;
}

// turn radius, angular step and its cosine/sine (aqua.py:159-170,176).  Discrete actions (aqua.py:33-42):
// the reference's own float64 values, computed once with its expressions.
struct ExactMotion { double r, w, cw, sw; };

__device__ __forceinline__ ExactMotion exact_motion_discrete(int idx)
{
    ExactMotion m;
    const bool line = idx == 2;
    const double sign = idx == 1 ? -1.0 : 1.0;
    m.r = line ? 0x1.dcd65p+26 : sign * 0x1.7555555555556p+1;                  // 1.25 (vR+vL) / d
    m.w = line ? 0x1.12e0be826d695p-28 : sign * 0x1.eb851eb851eb8p-4;          // d / 2.5
    m.cw = line ? 1.0 : 0x1.fc5169dc5b825p-1;
    m.sw = line ? 0x1.12e0be826d695p-28 : sign * 0x1.ea5758f3ce5cdp-4;
    return m;
}

__device__ __forceinline__ ExactMotion exact_motion_continuous(double vl, double vr)
{
#pragma clang fp contract(off)
    ExactMotion m;
    double diff = vr - vl;
    diff = copysign(fmax(fabs(diff), 1e-8), diff);
    m.r = 2.5 / 2 * (vr + vl) / diff;
    m.w = diff / 2.5;
    sincos_f64(m.w, m.sw, m.cw);
    return m;
}

// Per-world obstacle tables (domain randomisation: every world of the batch has its own list, as every env object
// of the reference has, aqua.py:56-68).  Struct of arrays over the worlds so that a wavefront's reads coalesce:
//   t32  float32 [K][6][ld]   row j of world i at t32[(6 j + c) ld + i],  c = cx, cy, hx, hy, r2, w  (ObstF's fields;
//                             an absent row has r2 = -3e38 and never wins a minimum)
//   t64  float64 [K][5][ld]   the reference's own row (cx, cy, kind, a, b; kind < 0: absent), for the float64 path
// A lane names its world as a wave-uniform tile base (t32 / t64 already advanced to the tile's first world) plus a
// 32-bit offset inside the tile: every access is then `global_load v, voffset, s[base]` -- the row and column arithmetic
// runs on the scalar unit and no lane holds a 64-bit address (with per-lane addresses the per-world step kernel needed
// 187 registers: two wavefronts per SIMD).
struct WorldTable {
    const float* t32;           // + tile
    const double* t64;          // + tile (NULL where the float64 path is not used)
    int64_t ld;
    uint32_t off;               // this lane's world inside the tile
};
template <typename T>
__device__ __forceinline__ T world_ld(const T* base, uint32_t elem)
{
    return *reinterpret_cast<const T*>(reinterpret_cast<const char*>(base) + elem * static_cast<uint32_t>(sizeof(T)));
}
__device__ __forceinline__ ObstF world_row(const WorldTable& t, int j)
{
    ObstF r;
    const float* p = t.t32 + (6 * j) * t.ld;
    r.cx = world_ld(p, t.off); r.cy = world_ld(p + t.ld, t.off); r.hx = world_ld(p + 2 * t.ld, t.off);
    r.hy = world_ld(p + 3 * t.ld, t.off); r.r2 = world_ld(p + 4 * t.ld, t.off); r.w = world_ld(p + 5 * t.ld, t.off);
    r.pad[0] = r.pad[1] = 0.0f;
    return r;
}

template <bool PER_WORLD>
__device__ __noinline__ ExactOut exact_step_impl(float fx, float fy, float fth, float fgx, float fgy, float fwx,
                                                 float fwy, int t_new, ExactMotion mo, int K,
                                                 const double* __restrict__ obst64, ObstPtr obst32, float band2,
                                                 int time_limit, WorldTable wt)
{
#pragma clang fp contract(off)
    constexpr double PI_D = 3.141592653589793;
    K = uni(K); band2 = uni(band2); time_limit = uni(time_limit);
    if constexpr (!PER_WORLD) {
        obst32 = uni_ptr(obst32);
        obst64 = (const double*)uni((uint64_t)(uintptr_t)obst64);
    }
    const double px = fx, py = fy, th = fth, gx = fgx, gy = fgy;
    const double r = mo.r, w = mo.w, c = mo.cw, s = mo.sw;
    const double angle = PI_D / 2 + th;
    double sa, ca;
    sincos_f64(angle, sa, ca);
    const double icc_x = px + r * (-sa);
    const double icc_y = py + r * ca;
    const double qx = px - icc_x, qy = py - icc_y;
    const double nx = (c * qx + (-s) * qy) + icc_x + static_cast<double>(fwx);
    const double ny = (s * qx + c * qy) + icc_y + static_cast<double>(fwy);
    // aqua.py:128-133: off - floor(off / width) * width - pi.  floor(off / width) is 0, 1 or -1 here; the
    // comparison form gives the same integer without a float64 division (|off| < 2 * width).
    const double width = PI_D - (-PI_D);
    const double off = (th + w) - (-PI_D);
    const double turns = off >= width ? 1.0 : (off < 0.0 ? -1.0 : 0.0);
    const double nth = (off - (turns * width)) + (-PI_D);

    bool hit = (nx - 2.5 < 0.0) || (ny - 2.5 < 0.0) || (nx + 2.5 > 100.0) || (ny + 2.5 > 100.0);
    const float xs = static_cast<float>(nx), ys = static_cast<float>(ny);
    constexpr int XL = PER_WORLD ? 4 : 1;                  // per-world rows come from memory: four per round trip (as the second look)
    ObstF chunk[XL];
    for (int k = 0; k < K; ++k) {
        ObstF row;
        if constexpr (PER_WORLD) {
            if (k % XL == 0) {
#pragma unroll
                for (int u = 0; u < XL; ++u) chunk[u] = world_row(wt, k + u < K ? k + u : K - 1);
            }
            row = chunk[0];
#pragma unroll
            for (int u = 1; u < XL; ++u) if (k % XL == u) row = chunk[u];
        }
        else { row.cx = obst32[k].cx; row.cy = obst32[k].cy; row.hx = obst32[k].hx; row.hy = obst32[k].hy; row.r2 = obst32[k].r2; }
        const float bx = fmaxf(fabsf(xs - row.cx) - row.hx, 0.0f);
        const float by = fmaxf(fabsf(ys - row.cy) - row.hy, 0.0f);
        const float m32 = fmaf(bx, bx, fmaf(by, by, -row.r2));
        if (fabsf(m32) >= band2) { hit = hit || (m32 < 0.0f); continue; }
        double o[5];
        if constexpr (PER_WORLD) {
#pragma unroll
            for (int c5 = 0; c5 < 5; ++c5) o[c5] = world_ld(wt.t64 + (5 * k + c5) * wt.ld, wt.off);
        } else {
#pragma unroll
            for (int c5 = 0; c5 < 5; ++c5) o[c5] = obst64[5 * k + c5];
        }
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
    ExactOut out;
    if (hit) { out.term = 1; out.reward = -10.0f; }
    else if (t_new > time_limit) { out.term = 2; out.reward = -10.0f; }
    else if (d_cur <= 0.0) { out.term = 3; out.reward = 10.0f; }
    else {
        const double ox = gx - px, oy = gy - py;
        const double d_prev = sqrt(ox * ox + oy * oy) - (2.5 + 2.5);
        out.term = 0; out.reward = static_cast<float>((d_prev - d_cur) * 0.7);
    }
    out.x = static_cast<float>(nx);
    out.y = static_cast<float>(ny);
    out.th = static_cast<float>(nth);
    return out;
}

__device__ __forceinline__ ExactOut exact_step(float fx, float fy, float fth, float fgx, float fgy, float fwx, float fwy,
                                               int t_new, ExactMotion mo, int K, const double* __restrict__ obst64,
                                               ObstPtr obst32, float band2, int time_limit)
{
    return exact_step_impl<false>(fx, fy, fth, fgx, fgy, fwx, fwy, t_new, mo, K, obst64, obst32, band2, time_limit, WorldTable{});
}
__device__ __forceinline__ ExactOut exact_step_world(float fx, float fy, float fth, float fgx, float fgy, float fwx,
                                                     float fwy, int t_new, ExactMotion mo, int K, float band2,
                                                     int time_limit, const WorldTable& wt)
{
    return exact_step_impl<true>(fx, fy, fth, fgx, fgy, fwx, fwy, t_new, mo, K, nullptr, nullptr, band2, time_limit, wt);
}

// ------------------------------------------------------------------------------------ fast path
// Advances one world in float32.  Returns true when a margin falls inside the knife-edge band
// (the caller then overrides pose/reward/term with exact_step()).
//
// Collision is decided from two signed margins: the border margin min(x, y) - 2.5, 97.5 - max(x, y)
// (aqua.py:424-427) and the minimum over obstacles of d^2 - R^2 with d the distance from the boat centre
// to the obstacle's box (aqua.py:373-390, 429-439): 5 VALU operations per circle, 9 per rectangle, no
// compares.  d^2 - R^2 has exactly the sign of d - R, so outside the bands the sign of the minimum IS the
// reference's OR-of-tests; |d^2 - R^2| < band2 = 2.5 R_max BAND covers |d - R| < BAND for every R <= R_max,
// and the strict/non-strict difference of the reference's comparisons only matters at margin == 0,
// which is inside the band by construction.
// PER_WORLD: the obstacle rows come from this lane's own table (`wt`, generic box formula for every row) instead of
// the batch's shared one.
// Plain one-obstacle operations on purpose.  The packed forms (v_pk_add_f32 / v_pk_fma_f32 on SGPR pairs, two obstacles
// per operation) and d - v_med3_f32(d, -h, h) for the rectangles' fabs / subtract / max triple are 8 and 12 vector
// instructions fewer per group and measured 0.07 us per step SLOWER (5.16 vs 5.09).
__device__ __forceinline__ float quick_min(float mo, float xn, float yn, const QuickCircles& g)
{
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const float dx = xn - g.cx[j], dy = yn - g.cy[j];
        mo = fminf(mo, fmaf(dx, dx, fmaf(dy, dy, g.nr2[j])));
    }
    return mo;
}
__device__ __forceinline__ float quick_min(float mo, float xn, float yn, const QuickRects& g)
{
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const float dx = fmaxf(fabsf(xn - g.cx[j]) - g.hx[j], 0.0f);
        const float dy = fmaxf(fabsf(yn - g.cy[j]) - g.hy[j], 0.0f);
        mo = fminf(mo, fmaf(dx, dx, fmaf(dy, dy, g.nr2[j])));
    }
    return mo;
}

// QUICK: the first look reads the quick table (k.qc0, k.qr0, k.quick) instead of walking the rows --
// QUICK_ALWAYS: the caller knows the table has one; QUICK_IF_PRESENT: when k.quick is not NULL (wave-uniform).
enum : int { QUICK_NEVER = 0, QUICK_ALWAYS = 1, QUICK_IF_PRESENT = 2 };
// INFLIGHT (PER_WORLD with KREG == 0): rows of its table a lane has in flight while it streams them.  The loop is bound by
// memory latency x loads in flight per SIMD: the no-restart kernel runs seven wavefronts per SIMD on 71 registers and reaches
// the copy bandwidth with two rows in flight; the restart kernels carry the re-seeding code's 91-95 registers (five
// wavefronts per SIMD) and ran at exactly 5/7 of its rate -- "the restart's cost" of rounds 3 and 4 (64 rows: 98 against 70
// us per step) was this, not the re-seeding.  They have the registers for more rows in flight at no cost in occupancy.
#ifndef AQUA_SECOND_LOOK_ROWS
#define AQUA_SECOND_LOOK_ROWS 8
#endif
#ifndef AQUA_SECOND_LOOK_ROWS_KREG
#define AQUA_SECOND_LOOK_ROWS_KREG 4
#endif
constexpr int SECOND_LOOK_ROWS = AQUA_SECOND_LOOK_ROWS, SECOND_LOOK_ROWS_KREG = AQUA_SECOND_LOOK_ROWS_KREG;   // (fast_step's second look)
#ifndef AQUA_TABLE_ROWS_IN_FLIGHT_RESTART
#define AQUA_TABLE_ROWS_IN_FLIGHT_RESTART 4
#endif
// KREG > 0 (PER_WORLD only): the caller has read the lane's first KREG rows into registers (`regs`, loaded with the
// state, one memory round trip for everything; rows past the table's end repeat its last row, which leaves a minimum
// unchanged) and the first look is arithmetic only; the second look and the float64 path (rare) still read `wt`.
// SINK (PER_WORLD with KREG == 0 only): a lane whose `sink` is not NULL leaves every row it reads, as it reads it, at
// sink[5 j .. 5 j + 4] (cx, cy, hx, hy, r2; an LDS slot of the caller's) -- the rows of a world that restarts this tick,
// for the lanes that re-seed it (reset_env_group<.., RESEED_LDS5>): they ride along with the wavefront's own coalesced row
// loads instead of being fetched again, one 128-byte line per float, by the re-seeding group.
template <bool PER_WORLD = false, int QUICK = QUICK_NEVER, int KREG = 0, bool SINK = false, int INFLIGHT = 2>
__device__ __forceinline__ bool fast_step(EnvState& e, float h, float w, float chord, float u0, float u1,
                                          const StepConst& k, float& reward, uint32_t& term, const WorldTable* wt = nullptr,
                                          const ObstF* regs = nullptr, float* sink = nullptr)
{
    static_assert(!SINK || (PER_WORLD && KREG == 0), "rows are left behind by the loop that streams them");
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

    const float mc = fminf(fminf(xn, yn) - 2.5f, 97.5f - fmaxf(xn, yn));
    float mo = 3.0e38f;                                    // min over obstacles of d^2 - R^2
    if constexpr (PER_WORLD && KREG > 0) {
#pragma unroll
        for (int j = 0; j < KREG; ++j) {                   // (same operations per row as below: same bits)
            const float dx = fmaxf(fabsf(xn - regs[j].cx) - regs[j].hx, 0.0f);
            const float dy = fmaxf(fabsf(yn - regs[j].cy) - regs[j].hy, 0.0f);
            mo = fminf(mo, fmaf(dx, dx, fmaf(dy, dy, -regs[j].r2)));
        }
    } else if constexpr (PER_WORLD) {
#pragma unroll INFLIGHT
        for (int j = 0; j < k.K; ++j) {                    // a circle is a box with zero half extents: same bits
            const ObstF r = world_row(*wt, j);
            if constexpr (SINK) {
                if (sink != nullptr) {
                    float* const d = sink + 5 * j;
                    d[0] = r.cx; d[1] = r.cy; d[2] = r.hx; d[3] = r.hy; d[4] = r.r2;
                }
            }
            const float dx = fmaxf(fabsf(xn - r.cx) - r.hx, 0.0f);
            const float dy = fmaxf(fabsf(yn - r.cy) - r.hy, 0.0f);
            mo = fminf(mo, fmaf(dx, dx, fmaf(dy, dy, -r.r2)));
        }
    } else if (QUICK == QUICK_ALWAYS || (QUICK == QUICK_IF_PRESENT && k.quick != nullptr)) {
        const int nr = k.K - k.Kc;
        if (k.Kc > 0) {
            mo = quick_min(mo, xn, yn, k.qc0);
            if (k.Kc > 4) mo = quick_min(mo, xn, yn, quick_circles(k.quick, QUICK_C1));
        }
        if (nr > 0) {
            mo = quick_min(mo, xn, yn, k.qr0);
            if (nr > 4) mo = quick_min(mo, xn, yn, quick_rects(k.quick, QUICK_R1));
        }
    } else {
#pragma unroll 2
    for (int j = 0; j < k.Kc; ++j) {                       // circles: distance to the centre
        const float dx = xn - k.obst[j].cx, dy = yn - k.obst[j].cy;
        mo = fminf(mo, fmaf(dx, dx, fmaf(dy, dy, -k.obst[j].r2)));
    }
#pragma unroll 2
    for (int j = k.Kc; j < k.K; ++j) {                     // rectangles: distance to the box
        const float dx = fmaxf(fabsf(xn - k.obst[j].cx) - k.obst[j].hx, 0.0f);
        const float dy = fmaxf(fabsf(yn - k.obst[j].cy) - k.obst[j].hy, 0.0f);
        mo = fminf(mo, fmaf(dx, dx, fmaf(dy, dy, -k.obst[j].r2)));
    }
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

    bool knife = (fminf(fabsf(mc), fabsf(mg)) < BAND) || (fabsf(mo) < k.band2);
#ifdef AQUA_NO_SECOND_LOOK                   // (timing experiment: neither the second look nor the float64 path; results differ)
    knife = false;
#endif
    float mc_f = mc, mo_f = mo, mg_f = mg;
    if (__builtin_expect(any_lane(knife), 0)) {
        // Second look, still float32, for the worlds inside the band: the float32 margins above are limited
        // by the rounding of x' = x + ddx itself (up to 3.8e-6).  Carry the rounding error of that sum
        // (xlo = ddx - (x' - x), exact: FastTwoSum) through the three margins; what is left is the error of
        // ddx (~3e-7) and of d^2 (~4 ulp of R^2), so the band shrinks from 1e-4 to 2e-6 (goal: 4e-6) and the float64
        // path -- whose length is added to the launch whenever ANY wavefront takes it -- is entered
        // ~25 times less often.
        const float xlo = ddx - (xn - e.x), ylo = ddy - (yn - e.y);
        const float mb2 = fminf(fminf((xn - 2.5f) + xlo, (yn - 2.5f) + ylo), fminf((97.5f - xn) - xlo, (97.5f - yn) - ylo));
        float mo2 = 3.0e38f;
        // Each margin is scaled by its obstacle's w >= 1 (sign unchanged): |d^2 - R^2| < band2_tight(R_max) / w is the
        // band of THAT obstacle's radius, 2.5 (R + BAND) BAND_TIGHT + 4 ulp(R^2) -- not R_max's, which for the
        // R = 2.5 of every rectangle would be several times wider than its own in distance.
        if constexpr (PER_WORLD) {
            // A cold path -- a wavefront in two hundred takes it -- but one that ENDS the launch: every wavefront of a
            // streaming launch reaches this point at about the same time, so the launch lasts as long as the slowest second
            // look.  One row at a time (rounds 1-4: the loop was kept out of the kernel's register count) that was K
            // dependent memory round trips at the tail of EVERY launch -- 13 of the 70 us of a 64-row step, 1.5 of the 10 us
            // of an 8-row one.  Now SECOND_LOOK_ROWS rows per round trip; a row index past the table repeats its last row,
            // which leaves a minimum unchanged.  (The registers are there: capping these kernels at five or four wavefronts
            // per SIMD changes nothing, profiles/r05/tables/.)
            constexpr int SL = KREG == 0 ? SECOND_LOOK_ROWS : SECOND_LOOK_ROWS_KREG;
#pragma unroll 1
            for (int j0 = 0; j0 < k.K; j0 += SL) {
                ObstF r[SL];
#pragma unroll
                for (int u = 0; u < SL; ++u) r[u] = world_row(*wt, j0 + u < k.K ? j0 + u : k.K - 1);
#pragma unroll
                for (int u = 0; u < SL; ++u) {
                    const float dx = fmaxf(fabsf((xn - r[u].cx) + xlo) - r[u].hx, 0.0f);
                    const float dy = fmaxf(fabsf((yn - r[u].cy) + ylo) - r[u].hy, 0.0f);
                    mo2 = fminf(mo2, fmaf(dx, dx, fmaf(dy, dy, -r[u].r2)) * r[u].w);
                }
            }
        } else {
        for (int j = 0; j < k.Kc; ++j) {
            const float dx = (xn - k.obst[j].cx) + xlo, dy = (yn - k.obst[j].cy) + ylo;
            mo2 = fminf(mo2, fmaf(dx, dx, fmaf(dy, dy, -k.obst[j].r2)) * k.obst[j].w);
        }
        for (int j = k.Kc; j < k.K; ++j) {
            const float dx = fmaxf(fabsf((xn - k.obst[j].cx) + xlo) - k.obst[j].hx, 0.0f);
            const float dy = fmaxf(fabsf((yn - k.obst[j].cy) + ylo) - k.obst[j].hy, 0.0f);
            mo2 = fminf(mo2, fmaf(dx, dx, fmaf(dy, dy, -k.obst[j].r2)) * k.obst[j].w);
        }
        }
        const float gx2 = (e.gx - xn) - xlo, gy2 = (e.gy - yn) - ylo;
        const float mg2 = __builtin_amdgcn_sqrtf(fmaf(gx2, gx2, gy2 * gy2)) - 5.0f;
        if (knife) { mc_f = mb2; mo_f = mo2; mg_f = mg2; }
        knife = knife && ((fabsf(mb2) < BAND_TIGHT) || (fabsf(mg2) < BAND_TIGHT_GOAL) || (fabsf(mo2) < k.band2_tight));
    }
    term = (fminf(mc_f, mo_f) < 0.0f) ? 1u : (tn > k.time_limit ? 2u : (mg_f <= 0.0f ? 3u : 0u));   // aqua.py:200-211
    reward = term == 0u ? shaped : (term == 3u ? 10.0f : -10.0f);
    e.x = xn; e.y = yn; e.th = thn; e.wx = wxn; e.wy = wyn; e.t = tn;
#ifdef AQUA_NO_EXACT                         // (timing experiment: the second look but never the float64 path; results differ)
    return false;
#endif
    return knife;
}

// ------------------------------------------------------------------------------------ reset
// float32 specification shared with nothing: oracle/aqua_oracle.c restates it independently and the
// two are compared bit for bit.  Every rounding is explicit (fmaf or one operation per statement).
// COLD: a path taken once in thousands of launches (the serial scan of reset_env_group): one row at a time.  Unrolled
// and vectorised, that loop's preheader alone spilled fourteen SGPRs of the next-step kernel to VGPR lanes.
template <bool COLD = false>
__device__ __forceinline__ bool reset_hit(int K, ObstPtr t, float px, float py)
{
#pragma clang fp contract(off)
    bool hit = false;
    const auto test = [&](int j) {
        const float ax = fabsf(px - t[j].cx), ay = fabsf(py - t[j].cy);
        const float dx = fmaxf(ax - t[j].hx, 0.0f), dy = fmaxf(ay - t[j].hy, 0.0f);
        const float dy2 = dy * dy;
        const float d2 = fmaf(dx, dx, dy2);
        hit |= d2 <= t[j].r2;
    };
    if constexpr (COLD) {
#pragma clang loop unroll(disable) vectorize(disable)
        for (int j = 0; j < K; ++j) test(j);
    } else {
#pragma unroll 4
        for (int j = 0; j < K; ++j) test(j);
    }
    return hit;
}

__device__ __noinline__ EnvState reset_env(uint64_t seed, uint64_t env, uint64_t tick, int waves, int random_boat,
                                           int random_goal, int K, ObstPtr t)
{
#pragma clang fp contract(off)
    EnvState e;
    constexpr float PI_F = 3.14159274101257324f, TWO_PI_F = 6.28318548202514648f;
    K = uni(K); t = uni_ptr(t); seed = uni(seed); tick = uni(tick);
    waves = uni(waves); random_boat = uni(random_boat); random_goal = uni(random_goal);
    uint32_t r[4];
    float gx = 25.0f, gy = 80.0f;                          // aqua.py:107
    if (random_goal) {
        for (uint32_t a = 0; a < RESET_TRIES; ++a) {       // aqua.py:103-105
            draw(seed, env, tick, STREAM_PLACE, a, r);
            const float cx = fmaf(95.0f, u_01(r[0]), 2.5f);
            const float cy = fmaf(95.0f, u_01(r[1]), 2.5f);
            if (!reset_hit(K, t, cx, cy)) { gx = cx; gy = cy; break; }
        }
    }
    draw(seed, env, tick, STREAM_POSE, 0, r);              // heading (aqua.py:111) and wave (aqua.py:124)
    const float W = 0.05f * static_cast<float>(waves);
    const float heading = fmaf(TWO_PI_F, u_01(r[0]), -PI_F);
    e.wx = W * u_pm1(r[1]);
    e.wy = W * u_pm1(r[2]);
    float bx = 85.0f, by = 45.0f, bt = 0.0f;               // aqua.py:117
    if (random_boat) {
        for (uint32_t a = 0; a < RESET_TRIES; ++a) {       // aqua.py:111-115
            draw(seed, env, tick, STREAM_PLACE, a, r);
            const float cx = fmaf(95.0f, u_01(r[2]), 2.5f);
            const float cy = fmaf(95.0f, u_01(r[3]), 2.5f);
            const float ex = gx - cx, ey = gy - cy;
            const float ey2 = ey * ey;
            const float g2 = fmaf(ex, ex, ey2);
            if (g2 <= 25.0f) continue;
            if (reset_hit(K, t, cx, cy)) continue;
            bx = cx; by = cy; bt = heading;
            break;
        }
    }
    e.x = bx; e.y = by; e.th = bt; e.gx = gx; e.gy = gy;
    e.t = 0;                                               // aqua.py:125
    return e;
}

// The same specification against this lane's own table (per-world tables): the serial loop, one world per lane.
// Tables of up to WORLD_ROWS_IN_REGS rows are read ONCE into registers (all loads in flight together, one memory
// round trip) and every hit test after that is arithmetic only; longer tables are re-read per test.
constexpr int WORLD_ROWS_IN_REGS = 8;
struct WorldRows {
    ObstF r[WORLD_ROWS_IN_REGS];
    bool cached;
};
__device__ __forceinline__ bool reset_hit_world(int K, const WorldTable& t, const WorldRows& rows, float px, float py)
{
#pragma clang fp contract(off)
    bool hit = false;
    auto test = [&](const ObstF& r) {
        const float ax = fabsf(px - r.cx), ay = fabsf(py - r.cy);
        const float dx = fmaxf(ax - r.hx, 0.0f), dy = fmaxf(ay - r.hy, 0.0f);
        const float dy2 = dy * dy;
        const float d2 = fmaf(dx, dx, dy2);
        hit |= d2 <= r.r2;
    };
    if (rows.cached) {
#pragma unroll
        for (int j = 0; j < WORLD_ROWS_IN_REGS; ++j) test(rows.r[j]);       // absent rows: r2 = -3e38, never hit
    } else {
        for (int j = 0; j < K; ++j) test(world_row(t, j));
    }
    return hit;
}

__device__ __forceinline__ WorldRows load_world_rows(int K, const WorldTable& t)
{
    WorldRows rows;
    rows.cached = K <= WORLD_ROWS_IN_REGS;                 // uniform
#pragma unroll
    for (int j = 0; j < WORLD_ROWS_IN_REGS; ++j) {
        if (rows.cached && j < K) rows.r[j] = world_row(t, j);
        else { rows.r[j].cx = rows.r[j].cy = rows.r[j].hx = rows.r[j].hy = 0.0f; rows.r[j].r2 = -3.0e38f; rows.r[j].w = 1.0f; }
    }
    return rows;
}

__device__ __forceinline__ EnvState reset_env_world(uint64_t seed, uint64_t env, uint64_t tick, int waves, int random_boat,
                                                    int random_goal, int K, const WorldTable& t)
{
#pragma clang fp contract(off)
    const WorldRows rows = load_world_rows(K, t);
    EnvState e;
    constexpr float PI_F = 3.14159274101257324f, TWO_PI_F = 6.28318548202514648f;
    uint32_t r[4];
    float gx = 25.0f, gy = 80.0f;                          // aqua.py:107
    if (random_goal) {
        for (uint32_t a = 0; a < RESET_TRIES; ++a) {       // aqua.py:103-105
            draw(seed, env, tick, STREAM_PLACE, a, r);
            const float cx = fmaf(95.0f, u_01(r[0]), 2.5f);
            const float cy = fmaf(95.0f, u_01(r[1]), 2.5f);
            if (!reset_hit_world(K, t, rows, cx, cy)) { gx = cx; gy = cy; break; }
        }
    }
    draw(seed, env, tick, STREAM_POSE, 0, r);              // heading (aqua.py:111) and wave (aqua.py:124)
    const float W = 0.05f * static_cast<float>(waves);
    const float heading = fmaf(TWO_PI_F, u_01(r[0]), -PI_F);
    e.wx = W * u_pm1(r[1]);
    e.wy = W * u_pm1(r[2]);
    float bx = 85.0f, by = 45.0f, bt = 0.0f;               // aqua.py:117
    if (random_boat) {
        for (uint32_t a = 0; a < RESET_TRIES; ++a) {       // aqua.py:111-115
            draw(seed, env, tick, STREAM_PLACE, a, r);
            const float cx = fmaf(95.0f, u_01(r[2]), 2.5f);
            const float cy = fmaf(95.0f, u_01(r[3]), 2.5f);
            const float ex = gx - cx, ey = gy - cy;
            const float ey2 = ey * ey;
            const float g2 = fmaf(ex, ex, ey2);
            if (g2 <= 25.0f) continue;
            if (reset_hit_world(K, t, rows, cx, cy)) continue;
            bx = cx; by = cy; bt = heading;
            break;
        }
    }
    e.x = bx; e.y = by; e.th = bt; e.gx = gx; e.gy = gy;
    e.t = 0;                                               // aqua.py:125
    return e;
}

// The same specification with the attempts of ONE world spread over a group of G adjacent lanes:
// lane `sub` of the group evaluates attempts sub, sub + G, ... and the group keeps the lowest accepted
// attempt (ballot + find-first), which is exactly the attempt the serial loop above stops at.  The
// two Philox chains (placement attempt, heading + wave) of the first round are independent and interleave.  Must be
// called by all 64 lanes of a wavefront together; `active` says whether this lane's group has a world.
// Every lane of a group returns the group's result.
// (inlined: as an out-of-line function a callee-saved VGPR was spilled to scratch, 9.5 -> 9.1 us per step)
// ROWS > 0: the table (ROWS rows, absent ones with r2 < 0) is read from `rows` in LDS, a few rows at a time, instead
// of through the scalar path, which waits once per two rows.
// ROWS == RESEED_QUICK: the table is read from its quick table `quick` (Kc circles first): one scalar-memory round trip
// per group of four obstacles, and the hit tests are the row tests on the same values (a circle's half extents are
// the zeros they are in its row; r2 = -(-r2)).
// ROWS == RESEED_WORLD: the table is the group's OWN (per-world tables): `wt` names the world the group re-seeds; its
// lanes read the same addresses, so a row costs the group one memory transaction per field.  Rows are read two at
// a time inside the attempt loop (measured: all rows up front, 40 more live registers, made the masked reset
// launch 10.2 us instead of 8.8; one world per lane with the rows cached: 10 us).
// ROWS == RESEED_HANDOFF8: per-world tables of at most eight rows whose rows the world's OWN lane already holds in
// registers: it has left them in LDS (`rows` is the group's slot, a per-lane LDS pointer; rows past the table's end
// repeat its last row) -- no memory round trip at all.  The rare serial scan falls back to the table in memory (`wt`).
// ROWS == RESEED_SOA: the same, for a block that keeps the tables of ALL its worlds in LDS as [row][field][world] (the
// fused per-world rollout: SOA_ROWS = 8 / 16 / 32 / 64 rows, SOA_STRIDE = 256 / 256 / 128 / 64 worlds per block): `rows`
// points at the world's column, fields SOA_STRIDE floats apart.  SOA_SPLIT > 1: a world is re-seeded by SOA_SPLIT groups
// of G lanes side by side; every group makes the same G attempts, each against its own SOA_ROWS / SOA_SPLIT rows, and the
// hits are OR-ed across the groups before anything is decided (a pass over a long table is then SOA_SPLIT times
// shorter; everything downstream is computed redundantly, and identically, by every group).
// ROWS == RESEED_LDS5: per-world tables of ANY length whose rows the world's own lane left in LDS while it streamed them
// (fast_step<.., SINK>): `rows` points at the slot, K rows of five floats (cx, cy, hx, hy, r2) back to back; two rows in
// registers at a time, a run-time loop.  The rare serial scan reads the table in memory (`wt`).
constexpr int RESEED_QUICK = -1, RESEED_WORLD = -2, RESEED_HANDOFF8 = -4, RESEED_SOA = -5, RESEED_HANDOFF16 = -6, RESEED_LDS5 = -7;
template <int G, int ROWS = 0, int SOA_ROWS = 8, int SOA_STRIDE = 256, int SOA_SPLIT = 1>
__device__ __forceinline__ EnvState reset_env_group(bool active, uint64_t seed, uint64_t env, uint64_t tick, int waves,
                                                     int random_boat, int random_goal, int K, ObstPtr t,
                                                     const ObstF* rows = nullptr, QuickPtr quick = nullptr, int Kc = 0,
                                                     const WorldTable* wt = nullptr)
{
#pragma clang fp contract(off)
    static_assert(G >= 2 && G <= 64 && (G & (G - 1)) == 0, "group size");
    constexpr float PI_F = 3.14159274101257324f, TWO_PI_F = 6.28318548202514648f;
    K = uni(K); t = uni_ptr(t); seed = uni(seed); tick = uni(tick);
    waves = uni(waves); random_boat = uni(random_boat); random_goal = uni(random_goal);
    const int lane = static_cast<int>(threadIdx.x) & 63;
    const int sub = lane & (G - 1), gbase = lane & ~(G - 1);
    const uint64_t gmask = (G == 64) ? ~0ull : ((1ull << G) - 1ull);
    uint32_t rp[4], rw[4];
    draw(seed, env, tick, STREAM_PLACE, static_cast<uint32_t>(sub), rp);   // attempt `sub`: goal AND boat candidates
    draw(seed, env, tick, STREAM_POSE, 0, rw);
    const float W = 0.05f * static_cast<float>(waves);
    const float heading = fmaf(TWO_PI_F, u_01(rw[0]), -PI_F);

    float gx = 25.0f, gy = 80.0f, bx = 85.0f, by = 45.0f, bt = 0.0f;
    bool goal_found = !active || !random_goal;
    bool boat_done = !active || !random_boat;
    bool serial = false;                 // this group must scan the boat attempts serially from 0 (rare, see below)
    for (uint32_t base = 0; base < RESET_TRIES; base += G) {
        if (base != 0) draw(seed, env, tick, STREAM_PLACE, base + sub, rp);
        const float cgx = fmaf(95.0f, u_01(rp[0]), 2.5f), cgy = fmaf(95.0f, u_01(rp[1]), 2.5f);
        const float cbx = fmaf(95.0f, u_01(rp[2]), 2.5f), cby = fmaf(95.0f, u_01(rp[3]), 2.5f);
        // one pass over the obstacle table tests both candidates of this attempt
        bool hit_g = false, hit_b = false;
        auto test = [&](float cx, float cy, float hx, float hy, float r2) {
            const float gax = fabsf(cgx - cx), gay = fabsf(cgy - cy);
            const float gdx = fmaxf(gax - hx, 0.0f), gdy = fmaxf(gay - hy, 0.0f);
            const float gdy2 = gdy * gdy;
            hit_g |= fmaf(gdx, gdx, gdy2) <= r2;
            const float bax = fabsf(cbx - cx), bay = fabsf(cby - cy);
            const float bdx = fmaxf(bax - hx, 0.0f), bdy = fmaxf(bay - hy, 0.0f);
            const float bdy2 = bdy * bdy;
            hit_b |= fmaf(bdx, bdx, bdy2) <= r2;
        };
        if constexpr (ROWS > 0 || ROWS == RESEED_HANDOFF8 || ROWS == RESEED_HANDOFF16) {   // `rows` is in LDS: a few rows in registers at a time
            constexpr int RS = 2;                         // rows in registers at a time
            constexpr int NROWS = ROWS > 0 ? ROWS : (ROWS == RESEED_HANDOFF16 ? 16 : 8);
#pragma unroll
            for (int h = 0; h < NROWS; h += RS) {
                // (the index is laundered so that the reads stay here, next to their use, instead of being
                // hoisted out of the attempt loop into forty long-lived registers)
                int first_row = h;
                asm volatile("" : "+v"(first_row));
                const ObstF* r = rows + first_row;
                float c[RS][5];
#pragma unroll
                for (int j = 0; j < RS; ++j) {
                    c[j][0] = r[j].cx; c[j][1] = r[j].cy; c[j][2] = r[j].hx; c[j][3] = r[j].hy; c[j][4] = r[j].r2;
                }
#pragma unroll
                for (int j = 0; j < RS; ++j) test(c[j][0], c[j][1], c[j][2], c[j][3], c[j][4]);
                // finish these rows before the next ones are read: without it the vectoriser pairs operations
                // across ALL rows and keeps the whole table (and sixteen partial results) live at once
                uint32_t fg = hit_g, fb = hit_b;
                asm volatile("" : "+v"(fg), "+v"(fb));
                hit_g = fg != 0u; hit_b = fb != 0u;
            }
        } else if constexpr (ROWS == RESEED_SOA) {
            // rows in flight at a time: two (as above) for short tables; eight for the long ones, whose pass is otherwise
            // 32 LDS round trips one behind the other (the block has registers to spare: it is alone on its CU)
            static_assert(SOA_SPLIT >= 1 && G * SOA_SPLIT <= 64 && SOA_ROWS % SOA_SPLIT == 0, "whole groups, whole shares of the rows");
            constexpr int MINE = SOA_ROWS / SOA_SPLIT;     // rows this lane's group tests
            constexpr int RS = MINE >= 8 && SOA_ROWS >= 32 ? 8 : 2;
            static_assert(MINE % RS == 0, "whole batches of rows");
            const int share = SOA_SPLIT > 1 ? (lane & (G * SOA_SPLIT - 1)) / G : 0;
            const float* const soa = reinterpret_cast<const float*>(rows) + share * (MINE * 5 * SOA_STRIDE);
#pragma unroll
            for (int h = 0; h < MINE; h += RS) {
                int first_row = h;
                asm volatile("" : "+v"(first_row));      // (as above: the reads stay next to their use)
                const float* r = soa + first_row * 5 * SOA_STRIDE;
                float c[RS][5];
#pragma unroll
                for (int j = 0; j < RS; ++j)
#pragma unroll
                    for (int f = 0; f < 5; ++f) c[j][f] = r[(j * 5 + f) * SOA_STRIDE];
#pragma unroll
                for (int j = 0; j < RS; ++j) test(c[j][0], c[j][1], c[j][2], c[j][3], c[j][4]);
                uint32_t fg = hit_g, fb = hit_b;
                asm volatile("" : "+v"(fg), "+v"(fb));
                hit_g = fg != 0u; hit_b = fb != 0u;
            }
            if constexpr (SOA_SPLIT > 1) {                 // a candidate is hit if any group's share of the rows hits it
                constexpr int GW = G * SOA_SPLIT;
                uint64_t every = 0;                        // bit 0 of every group of the world
#pragma unroll
                for (int c = 0; c < SOA_SPLIT; ++c) every |= 1ull << (c * G);
                const int wbase = lane & ~(GW - 1);
                const uint64_t mg = __ballot(hit_g) >> wbase, mb = __ballot(hit_b) >> wbase;
                hit_g = ((mg >> sub) & every) != 0ull;
                hit_b = ((mb >> sub) & every) != 0ull;
            }
        } else if constexpr (ROWS == RESEED_LDS5) {
            // SOA_SPLIT > 1 (as for RESEED_SOA): the world is re-seeded by SOA_SPLIT groups of G lanes side by side, every
            // group makes the same G attempts against its own share of the rows (pairs of rows dealt round robin), and the
            // hits are OR-ed across the groups before anything is decided: the pass over a long table -- the tail of its
            // block, and with all blocks resident of the launch -- is SOA_SPLIT times shorter
            static_assert(SOA_SPLIT >= 1 && G * SOA_SPLIT <= 64, "whole groups");
            const float* const slot = reinterpret_cast<const float*>(rows);
            const int share = SOA_SPLIT > 1 ? (lane & (G * SOA_SPLIT - 1)) / G : 0;
#pragma unroll 1
            for (int j = 2 * share; j < K; j += 2 * SOA_SPLIT) {       // two rows in registers; an odd K tests its last row twice
                const float* const r0 = slot + 5 * j;
                const float* const r1 = slot + 5 * (j + 1 < K ? j + 1 : j);
                const float c0[5] = {r0[0], r0[1], r0[2], r0[3], r0[4]}, c1[5] = {r1[0], r1[1], r1[2], r1[3], r1[4]};
                test(c0[0], c0[1], c0[2], c0[3], c0[4]);
                test(c1[0], c1[1], c1[2], c1[3], c1[4]);
            }
            if constexpr (SOA_SPLIT > 1) {                 // a candidate is hit if any group's share of the rows hits it
                constexpr int GW = G * SOA_SPLIT;
                uint64_t every = 0;                        // bit 0 of every group of the world
#pragma unroll
                for (int c = 0; c < SOA_SPLIT; ++c) every |= 1ull << (c * G);
                const int wbase = lane & ~(GW - 1);
                const uint64_t mg = __ballot(hit_g) >> wbase, mb = __ballot(hit_b) >> wbase;
                hit_g = ((mg >> sub) & every) != 0ull;
                hit_b = ((mb >> sub) & every) != 0ull;
            }
        } else if constexpr (ROWS == RESEED_WORLD) {
            if constexpr (SOA_SPLIT > 1) {
                // SOA_SPLIT groups per world, four rows per round trip each (as RESEED_LDS5's split, but the rows come from
                // memory): K / (4 SOA_SPLIT) dependent round trips at the block's tail instead of K / 2
                static_assert(G * SOA_SPLIT <= 64, "whole groups");
                const int share = (lane & (G * SOA_SPLIT - 1)) / G;
#pragma unroll 1
                for (int j = 4 * share; j < K; j += 4 * SOA_SPLIT) {
                    ObstF r[4];
#pragma unroll
                    for (int u = 0; u < 4; ++u) r[u] = world_row(*wt, j + u < K ? j + u : K - 1);     // (a repeated row changes no OR)
#pragma unroll
                    for (int u = 0; u < 4; ++u) test(r[u].cx, r[u].cy, r[u].hx, r[u].hy, r[u].r2);
                }
                constexpr int GW = G * SOA_SPLIT;
                uint64_t every = 0;
#pragma unroll
                for (int c = 0; c < SOA_SPLIT; ++c) every |= 1ull << (c * G);
                const int wbase = lane & ~(GW - 1);
                const uint64_t mg = __ballot(hit_g) >> wbase, mb = __ballot(hit_b) >> wbase;
                hit_g = ((mg >> sub) & every) != 0ull;
                hit_b = ((mb >> sub) & every) != 0ull;
            } else {
#pragma unroll 1
            for (int j = 0; j < K; j += 2) {             // two rows in flight; an odd K tests its last row twice
                const ObstF r0 = world_row(*wt, j), r1 = world_row(*wt, j + 1 < K ? j + 1 : j);
                test(r0.cx, r0.cy, r0.hx, r0.hy, r0.r2);
                test(r1.cx, r1.cy, r1.hx, r1.hy, r1.r2);
            }
            }
        } else if constexpr (ROWS == RESEED_QUICK) {
            const auto circles = [&](const QuickCircles& g) {
#pragma unroll
                for (int j = 0; j < 4; ++j) test(g.cx[j], g.cy[j], 0.0f, 0.0f, -g.nr2[j]);
            };
            const auto rects = [&](const QuickRects& g) {
#pragma unroll
                for (int j = 0; j < 4; ++j) test(g.cx[j], g.cy[j], g.hx[j], g.hy[j], -g.nr2[j]);
            };
            Kc = uni(Kc); quick = uni_ptr(quick);
            if (Kc > 0) {
                circles(quick_circles(quick, QUICK_C0));
                if (Kc > 4) circles(quick_circles(quick, QUICK_C1));
            }
            if (K - Kc > 0) {
                rects(quick_rects(quick, QUICK_R0));
                if (K - Kc > 4) rects(quick_rects(quick, QUICK_R1));
            }
        } else {
#pragma unroll 2
            for (int j = 0; j < K; ++j) test(t[j].cx, t[j].cy, t[j].hx, t[j].hy, t[j].r2);
        }
        // goal: lowest accepted attempt of the group (aqua.py:103-105)
        const uint64_t gm = (__ballot(!goal_found && !hit_g) >> gbase) & gmask;
        const int gsrc = gm ? gbase + __builtin_ctzll(gm) : lane;
        const float sgx = __shfl(cgx, gsrc), sgy = __shfl(cgy, gsrc);
        if (!goal_found && gm) {
            gx = sgx; gy = sgy; goal_found = true;
            // the boat attempts are scanned from 0 with the FINAL goal (aqua.py:111-115): a goal that only
            // turns up in a later round (all G attempts of a round rejected, ~(1 - p)^G) invalidates the
            // boat candidates of the earlier rounds -> serial scan below
            if (base != 0 && !boat_done) serial = true;
        }
        const float ex = gx - cbx, ey = gy - cby;
        const float ey2 = ey * ey;
        const float g2 = fmaf(ex, ex, ey2);
        const bool boat_ok = goal_found && !boat_done && !serial && !(g2 <= 25.0f) && !hit_b;
        const uint64_t bm = (__ballot(boat_ok) >> gbase) & gmask;
        const int bsrc = bm ? gbase + __builtin_ctzll(bm) : lane;
        const float sbx = __shfl(cbx, bsrc), sby = __shfl(cby, bsrc);
        if (goal_found && !boat_done && !serial && bm) { bx = sbx; by = sby; bt = heading; boat_done = true; }
        if (!any_lane((!goal_found || !boat_done) && !serial)) break;
    }
    // goal attempts exhausted: the goal stays at its fixed default (aqua.py:107) and the boat is still scanned
    if (active && random_boat && !boat_done && !goal_found) serial = true;
    if (__builtin_expect(any_lane(serial), 0)) {
        if (serial && sub == 0) {
            uint32_t rr[4];
            for (uint32_t a = 0; a < RESET_TRIES; ++a) {
                draw(seed, env, tick, STREAM_PLACE, a, rr);
                const float cx = fmaf(95.0f, u_01(rr[2]), 2.5f), cy = fmaf(95.0f, u_01(rr[3]), 2.5f);
                const float fx = gx - cx, fy = gy - cy;
                const float fy2 = fy * fy;
                if (fmaf(fx, fx, fy2) <= 25.0f) continue;
                if constexpr (ROWS == RESEED_WORLD || ROWS == RESEED_HANDOFF8 || ROWS == RESEED_HANDOFF16 || ROWS == RESEED_SOA ||
                              ROWS == RESEED_LDS5) {
                    WorldRows uncached;
                    uncached.cached = false;
                    if (reset_hit_world(K, *wt, uncached, cx, cy)) continue;
                } else {
                    if (reset_hit<true>(K, t, cx, cy)) continue;
                }
                bx = cx; by = cy; bt = heading;
                break;
            }
        }
        const float rbx = __shfl(bx, gbase), rby = __shfl(by, gbase), rbt = __shfl(bt, gbase);
        if (serial) { bx = rbx; by = rby; bt = rbt; }
    }
    EnvState e;
    e.x = bx; e.y = by; e.th = bt; e.gx = gx; e.gy = gy;
    e.wx = W * u_pm1(rw[1]);
    e.wy = W * u_pm1(rw[2]);
    e.t = 0;
    return e;
}

}  // namespace aqua
