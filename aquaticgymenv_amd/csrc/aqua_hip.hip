// aqua_hip.hip -- kernels and C ABI (include/aqua_hip.h) of the batched AquaEnv hot path, gfx950 only.
//
// Kernels
//   step_kernel<VEC, AK>   one fused launch per batched step (reference: AquaEnv.step,
//                          gym_aqua/envs/aqua.py:135-213).  One lane advances VEC consecutive worlds
//                          (16-byte coalesced loads/stores on the SoA rows when VEC == 4), the obstacle
//                          table is staged into LDS once per workgroup, done flags are packed with
//                          wavefront ballots, finished worlds are re-seeded in the same launch.
//   rollout_kernel<AK>     T steps in one launch with the world state held in registers.
//   reset_kernel           masked reset (reference: AquaEnv.reset, aqua.py:100-126).
//   tick_kernel            *tick_base += delta (tail node of a captured rollout graph).
// HBM-bound integer/float streaming work: no MFMA anywhere (there is no contraction to feed it).
#include <hip/hip_runtime.h>

#include <atomic>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstring>

#include "../../include/aqua_hip.h"
#include "aqua_device.hpp"

using namespace aqua;

namespace {

constexpr int BLOCK = 256;
constexpr int MAX_GRID = 256 * 8;        // 256 CUs x 8 workgroups: grid-stride beyond that

struct StepArgs {
    float* state;
    int64_t ld;
    int32_t* time;
    const void* action;
    int64_t action_ld;
    const float* noise;
    int64_t noise_ld;
    float* reward;
    uint8_t* term;
    uint64_t* done_bits;
    const void* obst_blob;
    const uint64_t* tick_base;
    uint64_t seed, tick;
    int64_t N, env_offset;
    // rollout only
    int64_t T, action_step_stride, out_step_stride;
    int K, waves, time_limit, auto_reset, random_boat, random_goal;
    float W, sigma;
};

// ------------------------------------------------------------------ vector load/store helpers
template <int VEC> struct VecF;
template <> struct VecF<1> { using type = float; };
template <> struct VecF<2> { using type = float2; };
template <> struct VecF<4> { using type = float4; };

template <int VEC, typename T>
__device__ __forceinline__ void load_row(const T* __restrict__ p, int64_t i0, int64_t n, T (&v)[VEC])
{
    if constexpr (VEC == 1) {
        v[0] = i0 < n ? p[i0] : T(0);
    } else {
        if (i0 + VEC <= n) {
            struct alignas(sizeof(T) * VEC) Pack { T e[VEC]; };
            const Pack q = *reinterpret_cast<const Pack*>(p + i0);
#pragma unroll
            for (int j = 0; j < VEC; ++j) v[j] = q.e[j];
        } else {
#pragma unroll
            for (int j = 0; j < VEC; ++j) v[j] = (i0 + j < n) ? p[i0 + j] : T(0);
        }
    }
}

template <int VEC, typename T>
__device__ __forceinline__ void store_row(T* __restrict__ p, int64_t i0, int64_t n, const T (&v)[VEC])
{
    if constexpr (VEC == 1) {
        if (i0 < n) p[i0] = v[0];
    } else {
        if (i0 + VEC <= n) {
            struct alignas(sizeof(T) * VEC) Pack { T e[VEC]; };
            Pack q;
#pragma unroll
            for (int j = 0; j < VEC; ++j) q.e[j] = v[j];
            *reinterpret_cast<Pack*>(p + i0) = q;
        } else {
#pragma unroll
            for (int j = 0; j < VEC; ++j)
                if (i0 + j < n) p[i0 + j] = v[j];
        }
    }
}

// spread the low 64/VEC bits of x so that bit i lands at bit i*VEC (wave-uniform, runs on the SALU)
template <int VEC>
__device__ __forceinline__ uint64_t spread_bits(uint64_t x)
{
    if constexpr (VEC == 1) return x;
    if constexpr (VEC == 2) {
        x &= 0xFFFFFFFFull;
        x = (x | (x << 16)) & 0x0000FFFF0000FFFFull;
        x = (x | (x << 8)) & 0x00FF00FF00FF00FFull;
        x = (x | (x << 4)) & 0x0F0F0F0F0F0F0F0Full;
        x = (x | (x << 2)) & 0x3333333333333333ull;
        x = (x | (x << 1)) & 0x5555555555555555ull;
        return x;
    }
    if constexpr (VEC == 4) {
        x &= 0xFFFFull;
        x = (x | (x << 24)) & 0x000000FF000000FFull;
        x = (x | (x << 12)) & 0x000F000F000F000Full;
        x = (x | (x << 6)) & 0x0303030303030303ull;
        x = (x | (x << 3)) & 0x1111111111111111ull;
        return x;
    }
    return 0;
}

__device__ __forceinline__ void stage_obstacles(ObstF* s_obst, const void* blob, int K)
{
    const float4* src = reinterpret_cast<const float4*>(blob);
    float4* dst = reinterpret_cast<float4*>(s_obst);
    for (int i = threadIdx.x; i < 2 * K; i += BLOCK) dst[i] = src[i];
    __syncthreads();
}

__device__ __forceinline__ StepConst make_const(const StepArgs& a, const ObstF* s_obst)
{
    StepConst k;
    k.W = a.W; k.sigma = a.sigma; k.waves = a.waves; k.time_limit = a.time_limit; k.K = a.K;
    k.obst = s_obst;
    k.obst64 = reinterpret_cast<const double*>(reinterpret_cast<const char*>(a.obst_blob) + sizeof(ObstF) * a.K);
    return k;
}

// discrete index as the reference's list lookup sees it (aqua.py:154): -3..-1 wrap; the rest clamps
__device__ __forceinline__ int fold_index(int64_t a)
{
    if (a < 0) a += 3;
    return a < 0 ? 0 : (a > 2 ? 2 : static_cast<int>(a));
}

struct Motion {          // per world: what the action means
    float h, w, chord;
    float vl, vr;        // continuous: thrusts as given (float32, unclipped); discrete: unused
    int idx;             // discrete: 0..2
};

template <int AK>
__device__ __forceinline__ Motion decode_motion(const StepConst& k, int idx, float vl, float vr)
{
    Motion m;
    m.idx = idx; m.vl = vl; m.vr = vr;
    if constexpr (AK == AQUA_ACT_F32X2 || AK == AQUA_ACT_SAMPLE_C) {
        const float cl = fminf(fmaxf(vl, 0.2f), 0.5f), cr = fminf(fmaxf(vr, 0.2f), 0.5f);   // aqua.py:145-150
        thrust_to_motion(cl, cr, m.h, m.w, m.chord);
    } else {
        m.h = idx == 0 ? ACT_H_TURN : (idx == 1 ? -ACT_H_TURN : ACT_H_LINE);
        m.w = idx == 0 ? ACT_W_TURN : (idx == 1 ? -ACT_W_TURN : ACT_W_LINE);
        m.chord = idx == 2 ? ACT_C_LINE : ACT_C_TURN;
    }
    return m;
}

template <int AK>
__device__ __forceinline__ void exact_thrusts(const Motion& m, double& vl, double& vr)
{
    if constexpr (AK == AQUA_ACT_F32X2 || AK == AQUA_ACT_SAMPLE_C) {
        vl = fmin(fmax(static_cast<double>(m.vl), 0.2), 0.5);
        vr = fmin(fmax(static_cast<double>(m.vr), 0.2), 0.5);
    } else {
        vl = m.idx == 0 ? 0.2 : 0.5;     // aqua.py:36-41
        vr = m.idx == 1 ? 0.2 : 0.5;
    }
}

__device__ __forceinline__ int sample_discrete(uint32_t r) { return static_cast<int>((static_cast<uint64_t>(r >> 8) * 3u) >> 24); }
__device__ __forceinline__ float sample_thrust(uint32_t r) { return fmaf(0.3f, u_01(r), 0.2f); }

// ------------------------------------------------------------------ one launch per step
template <int VEC, int AK>
__global__ __launch_bounds__(BLOCK) void step_kernel(const StepArgs a)
{
    __shared__ ObstF s_obst[MAX_OBST];
    stage_obstacles(s_obst, a.obst_blob, a.K);
    const StepConst k = make_const(a, s_obst);
    const uint64_t tick = a.tick + (a.tick_base ? *a.tick_base : 0ull);
    const int64_t N = a.N, ld = a.ld;
    const int64_t n_items = (N + VEC - 1) / VEC;
    const int lane = threadIdx.x & 63;
    const int64_t n_words = (N + 63) >> 6;

    for (int64_t wave_item = (static_cast<int64_t>(blockIdx.x) * BLOCK + (threadIdx.x & ~63)); wave_item < n_items;
         wave_item += static_cast<int64_t>(gridDim.x) * BLOCK) {
        const int64_t i0 = (wave_item + lane) * VEC;

        float x[VEC], y[VEC], th[VEC], gx[VEC], gy[VEC], wx[VEC], wy[VEC];
        int32_t t[VEC];
        load_row<VEC>(a.state + 0 * ld, i0, N, x);
        load_row<VEC>(a.state + 1 * ld, i0, N, y);
        load_row<VEC>(a.state + 2 * ld, i0, N, th);
        load_row<VEC>(a.state + 3 * ld, i0, N, gx);
        load_row<VEC>(a.state + 4 * ld, i0, N, gy);
        load_row<VEC>(a.state + 5 * ld, i0, N, wx);
        load_row<VEC>(a.state + 6 * ld, i0, N, wy);
        load_row<VEC>(a.time, i0, N, t);

        int aidx[VEC];
        float avl[VEC], avr[VEC];
#pragma unroll
        for (int j = 0; j < VEC; ++j) { aidx[j] = 2; avl[j] = 0.5f; avr[j] = 0.5f; }
        if constexpr (AK == AQUA_ACT_U8) {
            uint8_t v[VEC];
            load_row<VEC>(static_cast<const uint8_t*>(a.action), i0, N, v);
#pragma unroll
            for (int j = 0; j < VEC; ++j) aidx[j] = fold_index(v[j]);
        } else if constexpr (AK == AQUA_ACT_I32) {
            int32_t v[VEC];
            load_row<VEC>(static_cast<const int32_t*>(a.action), i0, N, v);
#pragma unroll
            for (int j = 0; j < VEC; ++j) aidx[j] = fold_index(v[j]);
        } else if constexpr (AK == AQUA_ACT_I64) {
#pragma unroll
            for (int j = 0; j < VEC; ++j)
                aidx[j] = fold_index((i0 + j < N) ? static_cast<const int64_t*>(a.action)[i0 + j] : 2);
        } else if constexpr (AK == AQUA_ACT_F32X2) {
            load_row<VEC>(static_cast<const float*>(a.action), i0, N, avl);
            load_row<VEC>(static_cast<const float*>(a.action) + a.action_ld, i0, N, avr);
        }

        float u0[VEC], u1[VEC];
        if (a.noise != nullptr) {
            load_row<VEC>(a.noise, i0, N, u0);
            load_row<VEC>(a.noise + a.noise_ld, i0, N, u1);
        }
        if (a.noise == nullptr || AK >= AQUA_ACT_SAMPLE_D) {
#pragma unroll
            for (int j = 0; j < VEC; ++j) {
                uint32_t r[4];
                draw(a.seed, static_cast<uint64_t>(a.env_offset + i0 + j), tick, STREAM_STEP, 0, r);
                if (a.noise == nullptr) { u0[j] = u_pm1(r[0]); u1[j] = u_pm1(r[1]); }
                if constexpr (AK == AQUA_ACT_SAMPLE_D) aidx[j] = sample_discrete(r[2]);
                if constexpr (AK == AQUA_ACT_SAMPLE_C) { avl[j] = sample_thrust(r[2]); avr[j] = sample_thrust(r[3]); }
            }
        }

        float rew[VEC];
        uint8_t code[VEC];
        Motion mo[VEC];
        uint32_t knife_mask = 0, done_mask = 0;
#pragma unroll
        for (int j = 0; j < VEC; ++j) {
            EnvState e{x[j], y[j], th[j], gx[j], gy[j], wx[j], wy[j], t[j]};
            mo[j] = decode_motion<AK>(k, aidx[j], avl[j], avr[j]);
            uint32_t c;
            const bool knife = fast_step(e, mo[j].h, mo[j].w, mo[j].chord, u0[j], u1[j], k, rew[j], c);
            const bool valid = i0 + j < N;
            knife_mask |= (knife && valid) ? (1u << j) : 0u;
            code[j] = static_cast<uint8_t>(c);
            x[j] = e.x; y[j] = e.y; th[j] = e.th; wx[j] = e.wx; wy[j] = e.wy; t[j] = e.t;
        }
        // knife-edge worlds: redo pose, reward and termination in float64 from the inputs still in memory
        if (__any(knife_mask != 0)) {
#pragma unroll
            for (int j = 0; j < VEC; ++j) {
                if (knife_mask & (1u << j)) {
                    const int64_t i = i0 + j;
                    double vl, vr;
                    exact_thrusts<AK>(mo[j], vl, vr);
                    const ExactOut o = exact_step(a.state[0 * ld + i], a.state[1 * ld + i], a.state[2 * ld + i], gx[j],
                                                  gy[j], a.state[5 * ld + i], a.state[6 * ld + i], t[j], vl, vr, k.K,
                                                  k.obst64, k.time_limit);
                    x[j] = o.x; y[j] = o.y; th[j] = o.th; rew[j] = o.reward; code[j] = static_cast<uint8_t>(o.term);
                }
            }
        }
#pragma unroll
        for (int j = 0; j < VEC; ++j) done_mask |= (code[j] != 0 && i0 + j < N) ? (1u << j) : 0u;

        // outputs of the step that just happened
        store_row<VEC>(a.reward, i0, N, rew);
        store_row<VEC>(a.term, i0, N, code);
        if (a.done_bits != nullptr) {
            uint64_t mine = 0;
#pragma unroll
            for (int j = 0; j < VEC; ++j) {
                const uint64_t b = __ballot((done_mask >> j) & 1u);
#pragma unroll
                for (int wq = 0; wq < VEC; ++wq) {
                    const uint64_t piece = spread_bits<VEC>(b >> (wq * (64 / VEC))) << j;
                    if (lane == wq) mine |= piece;
                }
            }
            const int64_t word = (wave_item / 64) * VEC + lane;
            if (lane < VEC && word < n_words) a.done_bits[word] = mine;
        }

        // finished worlds start a new episode inside the same launch
        if (a.auto_reset && __any(done_mask != 0)) {
#pragma unroll
            for (int j = 0; j < VEC; ++j) {
                if (done_mask & (1u << j)) {
                    const EnvState e = reset_env(a.seed, static_cast<uint64_t>(a.env_offset + i0 + j), tick, k.waves,
                                                 a.random_boat, a.random_goal, k.K, k.obst);
                    x[j] = e.x; y[j] = e.y; th[j] = e.th; gx[j] = e.gx; gy[j] = e.gy; wx[j] = e.wx; wy[j] = e.wy;
                    t[j] = e.t;
                }
            }
            if (done_mask != 0) {
                store_row<VEC>(a.state + 3 * ld, i0, N, gx);
                store_row<VEC>(a.state + 4 * ld, i0, N, gy);
            }
        }
        store_row<VEC>(a.state + 0 * ld, i0, N, x);
        store_row<VEC>(a.state + 1 * ld, i0, N, y);
        store_row<VEC>(a.state + 2 * ld, i0, N, th);
        store_row<VEC>(a.state + 5 * ld, i0, N, wx);
        store_row<VEC>(a.state + 6 * ld, i0, N, wy);
        store_row<VEC>(a.time, i0, N, t);
    }
}

// ------------------------------------------------------------------ T steps in one launch
template <int AK>
__global__ __launch_bounds__(BLOCK) void rollout_kernel(const StepArgs a)
{
    __shared__ ObstF s_obst[MAX_OBST];
    stage_obstacles(s_obst, a.obst_blob, a.K);
    const StepConst k = make_const(a, s_obst);
    const uint64_t tick0 = a.tick + (a.tick_base ? *a.tick_base : 0ull);
    const int64_t N = a.N, ld = a.ld;
    for (int64_t i = static_cast<int64_t>(blockIdx.x) * BLOCK + threadIdx.x; i < N;
         i += static_cast<int64_t>(gridDim.x) * BLOCK) {
        EnvState e{a.state[0 * ld + i], a.state[1 * ld + i], a.state[2 * ld + i], a.state[3 * ld + i],
                   a.state[4 * ld + i], a.state[5 * ld + i], a.state[6 * ld + i], a.time[i]};
        const uint64_t env = static_cast<uint64_t>(a.env_offset + i);
        for (int64_t s = 0; s < a.T; ++s) {
            const uint64_t tick = tick0 + static_cast<uint64_t>(s);
            int idx = 2;
            float vl = 0.5f, vr = 0.5f;
            if constexpr (AK == AQUA_ACT_U8) idx = fold_index(static_cast<const uint8_t*>(a.action)[s * a.action_step_stride + i]);
            if constexpr (AK == AQUA_ACT_I32) idx = fold_index(static_cast<const int32_t*>(a.action)[s * a.action_step_stride + i]);
            if constexpr (AK == AQUA_ACT_I64) idx = fold_index(static_cast<const int64_t*>(a.action)[s * a.action_step_stride + i]);
            if constexpr (AK == AQUA_ACT_F32X2) {
                const float* base = static_cast<const float*>(a.action) + s * a.action_step_stride;
                vl = base[i]; vr = base[a.action_ld + i];
            }
            uint32_t r[4];
            draw(a.seed, env, tick, STREAM_STEP, 0, r);
            const float u0 = u_pm1(r[0]), u1 = u_pm1(r[1]);
            if constexpr (AK == AQUA_ACT_SAMPLE_D) idx = sample_discrete(r[2]);
            if constexpr (AK == AQUA_ACT_SAMPLE_C) { vl = sample_thrust(r[2]); vr = sample_thrust(r[3]); }
            const Motion m = decode_motion<AK>(k, idx, vl, vr);
            const EnvState before = e;
            float rew;
            uint32_t code;
            const bool knife = fast_step(e, m.h, m.w, m.chord, u0, u1, k, rew, code);
            if (__any(knife)) {
                if (knife) {
                    double dl, dr;
                    exact_thrusts<AK>(m, dl, dr);
                    const ExactOut o = exact_step(before.x, before.y, before.th, before.gx, before.gy, before.wx,
                                                  before.wy, e.t, dl, dr, k.K, k.obst64, k.time_limit);
                    e.x = o.x; e.y = o.y; e.th = o.th; rew = o.reward; code = o.term;
                }
            }
            a.reward[s * a.out_step_stride + i] = rew;
            a.term[s * a.out_step_stride + i] = static_cast<uint8_t>(code);
            if (a.auto_reset && __any(code != 0)) {
                if (code != 0) e = reset_env(a.seed, env, tick, k.waves, a.random_boat, a.random_goal, k.K, k.obst);
            }
        }
        a.state[0 * ld + i] = e.x; a.state[1 * ld + i] = e.y; a.state[2 * ld + i] = e.th;
        a.state[3 * ld + i] = e.gx; a.state[4 * ld + i] = e.gy;
        a.state[5 * ld + i] = e.wx; a.state[6 * ld + i] = e.wy;
        a.time[i] = e.t;
    }
}

// ------------------------------------------------------------------ masked reset
__global__ __launch_bounds__(BLOCK) void reset_kernel(const StepArgs a, const uint8_t* __restrict__ mask)
{
    __shared__ ObstF s_obst[MAX_OBST];
    stage_obstacles(s_obst, a.obst_blob, a.K);
    const uint64_t tick = a.tick + (a.tick_base ? *a.tick_base : 0ull);
    const int64_t N = a.N, ld = a.ld;
    for (int64_t i = static_cast<int64_t>(blockIdx.x) * BLOCK + threadIdx.x; i < N;
         i += static_cast<int64_t>(gridDim.x) * BLOCK) {
        if (mask != nullptr && mask[i] == 0) continue;
        const EnvState e = reset_env(a.seed, static_cast<uint64_t>(a.env_offset + i), tick, a.waves, a.random_boat,
                                     a.random_goal, a.K, s_obst);
        a.state[0 * ld + i] = e.x; a.state[1 * ld + i] = e.y; a.state[2 * ld + i] = e.th;
        a.state[3 * ld + i] = e.gx; a.state[4 * ld + i] = e.gy;
        a.state[5 * ld + i] = e.wx; a.state[6 * ld + i] = e.wy;
        a.time[i] = e.t;
    }
}

__global__ void tick_kernel(uint64_t* tick_base, uint64_t delta) { *tick_base += delta; }

// ------------------------------------------------------------------ host side
thread_local char g_err[512] = "";
std::atomic<int> g_vec_override{0};

int fail(int code, const char* fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}

int hip_fail(hipError_t e, const char* what)
{
    snprintf(g_err, sizeof(g_err), "%s: %s", what, hipGetErrorString(e));
    return static_cast<int>(e);
}

bool aligned(const void* p, size_t a) { return (reinterpret_cast<uintptr_t>(p) % a) == 0; }

int grid_for(int64_t items)
{
    int64_t g = (items + BLOCK - 1) / BLOCK;
    if (g < 1) g = 1;
    if (g > MAX_GRID) g = MAX_GRID;
    return static_cast<int>(g);
}

int fill_args(StepArgs& a, const AquaParams* p, const void* blob, int K, int64_t N, int64_t env_offset, float* state,
              int64_t ld, int32_t* time, uint64_t seed, uint64_t tick, const uint64_t* tick_base)
{
    if (p == nullptr) return fail(AQUA_E_INVALID, "params is NULL");
    if (N < 0 || ld < N) return fail(AQUA_E_INVALID, "bad sizes: N=%lld ld=%lld", (long long)N, (long long)ld);
    if (K < 0 || K > AQUA_MAX_OBSTACLES) return fail(AQUA_E_INVALID, "K=%d outside [0, %d]", K, AQUA_MAX_OBSTACLES);
    if (K > 0 && blob == nullptr) return fail(AQUA_E_INVALID, "K=%d but obstacle blob is NULL", K);
    if (K > 0 && !aligned(blob, 16)) return fail(AQUA_E_ALIGN, "obstacle blob must be 16-byte aligned");
    if (N > 0 && (state == nullptr || time == nullptr)) return fail(AQUA_E_INVALID, "state/time is NULL");
    if (!aligned(state, 4) || !aligned(time, 4)) return fail(AQUA_E_ALIGN, "state/time must be 4-byte aligned");
    if (env_offset < 0) return fail(AQUA_E_INVALID, "env_offset < 0");
    std::memset(&a, 0, sizeof(a));
    a.state = state; a.ld = ld; a.time = time; a.obst_blob = blob; a.tick_base = tick_base;
    a.seed = seed; a.tick = tick; a.N = N; a.env_offset = env_offset; a.K = K;
    a.waves = p->waves; a.time_limit = p->time_limit;
    a.random_boat = p->random_boat; a.random_goal = p->random_goal;
    a.W = static_cast<float>(0.05 * p->waves);
    a.sigma = static_cast<float>(0.001 * p->waves);
    return 0;
}

size_t action_elem_bytes(int kind)
{
    switch (kind) {
        case AQUA_ACT_U8: return 1;
        case AQUA_ACT_I32: return 4;
        case AQUA_ACT_I64: return 8;
        case AQUA_ACT_F32X2: return 4;
        default: return 0;
    }
}

int pick_vec(const float* state, int64_t ld, const int32_t* time, const float* reward, const void* action,
             int action_kind, int64_t action_ld, const float* noise, int64_t noise_ld, const uint8_t* term, int64_t N)
{
    int want = g_vec_override.load();
    if (want == 0) want = N >= 4 * BLOCK * 256 ? 4 : (N >= 2 * BLOCK * 256 ? 2 : 1);
    for (int v = want; v > 1; v >>= 1) {
        bool ok = aligned(state, 4 * v) && aligned(time, 4 * v) && aligned(reward, 4 * v) && (ld % v == 0) &&
                  aligned(term, v);
        if (action_kind == AQUA_ACT_U8) ok = ok && aligned(action, v);
        if (action_kind == AQUA_ACT_I32) ok = ok && aligned(action, 4 * v);
        if (action_kind == AQUA_ACT_F32X2) ok = ok && aligned(action, 4 * v) && (action_ld % v == 0);
        if (noise != nullptr) ok = ok && aligned(noise, 4 * v) && (noise_ld % v == 0);
        if (ok) return v;
    }
    return 1;
}

template <int VEC>
hipError_t launch_step(const StepArgs& a, int kind, hipStream_t s)
{
    const int64_t items = (a.N + VEC - 1) / VEC;
    const dim3 grid(grid_for(items)), block(BLOCK);
    switch (kind) {
        case AQUA_ACT_U8: hipLaunchKernelGGL((step_kernel<VEC, AQUA_ACT_U8>), grid, block, 0, s, a); break;
        case AQUA_ACT_I32: hipLaunchKernelGGL((step_kernel<VEC, AQUA_ACT_I32>), grid, block, 0, s, a); break;
        case AQUA_ACT_I64: hipLaunchKernelGGL((step_kernel<VEC, AQUA_ACT_I64>), grid, block, 0, s, a); break;
        case AQUA_ACT_F32X2: hipLaunchKernelGGL((step_kernel<VEC, AQUA_ACT_F32X2>), grid, block, 0, s, a); break;
        case AQUA_ACT_SAMPLE_D: hipLaunchKernelGGL((step_kernel<VEC, AQUA_ACT_SAMPLE_D>), grid, block, 0, s, a); break;
        case AQUA_ACT_SAMPLE_C: hipLaunchKernelGGL((step_kernel<VEC, AQUA_ACT_SAMPLE_C>), grid, block, 0, s, a); break;
        default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

hipError_t launch_step_any(const StepArgs& a, int kind, int vec, hipStream_t s)
{
    if (vec == 4) return launch_step<4>(a, kind, s);
    if (vec == 2) return launch_step<2>(a, kind, s);
    return launch_step<1>(a, kind, s);
}

int check_step_buffers(int64_t N, const void* action, int action_kind, int64_t action_ld, const float* noise,
                       int64_t noise_ld, const float* reward, const uint8_t* term)
{
    if (action_kind < AQUA_ACT_U8 || action_kind > AQUA_ACT_SAMPLE_C)
        return fail(AQUA_E_INVALID, "unknown action_kind %d", action_kind);
    if (N > 0 && action_kind <= AQUA_ACT_F32X2 && action == nullptr) return fail(AQUA_E_INVALID, "action is NULL");
    if (action_kind == AQUA_ACT_F32X2 && action_ld < N) return fail(AQUA_E_INVALID, "action_ld < N");
    if (action_kind <= AQUA_ACT_F32X2 && !aligned(action, action_elem_bytes(action_kind)))
        return fail(AQUA_E_ALIGN, "action pointer not aligned to its element size");
    if (noise != nullptr && noise_ld < N) return fail(AQUA_E_INVALID, "noise_ld < N");
    if (noise != nullptr && !aligned(noise, 4)) return fail(AQUA_E_ALIGN, "noise must be 4-byte aligned");
    if (N > 0 && (reward == nullptr || term == nullptr)) return fail(AQUA_E_INVALID, "reward/term is NULL");
    if (!aligned(reward, 4)) return fail(AQUA_E_ALIGN, "reward must be 4-byte aligned");
    return 0;
}

}  // namespace

struct AquaGraph {
    hipGraph_t graph;
    hipGraphExec_t exec;
};

extern "C" {

int aqua_version(void) { return AQUA_ABI_VERSION; }

const char* aqua_last_error(void) { return g_err; }

size_t aqua_obstacle_blob_bytes(int K)
{
    if (K <= 0) return 0;
    return static_cast<size_t>(K) * (sizeof(ObstF) + 5 * sizeof(double));
}

int aqua_pack_obstacles(const double* rows, int K, void* blob_host, size_t blob_bytes)
{
    if (K < 0 || K > AQUA_MAX_OBSTACLES) return fail(AQUA_E_INVALID, "K=%d outside [0, %d]", K, AQUA_MAX_OBSTACLES);
    if (K == 0) return 0;
    if (rows == nullptr || blob_host == nullptr) return fail(AQUA_E_INVALID, "rows/blob is NULL");
    if (blob_bytes < aqua_obstacle_blob_bytes(K)) return fail(AQUA_E_INVALID, "blob too small");
    ObstF* f = static_cast<ObstF*>(blob_host);
    double* d = reinterpret_cast<double*>(static_cast<char*>(blob_host) + sizeof(ObstF) * K);
    const double band = static_cast<double>(BAND);
    for (int k = 0; k < K; ++k) {
        const double* o = rows + 5 * k;
        if (!(o[2] == 0.0 || o[2] == 1.0)) return fail(AQUA_E_INVALID, "obstacle %d: kind must be 0 or 1", k);
        double hx = 0, hy = 0, R = 2.5;                 // boat radius (aqua.py:75)
        if (o[2] == 0.0) R += o[3]; else { hx = o[3] / 2; hy = o[4] / 2; }
        f[k].lox = static_cast<float>(o[0] - hx); f[k].hix = static_cast<float>(o[0] + hx);
        f[k].loy = static_cast<float>(o[1] - hy); f[k].hiy = static_cast<float>(o[1] + hy);
        const double lo = R - band > 0 ? R - band : 0.0;
        f[k].lo2 = static_cast<float>(lo * lo);
        f[k].hi2 = static_cast<float>((R + band) * (R + band));
        f[k].r2 = static_cast<float>(R * R);
        f[k].pad = 0.0f;
        for (int j = 0; j < 5; ++j) d[5 * k + j] = o[j];
    }
    return 0;
}

int aqua_step_vector_width(const float* state, int64_t ld, const int32_t* time, const float* reward,
                           const void* action, int action_kind, int64_t action_ld, const float* noise,
                           int64_t noise_ld, const uint8_t* term)
{
    return pick_vec(state, ld, time, reward, action, action_kind, action_ld, noise, noise_ld, term, INT64_MAX);
}

void aqua_discrete_constants(float out[9])
{
    const float v[9] = {ACT_H_TURN, -ACT_H_TURN, ACT_H_LINE, ACT_W_TURN, -ACT_W_TURN, ACT_W_LINE,
                        ACT_C_TURN, ACT_C_TURN, ACT_C_LINE};
    for (int j = 0; j < 9; ++j) out[j] = v[j];
}

int aqua_set_vector_width(int width)
{
    if (!(width == 0 || width == 1 || width == 2 || width == 4)) return fail(AQUA_E_INVALID, "width must be 0, 1, 2 or 4");
    return g_vec_override.exchange(width);
}

int aqua_step_f32(const AquaParams* p, const void* obst_blob_dev, int K, int64_t N, int64_t env_offset,
                  float* state, int64_t ld, int32_t* time, const void* action, int action_kind,
                  int64_t action_ld, const float* noise, int64_t noise_ld, uint64_t seed, uint64_t tick,
                  const uint64_t* tick_base_dev, float* reward, uint8_t* term, uint64_t* done_bits,
                  int auto_reset, void* stream)
{
    StepArgs a;
    int rc = fill_args(a, p, obst_blob_dev, K, N, env_offset, state, ld, time, seed, tick, tick_base_dev);
    if (rc) return rc;
    rc = check_step_buffers(N, action, action_kind, action_ld, noise, noise_ld, reward, term);
    if (rc) return rc;
    if (done_bits != nullptr && !aligned(done_bits, 8)) return fail(AQUA_E_ALIGN, "done_bits must be 8-byte aligned");
    if (N == 0) return 0;
    a.action = action; a.action_ld = action_ld; a.noise = noise; a.noise_ld = noise_ld;
    a.reward = reward; a.term = term; a.done_bits = done_bits; a.auto_reset = auto_reset;
    const int vec = pick_vec(state, ld, time, reward, action, action_kind, action_ld, noise, noise_ld, term, N);
    const hipError_t e = launch_step_any(a, action_kind, vec, static_cast<hipStream_t>(stream));
    return e == hipSuccess ? 0 : hip_fail(e, "aqua_step_f32 launch");
}

int aqua_reset_f32(const AquaParams* p, const void* obst_blob_dev, int K, int64_t N, int64_t env_offset,
                   float* state, int64_t ld, int32_t* time, const uint8_t* mask, uint64_t seed, uint64_t tick,
                   const uint64_t* tick_base_dev, void* stream)
{
    StepArgs a;
    const int rc = fill_args(a, p, obst_blob_dev, K, N, env_offset, state, ld, time, seed, tick, tick_base_dev);
    if (rc) return rc;
    if (N == 0) return 0;
    hipLaunchKernelGGL(reset_kernel, dim3(grid_for(N)), dim3(BLOCK), 0, static_cast<hipStream_t>(stream), a, mask);
    const hipError_t e = hipGetLastError();
    return e == hipSuccess ? 0 : hip_fail(e, "aqua_reset_f32 launch");
}

int aqua_rollout_f32(const AquaParams* p, const void* obst_blob_dev, int K, int64_t N, int64_t env_offset,
                     float* state, int64_t ld, int32_t* time, int64_t T, const void* actions, int action_kind,
                     int64_t action_ld, int64_t action_step_stride, uint64_t seed, uint64_t tick,
                     const uint64_t* tick_base_dev, float* reward, uint8_t* term, int64_t out_step_stride,
                     uint64_t* done_bits, int64_t done_step_stride, int auto_reset, void* stream)
{
    StepArgs a;
    int rc = fill_args(a, p, obst_blob_dev, K, N, env_offset, state, ld, time, seed, tick, tick_base_dev);
    if (rc) return rc;
    rc = check_step_buffers(N, actions, action_kind, action_ld, nullptr, 0, reward, term);
    if (rc) return rc;
    if (T < 0 || action_step_stride < 0 || out_step_stride < 0 || done_step_stride < 0)
        return fail(AQUA_E_INVALID, "negative T or stride");
    if (N == 0 || T == 0) return 0;
    a.action_ld = action_ld; a.auto_reset = auto_reset;
    const size_t esz = action_elem_bytes(action_kind);
    for (int64_t t = 0; t < T; ++t) {
        a.tick = tick + static_cast<uint64_t>(t);
        a.action = actions ? static_cast<const char*>(actions) + static_cast<size_t>(t * action_step_stride) * esz : nullptr;
        a.reward = reward + t * out_step_stride;
        a.term = term + t * out_step_stride;
        a.done_bits = done_bits ? done_bits + t * done_step_stride : nullptr;
        const int vec = pick_vec(state, ld, time, a.reward, a.action, action_kind, action_ld, nullptr, 0, a.term, N);
        const hipError_t e = launch_step_any(a, action_kind, vec, static_cast<hipStream_t>(stream));
        if (e != hipSuccess) return hip_fail(e, "aqua_rollout_f32 launch");
    }
    return 0;
}

int aqua_rollout_fused_f32(const AquaParams* p, const void* obst_blob_dev, int K, int64_t N, int64_t env_offset,
                           float* state, int64_t ld, int32_t* time, int64_t T, const void* actions,
                           int action_kind, int64_t action_ld, int64_t action_step_stride, uint64_t seed,
                           uint64_t tick, const uint64_t* tick_base_dev, float* reward, uint8_t* term,
                           int64_t out_step_stride, int auto_reset, void* stream)
{
    StepArgs a;
    int rc = fill_args(a, p, obst_blob_dev, K, N, env_offset, state, ld, time, seed, tick, tick_base_dev);
    if (rc) return rc;
    rc = check_step_buffers(N, actions, action_kind, action_ld, nullptr, 0, reward, term);
    if (rc) return rc;
    if (T < 0 || action_step_stride < 0 || out_step_stride < 0) return fail(AQUA_E_INVALID, "negative T or stride");
    if (N == 0 || T == 0) return 0;
    a.action = actions; a.action_ld = action_ld; a.action_step_stride = action_step_stride;
    a.reward = reward; a.term = term; a.out_step_stride = out_step_stride; a.T = T; a.auto_reset = auto_reset;
    const dim3 grid(grid_for(N)), block(BLOCK);
    hipStream_t s = static_cast<hipStream_t>(stream);
    switch (action_kind) {
        case AQUA_ACT_U8: hipLaunchKernelGGL((rollout_kernel<AQUA_ACT_U8>), grid, block, 0, s, a); break;
        case AQUA_ACT_I32: hipLaunchKernelGGL((rollout_kernel<AQUA_ACT_I32>), grid, block, 0, s, a); break;
        case AQUA_ACT_I64: hipLaunchKernelGGL((rollout_kernel<AQUA_ACT_I64>), grid, block, 0, s, a); break;
        case AQUA_ACT_F32X2: hipLaunchKernelGGL((rollout_kernel<AQUA_ACT_F32X2>), grid, block, 0, s, a); break;
        case AQUA_ACT_SAMPLE_D: hipLaunchKernelGGL((rollout_kernel<AQUA_ACT_SAMPLE_D>), grid, block, 0, s, a); break;
        case AQUA_ACT_SAMPLE_C: hipLaunchKernelGGL((rollout_kernel<AQUA_ACT_SAMPLE_C>), grid, block, 0, s, a); break;
        default: return fail(AQUA_E_INVALID, "unknown action_kind %d", action_kind);
    }
    const hipError_t e = hipGetLastError();
    return e == hipSuccess ? 0 : hip_fail(e, "aqua_rollout_fused_f32 launch");
}

int aqua_tick_advance(uint64_t* tick_base_dev, uint64_t delta, void* stream)
{
    if (tick_base_dev == nullptr) return fail(AQUA_E_INVALID, "tick_base_dev is NULL");
    hipLaunchKernelGGL(tick_kernel, dim3(1), dim3(1), 0, static_cast<hipStream_t>(stream), tick_base_dev, delta);
    const hipError_t e = hipGetLastError();
    return e == hipSuccess ? 0 : hip_fail(e, "aqua_tick_advance launch");
}

int aqua_graph_begin(void* stream)
{
    const hipError_t e = hipStreamBeginCapture(static_cast<hipStream_t>(stream), hipStreamCaptureModeThreadLocal);
    return e == hipSuccess ? 0 : hip_fail(e, "hipStreamBeginCapture");
}

int aqua_graph_end(void* stream, AquaGraph** out)
{
    if (out == nullptr) return fail(AQUA_E_INVALID, "out is NULL");
    hipGraph_t g = nullptr;
    hipError_t e = hipStreamEndCapture(static_cast<hipStream_t>(stream), &g);
    if (e != hipSuccess) return hip_fail(e, "hipStreamEndCapture");
    hipGraphExec_t x = nullptr;
    e = hipGraphInstantiate(&x, g, nullptr, nullptr, 0);
    if (e != hipSuccess) { (void)hipGraphDestroy(g); return hip_fail(e, "hipGraphInstantiate"); }
    *out = new AquaGraph{g, x};
    return 0;
}

int aqua_graph_launch(AquaGraph* g, void* stream)
{
    if (g == nullptr) return fail(AQUA_E_INVALID, "graph is NULL");
    const hipError_t e = hipGraphLaunch(g->exec, static_cast<hipStream_t>(stream));
    return e == hipSuccess ? 0 : hip_fail(e, "hipGraphLaunch");
}

int aqua_graph_destroy(AquaGraph* g)
{
    if (g == nullptr) return 0;
    (void)hipGraphExecDestroy(g->exec);
    (void)hipGraphDestroy(g->graph);
    delete g;
    return 0;
}

}  // extern "C"
