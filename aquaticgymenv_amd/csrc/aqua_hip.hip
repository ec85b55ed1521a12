// aqua_hip.hip -- kernels and C ABI (include/aqua_hip.h) of the batched AquaEnv hot path, gfx950 only.
//
// Kernels (reference: AquaEnv.step, gym_aqua/envs/aqua.py:135-213; AquaEnv.reset, aqua.py:100-126)
//   step_ns_kernel<AK, SMALL_TABLE, INTERLEAVE, WB>  one launch per batched step, next-step restart (auto_reset 2): the
//                          launch is split by role -- re-seeding blocks (at the head of a one-round grid; one in every five
//                          blocks, on the XCD of its stepping neighbours, in larger grids) and stepping blocks of 256
//                          worlds (one per lane) -- with no synchronisation between the two.  The benchmarked kernel.
//   step_kernel<AK, SMALL_TABLE, RESTART, WB>  one launch per batched step, no restart (auto_reset 0) or restart in the
//                          same launch (auto_reset 1): 1024-world tiles, finished worlds re-seeded after one barrier.
//                          Both take NsArgs (their own slim kernel arguments); every other kernel takes StepArgs.
//   rollout_kernel<AK>     T steps in one launch with the world state held in registers.
//   step_tables_kernel / step_tables_ns_kernel / reset_tables_kernel / rollout_tables*_kernel   the same for batches in
//                          which every world has its own obstacle table.
//   reset_kernel           masked reset.   obs_norm_kernel  the DQN's normalised observation after a reset.
//   ring_write_kernel<T>   one batch of rows into consecutive slots of a replay ring.
//   copy_words_kernel / copy_fanout_kernel   the done-mask blocks' way into (IPC-mapped) receive buffers.
//   tick_kernel            *tick_base += delta (tail node of a captured fused rollout or one-step graph; the per-step
//                          rollouts advance their tick base themselves, tick_housekeeping()).
// All of it is coalesced float/integer streaming work on struct-of-arrays rows; no MFMA anywhere (there is
// no contraction to feed it).  Arithmetic shared by the kernels lives in aqua_device.hpp.
//
// Build-time switches (diagnostic / development builds only, aquaticgymenv_amd/build.py build_variant(); never defined in
// the shipped library): AQUA_STAMPS = 1 | 2 | 3 (in-kernel phase stamps + the micro-benchmark kernels of
// csrc/aqua_tuning.inc), AQUA_NS_NOWORK / AQUA_NS_NOMAIN (one role of the next-step kernel alone), AQUA_ST1_PLAIN,
// AQUA_NS_LATE_LOADS=0 (the next-step kernel without its late loads), AQUA_DEV_U8_ONLY (one action kind instead of
// seven: a sixth of the compile time while a kernel is being worked on).  The tuning constants below (tile sizes, group
// sizes, cache scopes) are plain constants: every alternative that was measured is recorded with its timing in
// profiles/LOG.md.
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>

#include <cmath>
#include <cstdarg>
#include <cstddef>
#include <type_traits>
#include <cstdio>
#include <cstring>
#include <vector>

#include "../../include/aqua_hip.h"
#include "aqua_device.hpp"

using namespace aqua;

namespace {

constexpr int TILE_WORLDS = 1024;        // step kernel: worlds per workgroup tile (256 / 512: slower)
constexpr int BLOCK_SMALL = 256;         // reset / fused-rollout kernels
constexpr int RESET_GROUP = 8;           // lanes that share the re-seeding attempts of one finished world (4: 7.3, 16: 7.4 us)
constexpr int MAX_GRID = 1 << 30;        // step kernel: one workgroup per tile, no grid-stride loop

constexpr size_t AQUA_BLOB_MIN_BYTES = 320;

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
    float* obs_norm;                     // optional [5][ld]: the observation scaled as the reference's AquaStateNormalizer
    const void* obst_blob;
    const uint64_t* tick_base;
    uint64_t seed, tick;
    int64_t N, env_offset;
    // rollout only
    int64_t T, action_step_stride, out_step_stride;
    int64_t reseed_blocks;               // next-step kernel: this many blocks of the grid re-seed (see step_ns_kernel)
    int K, waves, time_limit, auto_reset, random_boat, random_goal;
    float W, sigma;
    // behind everything the prologues read (tick_housekeeping() fetches these itself, in block 0 only)
    uint64_t* tick_copy_to;              // nullable: block 0 stores the tick base it read here      } a captured rollout advances
    uint64_t* tick_bump_to;              // nullable: block 0 stores (tick base it read + tick_bump) } its own tick base (aqua_rollout_f32)
    uint64_t tick_bump;
};

// Kernel arguments of step_ns_kernel -- the benchmarked kernel -- alone.  As a StepArgs (~30 eight-byte fields of which
// this kernel uses half, 64-bit sizes, the seven state rows as base + r * ld) its prologue held 62 argument SGPRs next to
// the 32 of the quick table: 106 SGPRs, 37 of them spilled to VGPR lanes -- 80 v_readlane / v_writelane, 24 on the
// stepping path, each a vector instruction on a path that is bound by vector issue (profiles/r04/isa/).  Here: only what
// the two roles read, in the order they need it; the rows as seven pointers (an access is row pointer + the lane's 32-bit
// byte offset, tile included: no address arithmetic on the scalar unit in front of the loads); 32-bit sizes (the host
// splits batches of more than NS_LAUNCH_MAX_WORLDS worlds into several launches); whatever only a rare branch wants
// (injected noise, the normalised observation, the tick housekeeping) behind the fields the prologue fetches and read by
// that branch itself (kernarg_at()).
struct NsArgs {
    float* row[7];                       // x, y, theta, goal_x, goal_y, wave_x, wave_y
    int32_t* time;
    const void* action;                  // AQUA_ACT_F32X2: the vL row (the vR row: action_hi)
    float* reward;
    uint8_t* term;
    uint64_t* done_bits;
    const void* obst_blob;
    const uint64_t* tick_base;
    uint64_t seed, tick;
    int64_t env_offset;
    uint32_t N, reseed_blocks;
    int32_t K, time_limit;
    float W, sigma;
    uint32_t flags;                      // waves | NS_RANDOM_BOAT | NS_RANDOM_GOAL | NS_HAS_NOISE | NS_HAS_NORM | NS_DONE_WORD_WB
    uint32_t reserved;
    // ---- read by the branches that want them, through the kernel-argument segment pointer
    const float* action_hi;
    const float* noise[2];
    float* norm[5];
    uint64_t* tick_copy_to;
    uint64_t* tick_bump_to;
    uint64_t tick_bump;
};
enum : uint32_t { NS_WAVES_MASK = 0xFFu, NS_RANDOM_BOAT = 1u << 8, NS_RANDOM_GOAL = 1u << 9, NS_HAS_NOISE = 1u << 10,
                  NS_HAS_NORM = 1u << 11, NS_DONE_WORD_WB = 1u << 12 };
// one launch addresses its worlds with 32-bit byte offsets into rows of up to 8-byte elements
constexpr int64_t NS_LAUNCH_MAX_WORLDS = int64_t(1) << 28;

__device__ __forceinline__ int args_waves(const StepArgs& a) { return a.waves; }
__device__ __forceinline__ int args_waves(const NsArgs& a) { return static_cast<int>(a.flags & NS_WAVES_MASK); }
__device__ __forceinline__ int args_random_boat(const StepArgs& a) { return a.random_boat; }
__device__ __forceinline__ int args_random_boat(const NsArgs& a) { return (a.flags & NS_RANDOM_BOAT) != 0; }
__device__ __forceinline__ int args_random_goal(const StepArgs& a) { return a.random_goal; }
__device__ __forceinline__ int args_random_goal(const NsArgs& a) { return (a.flags & NS_RANDOM_GOAL) != 0; }

// ------------------------------------------------------------------ load/store helpers
// Stores carry agent scope (`global_store ... sc1`): they are written through the XCD's L2 while the kernel runs, so the
// end-of-kernel release has almost nothing left to write back (-0.3 us per launch; system scope: the same; non-temporal:
// -0.1; hinted loads: slower -- profiles/LOG.md).
template <typename T>
__device__ __forceinline__ T ld1(const T* p)
{
    return *p;
}
// `wb` (a constant false everywhere but in the kernels of large batches, see STORE_WB_* below): leave the store to the L2's
// write-back instead.
template <typename T>
__device__ __forceinline__ void st1(T* p, T v, bool wb = false)
{
#ifdef AQUA_ST1_PLAIN                    // (timing experiment: every store left to the L2's write-back)
    *p = v;
#else
    if (wb) *p = v;
    else __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#endif
}

// Written through, a store costs the launch nothing at its end, and that is what counts while a step is a few
// microseconds long.  In a batch of millions of worlds the end of the launch no longer matters and the L2 merging the
// partial lines of a tile -- the stepping wavefront's masked rows and the restarted worlds' 4-byte stores -- does:
// st1() as a plain store against agent scope, us per step (profiles/r03/ab_plain_stores.txt; @2 next-step, @1 same-step):
//     worlds     524 288   1 M     2.1 M   4.2 M   8.4 M   12.6 M   16.7 M
//     @2 plain    8.90    13.18    25.3    52.5    91.1    147.3    207.6
//     @2 agent    8.14    12.98    26.8    65.7    94.1    148.3    197.7
//     @1 plain   11.30      -      34.4    69.4   129.0    189.1    252.1
//     @1 agent   10.58      -      35.8    72.7   144.0    213.0    285.8
// (without restarts the two are within 3 % of each other from 8.4 M on and agent scope wins below: it stays.)
constexpr int64_t STORE_WB_SAME_STEP_MIN = int64_t(1) << 21;                                // same-step restart: from here on
constexpr int64_t STORE_WB_NEXT_STEP_MIN = int64_t(1) << 21, STORE_WB_NEXT_STEP_MAX = 3 * (int64_t(1) << 22);   // next-step: in between

// The packed done mask: one 8-byte ballot word per wavefront.  Left to a plain store, the words stay dirty in the XCD's
// L2 and the end-of-kernel release writes them back while nothing else runs: 0.15-0.2 us of a 5 us launch at 262 144
// worlds (5.01 -> 4.86 us next-step, 6.45 -> 6.33 same-step, 4.06 -> 3.87 without restarts; profiles/r03/ab_done_word.txt).
// Written through instead, every word is a partial-line write of its own; from a couple of million worlds on that costs
// more (nothing at 1 M, +0.8 % at 2.1 M and 4.2 M, +1 % at 16.7 M) than the write-back of lines the L2 has merged, so the
// scope follows the batch size.  (The fused rollouts' per-step reward/term stores stay plain: written through, the
// per-world fused rollout without restarts went from 1.73 to 2.05 us per step, the others did not move.)
constexpr int64_t DONE_WORD_WRITE_THROUGH_MAX_WORLDS = int64_t(1) << 20;
__device__ __forceinline__ void store_done_word(uint64_t* p, uint64_t v, int64_t n_worlds)
{
    if (n_worlds <= DONE_WORD_WRITE_THROUGH_MAX_WORLDS) st1(p, v);
    else *p = v;
}

// uniform base + zero-extended 32-bit BYTE offset: the form the backend turns into `global_load v, voffset, s[base]`
// (no 64-bit address arithmetic per lane)
template <typename T>
__device__ __forceinline__ T ld_at(const T* base, uint32_t byte_off)
{
    return ld1(reinterpret_cast<const T*>(reinterpret_cast<const char*>(base) + byte_off));
}
template <typename T>
__device__ __forceinline__ void st_at(T* base, uint32_t byte_off, T v, bool wb = false)
{
    st1(reinterpret_cast<T*>(reinterpret_cast<char*>(base) + byte_off), v, wb);
}

// rows are read with scalar loads from the constant address space (they follow the 32-byte header)
__device__ __forceinline__ ObstPtr obstacle_rows(const void* blob)
{
    return (ObstPtr)(uintptr_t)(static_cast<const char*>(blob) + sizeof(ObstHeader));
}

// (A: StepArgs, or the next-step kernel's own NsArgs)
template <int QUICK = QUICK_NEVER, typename A>
__device__ __forceinline__ StepConst make_const(const A& a, ObstPtr obst)
{
    StepConst k;
    k.quick = nullptr;                  // (qc0 / qr0 are read only when quick is set)
    k.W = a.W; k.sigma = a.sigma; k.waves = args_waves(a); k.time_limit = a.time_limit; k.K = a.K;
    k.obst = obst;
    k.Kc = 0; k.band2 = 0.0f; k.band2_tight = 0.0f;
    k.touch[0] = k.touch[1] = k.touch[2] = k.touch[3] = 0;
    if (a.obst_blob != nullptr) {        // header fields: uniform scalar loads (the pointer is NULL when K == 0)
        const ObstHeader __attribute__((address_space(4)))* h =
            (const ObstHeader __attribute__((address_space(4)))*)(uintptr_t)a.obst_blob;
        k.Kc = h->n_circles;
        k.band2 = h->band2;
        k.band2_tight = h->band2_tight;
        // Touch the table's cache lines now (rows 2l - 1 and 2l share line l; row 0 shares the header's): the
        // scalar cache starts every launch cold, and the row loads of the obstacle passes are issued two rows
        // at a time -- each of them would otherwise be a miss of its own, one memory round trip after the other.
        // Four independent loads, unconditional: a packed blob is never shorter than AQUA_BLOB_MIN_BYTES.
        const uint32_t __attribute__((address_space(4)))* w = (const uint32_t __attribute__((address_space(4)))*)(uintptr_t)a.obst_blob;
        if (QUICK == QUICK_ALWAYS || (QUICK == QUICK_IF_PRESENT && a.K <= QUICK_MAX)) {
            // The first look reads the quick table, not the rows: its first two groups go out with the header (one
            // scalar-memory round trip for everything, in the shadow of the state loads and the draws) and are the
            // lines worth having; the rows are left to the second look and the float64 path.
            k.quick = (QuickPtr)(uintptr_t)(reinterpret_cast<const char*>(a.obst_blob) + quick_offset(a.K));
            k.qc0 = quick_circles(k.quick, QUICK_C0);
            k.qr0 = quick_rects(k.quick, QUICK_R0);
        }
        // The rows' lines are touched in every case: the second look and the float64 path walk the rows, and although
        // only ~5 % of the launches have a wavefront that goes there, its cold misses then add to the launch
        // (measured without the touches: float64 path 0.15 us per launch on average instead of 0.08).
        k.touch[0] = w[16]; k.touch[1] = w[32]; k.touch[2] = w[48]; k.touch[3] = w[64];
    }
    k.obst64 = reinterpret_cast<const double*>(reinterpret_cast<const char*>(a.obst_blob) + sizeof(ObstHeader) +
                                               sizeof(ObstF) * a.K);
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
        static_assert(ACT_W_TURN == 2.0f * ACT_H_TURN && ACT_W_LINE == 2.0f * ACT_H_LINE, "w = 2 h, exactly");
        m.w = m.h + m.h;
        m.chord = idx == 2 ? ACT_C_LINE : ACT_C_TURN;
    }
    return m;
}

template <int AK>
__device__ __forceinline__ ExactMotion exact_motion(const Motion& m)
{
    if constexpr (AK == AQUA_ACT_F32X2 || AK == AQUA_ACT_SAMPLE_C) {
        const double vl = fmin(fmax(static_cast<double>(m.vl), 0.2), 0.5);      // aqua.py:145-150
        const double vr = fmin(fmax(static_cast<double>(m.vr), 0.2), 0.5);
        return exact_motion_continuous(vl, vr);
    } else {
        return exact_motion_discrete(m.idx);                                     // aqua.py:33-42
    }
}

__device__ __forceinline__ int sample_discrete(uint32_t r) { return static_cast<int>((static_cast<uint64_t>(r >> 8) * 3u) >> 24); }
__device__ __forceinline__ float sample_thrust(uint32_t r) { return fmaf(0.3f, u_01(r), 0.2f); }

// Diagnostic build only (-DAQUA_STAMPS=1, tools/stamps.py): s_memtime stamps per wavefront at phase
// boundaries, written to a buffer no product code reads.  Never defined in the shipped library.
#ifndef AQUA_STAMPS
#define AQUA_STAMPS 0
#endif

#if AQUA_STAMPS
__device__ unsigned long long* g_stamps = nullptr;
#define AQUA_STAMP_IMPL(slot, insn)                                                                   \
    do {                                                                                              \
        __builtin_amdgcn_sched_barrier(0);                                                            \
        unsigned long long t_;                                                                        \
        asm volatile(insn " %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_));                                  \
        __builtin_amdgcn_sched_barrier(0);                                                            \
        if (g_stamps != nullptr && (threadIdx.x & 63) == 0)                                           \
            g_stamps[(static_cast<size_t>(blockIdx.x) * (blockDim.x / 64) + threadIdx.x / 64) * 8 + (slot)] = t_; \
    } while (0)
// AQUA_STAMPS == 2: only the wavefront-start (3) and wavefront-end (2) wall-clock stamps, to perturb less
// AQUA_STAMPS == 3: those two, kept per launch for the last 32 launches (profiles/r03/scripts/burst_timeline.py)
#if AQUA_STAMPS == 3
template <typename A> __device__ __forceinline__ uint64_t launch_tick(const A& a);
#define AQUA_STAMP(slot) do { } while (0)
#define AQUA_RTSTAMP(slot)                                                                            \
    do {                                                                                              \
        if ((slot) == 2 || (slot) == 3) {                                                             \
            __builtin_amdgcn_sched_barrier(0);                                                        \
            unsigned long long t_;                                                                    \
            asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_));                      \
            __builtin_amdgcn_sched_barrier(0);                                                        \
            if (g_stamps != nullptr && (threadIdx.x & 63) == 0)                                       \
                g_stamps[(((launch_tick(a) & 31u) * gridDim.x + blockIdx.x) * (blockDim.x / 64) + threadIdx.x / 64) * 2 + ((slot) == 2)] = t_; \
        }                                                                                             \
    } while (0)
#elif AQUA_STAMPS == 2
#define AQUA_STAMP(slot) do { } while (0)
#define AQUA_RTSTAMP(slot) do { if ((slot) == 2 || (slot) == 3) AQUA_STAMP_IMPL(slot, "s_memrealtime"); } while (0)
#else
#define AQUA_STAMP(slot) AQUA_STAMP_IMPL(slot, "s_memtime")
// wall-clock stamp (s_memrealtime, 100 MHz, common to all XCDs)
#define AQUA_RTSTAMP(slot) AQUA_STAMP_IMPL(slot, "s_memrealtime")
#endif
#else
#define AQUA_STAMP(slot) do { } while (0)
#define AQUA_RTSTAMP(slot) do { } while (0)
#endif

// Philox draws of the lane's worlds (pairs share a call: see STREAM_STEP in aqua_device.hpp)
template <int VEC, bool SCALAR_KEY = true>
__device__ __forceinline__ void pair_draws(uint64_t seed, uint64_t env0, uint64_t tick, uint32_t stream,
                                           uint32_t (&w0)[VEC], uint32_t (&w1)[VEC])
{
    static_assert(VEC == 1, "one world per lane");
    const bool odd = (env0 & 1u) != 0;
    // lanes 2i and 2i + 1 hold the two worlds of one pair whenever the wavefront's first world is even
    // (always, unless the caller's env_offset is odd): the pair then computes its block together
    if (uni((static_cast<uint32_t>(env0) ^ threadIdx.x) & 1u) == 0u) {
        draw_pair(seed, env0 >> 1, tick, stream, 0, odd, w0[0], w1[0]);
        return;
    }
    uint32_t r[4];
    draw<SCALAR_KEY>(seed, env0 >> 1, tick, stream, 0, r);
    w0[0] = odd ? r[2] : r[0];
    w1[0] = odd ? r[3] : r[1];
}

// Optional fused epilogue: the observation as main/impl/utils.py:15-33 (AquaStateNormalizer) hands it to the DQN --
// obs / (high - low) with 0.5 added to the angle: x/100, y/100, theta/(2 pi) + 0.5, gx/100, gy/100.
__device__ __forceinline__ void write_norm(const StepArgs& a, int64_t i, float x, float y, float th, float gx, float gy,
                                           bool wb = false)
{
    if (a.obs_norm == nullptr) return;
    float* const o = a.obs_norm + i;
    st1(o + 0 * a.ld, x * 0.01f, wb);
    st1(o + 1 * a.ld, y * 0.01f, wb);
    st1(o + 2 * a.ld, fmaf(th, 0.15915494309189535f, 0.5f), wb);
    st1(o + 3 * a.ld, gx * 0.01f, wb);
    st1(o + 4 * a.ld, gy * 0.01f, wb);
}

// Scheduling fence between the Philox draws and the first use of anything loaded: the draws (~280 cycles,
// no memory operand) then always run in the shadow of the loads instead of after a wait for them.
template <int VEC, int AK>
__device__ __forceinline__ void hold_loads(float (&x)[VEC], float (&y)[VEC], float (&th)[VEC], float (&gx)[VEC],
                                           float (&gy)[VEC], float (&wx)[VEC], float (&wy)[VEC], int64_t (&araw)[VEC],
                                           float (&avl)[VEC], float (&avr)[VEC])
{
#pragma unroll
    for (int j = 0; j < VEC; ++j) {
        asm volatile("" : "+v"(x[j]), "+v"(y[j]), "+v"(th[j]), "+v"(gx[j]), "+v"(gy[j]), "+v"(wx[j]), "+v"(wy[j]));
        if constexpr (AK == AQUA_ACT_U8 || AK == AQUA_ACT_I32) {
            int lo = static_cast<int>(araw[j]);
            asm volatile("" : "+v"(lo));
            araw[j] = lo;
        } else if constexpr (AK == AQUA_ACT_I64) {
            asm volatile("" : "+v"(araw[j]));
        } else if constexpr (AK == AQUA_ACT_F32X2) {
            asm volatile("" : "+v"(avl[j]), "+v"(avr[j]));
        }
    }
}

template <int VEC, int AK>
__device__ __forceinline__ void fold_actions(const int64_t (&araw)[VEC], int (&aidx)[VEC])
{
#pragma unroll
    for (int j = 0; j < VEC; ++j) {
        if constexpr (AK == AQUA_ACT_U8) aidx[j] = araw[j] > 2 ? 2 : static_cast<int>(araw[j]);
        else if constexpr (AK == AQUA_ACT_I32 || AK == AQUA_ACT_I64) aidx[j] = fold_index(araw[j]);
    }
}

// tick of this launch.  The device-resident base is written only by tick_kernel, between launches, so it
// is read through the scalar path (constant address space) like the obstacle table.
template <typename A>
__device__ __forceinline__ uint64_t launch_tick(const A& a)
{
    if (a.tick_base == nullptr) return a.tick;
    return a.tick + *(const uint64_t __attribute__((address_space(4)))*)(uintptr_t)a.tick_base;
}

// Everything a step kernel's prologue needs from the kernel-argument segment, fetched in ONE batch ahead of the first
// wait.  Left to itself the compiler sinks some of these loads into the blocks that use them, and each of those is a
// scalar-memory round trip (0.3-0.5 us) in front of the state loads.
__device__ __forceinline__ void fetch_args(const StepArgs& a)
{
    asm volatile("" ::"s"(a.state), "s"(a.ld), "s"(a.time), "s"(a.action), "s"(a.N), "s"(a.seed), "s"(a.tick),
                 "s"(a.env_offset), "s"(a.tick_base), "s"(a.obst_blob), "s"(a.noise), "s"(a.reseed_blocks));
}

// A captured rollout of T >= 2 steps advances its device-resident tick base by itself, without a kernel of its own
// (a one-thread kernel behind the steps costs the graph 3.9 us per replay: its launch and a dependent load + store):
// the FIRST step launch copies the base to a scratch word, the LAST one takes its base from that copy and stores
// base + T where the next replay's launches read it.  No launch reads a word another thread of the same launch writes
// (the launches of a stream are serialised), so the scalar-path reads of launch_tick() stay coherent.  Called at the top
// of a kernel's first block only, by one thread.  Its arguments are read from the kernel-argument segment right here,
// through a laundered pointer: as ordinary uses of `a` the compiler hoists their loads into every block's prologue
// and, with the SGPR file full, spills them to VGPR lanes there (StepArgs is the kernels' first parameter: offset 0).
// A kernel argument read where it is used, through a pointer into the kernel-argument segment that the optimiser cannot
// trace back to the kernel's parameter: the value occupies SGPRs inside the branch that wants it and nowhere else.
template <typename T>
__device__ __forceinline__ T kernarg_at(size_t off)
{
    using KernArg = const char __attribute__((address_space(4)))*;
    KernArg kp = (KernArg)__builtin_amdgcn_kernarg_segment_ptr();
    asm volatile("" : "+s"(kp));
    return *(const T __attribute__((address_space(4)))*)(kp + off);
}

// (A is deduced from the kernel's own parameter, never defaulted: a kernel that took NsArgs and read StepArgs' offsets
// would dereference a pointer from the wrong word of its argument segment)
template <typename A>
__device__ __forceinline__ void tick_housekeeping(const A&)
{
    static_assert(std::is_standard_layout<A>::value, "offsetof() into the kernel-argument segment needs a standard-layout struct");
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        uint64_t* const copy_to = kernarg_at<uint64_t*>(offsetof(A, tick_copy_to));
        uint64_t* const bump_to = kernarg_at<uint64_t*>(offsetof(A, tick_bump_to));
        if (copy_to != nullptr || bump_to != nullptr) {
            const uint64_t base = *kernarg_at<const uint64_t*>(offsetof(A, tick_base));
            if (copy_to != nullptr) st1(copy_to, base);
            if (bump_to != nullptr) st1(bump_to, base + kernarg_at<uint64_t>(offsetof(A, tick_bump)));
        }
    }
}

// ------------------------------------------------------------------ one launch per step, next-step restart
// auto_reset == 2 ("next-step", the Gymnasium >= 1.0 convention): a world that finishes at tick t keeps its
// terminal state, is marked pending (negative time) and is restarted DURING tick t + 1, when it does not step:
// that tick reports its fresh observation with reward 0 and term 0.  Nothing of the restart is then on the
// step's dependency chain, and only ~2 % of the worlds restart in a step, so the launch is split by role:
//
//   blocks [0, R)      re-seed.  Block b reads the time row of worlds [b * NS_SCAN, (b + 1) * NS_SCAN), compacts the
//                      pending ones into an LDS list (one private segment per wavefront, one barrier) and
//                      re-seeds them NS_RESEED_GROUP lanes per world.  First in the grid: they start first.
//   blocks [R, R + M)  step.  NS_TILE worlds each, one per lane, straight-line: loads, Philox in the loads'
//                      shadow, fast path, stores.  No LDS, no barrier.
//
// The two roles share nothing inside a launch -- no barrier, no flag, and no assumption about which runs
// first.  What keeps that race-free are the markers in the time row, which carry the parity of the tick that
// wrote them:
//     done_code(t)    = -1 - (t & 1)   the world finished at tick t and waits for its restart
//     restart_code(t) = -3 - (t & 1)   the world was restarted during tick t (its time is 0 from tick t + 1 on)
// At tick T the re-seeding blocks restart exactly the worlds marked done_code(T - 1) and mark them
// restart_code(T); the stepping blocks step the worlds with time >= 0 or restart_code(T - 1) (time 0) and
// leave every other marker alone (reward 0, term 0, nothing written).  Whether a stepping wavefront sees
// done_code(T - 1) or the restart_code(T) that replaced it, it skips the world; a done_code(T) written by a
// stepping wavefront of this launch is not what the scan is looking for.  (Ticks therefore advance by one
// per step; after a jump in parity a marked world simply waits one more step.)
// measured alternatives: 6-wavefront blocks 6.7 us, 8: 6.45, 4: 5.99, 3: 6.3, 2: 7.1; 1024 worlds
// per re-seeding block 5.99, 512: 6.16, 2048: 6.9; 8 lanes per restarting world 5.99, 4: 7.3, 16: 7.4
constexpr int NS_MAIN_WAVES = 4;
constexpr int NS_TILE = NS_MAIN_WAVES * 64, NS_BLOCK = NS_TILE;
constexpr int NS_SCAN_ROWS = 4, NS_SCAN = NS_SCAN_ROWS * NS_BLOCK;      // worlds per re-seeding block
constexpr int NS_RESEED_GROUP = 8;       // (4 / 8 / 16 lanes per restarting world: 6.95 / 4.72 / 5.54 us; 16.7 M worlds: within the noise)
// Late loads.  A launch that is ONE round of blocks has two phases -- every wavefront waits for its nine rows (1.6 us of
// fabric traffic), then every wavefront computes -- and neither overlaps the other.  Three wavefronts of four therefore
// issue their loads BEHIND their Philox draws instead of ahead of them: their requests reach the memory system a few
// hundred nanoseconds later, and the arithmetic of the first wavefronts runs while those are served.  Per step at 262 144
// worlds (profiles/r04/stagger/): 4.89 -> 4.71 us; half of the wavefronts 4.74, one of four 4.82, all of them 4.84; any
// s_sleep in front of the late loads loses (4.98-5.28).
#ifndef AQUA_NS_LATE_LOADS
#define AQUA_NS_LATE_LOADS 1            // (0: a timing build without them, tools/ab.py)
#endif
constexpr bool NS_LATE_LOADS = AQUA_NS_LATE_LOADS != 0;
static_assert(NS_SCAN <= 65536, "list entries are 16-bit offsets");
static_assert(NS_SCAN % NS_TILE == 0, "a re-seeding block covers whole stepping tiles");
// batches of at least this many worlds interleave the two roles through the grid.  Measured per step, interleaved vs
// head-of-grid (profiles/r02/ab_interleave.txt): 16.7 M 215 vs 288 us, 8.4 M 98 vs 106, 6.3 M 90 vs 86, 4.2 M 63 vs 61,
// 1 M 14.6 vs 14.3, 262 144 5.60 vs 5.55
constexpr int64_t NS_INTERLEAVE_MIN = 1 << 19;

static_assert(STORE_WB_NEXT_STEP_MIN >= NS_INTERLEAVE_MIN, "only the interleaved layout has a write-back kernel");

__device__ __forceinline__ int32_t done_code(uint64_t tick) { return -1 - static_cast<int32_t>(tick & 1u); }
__device__ __forceinline__ int32_t restart_code(uint64_t tick) { return -3 - static_cast<int32_t>(tick & 1u); }

constexpr int NS_TABLE_ROWS = 8;   // tables of up to this many obstacles are staged in LDS for the re-seeding pass (0: never)
// launch_step() / launch_step_ns_range() pick the SMALL_TABLE kernels by K <= NS_TABLE_ROWS, and those read the quick table
// (QUICK_ALWAYS, RESEED_QUICK), which aqua_pack_obstacles() writes only for K <= QUICK_MAX
static_assert(NS_TABLE_ROWS <= QUICK_MAX, "the SMALL_TABLE kernels read a quick table that exists only for K <= QUICK_MAX");
struct NsReseedShared {
    ObstF rows[NS_TABLE_ROWS > 0 ? NS_TABLE_ROWS : 1];
    uint32_t count[NS_MAIN_WAVES];
    uint16_t list[NS_MAIN_WAVES][NS_SCAN_ROWS * 64];
};

// the normalised observation of one world (see write_norm): the five row pointers are fetched here, by the launches that
// have such a buffer
__device__ __forceinline__ void ns_write_norm(const NsArgs& a, uint32_t byte_off, float x, float y, float th, float gx,
                                              float gy, bool wb)
{
    if (!(a.flags & NS_HAS_NORM)) return;
    const auto row = [](int r) { return kernarg_at<float*>(offsetof(NsArgs, norm) + sizeof(float*) * r); };
    st_at(row(0), byte_off, x * 0.01f, wb);
    st_at(row(1), byte_off, y * 0.01f, wb);
    st_at(row(2), byte_off, fmaf(th, 0.15915494309189535f, 0.5f), wb);
    st_at(row(3), byte_off, gx * 0.01f, wb);
    st_at(row(4), byte_off, gy * 0.01f, wb);
}

// SMALL_TABLE: the launch has at most NS_TABLE_ROWS obstacles (decided on the host: one kernel per case keeps the
// code each launch has to fetch short -- the instruction cache starts every launch cold)
template <bool SMALL_TABLE, bool WB>
__device__ __forceinline__ void ns_reseed_block(const NsArgs& a, uint32_t block, NsReseedShared& sh)
{
    __builtin_amdgcn_s_setprio(3);                      // the longest chain of the launch: issue first (priority 0 / 1 / 3:
                                                        // 4.72 each since the late loads, profiles/r04/late_loads/)
    const uint32_t base = block * NS_SCAN;              // the block's first world; every access below is row pointer +
    const uint32_t rem = a.N - base;                    // 32-bit byte offset (> 0)
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const uint32_t last = (rem < NS_SCAN ? rem : static_cast<uint32_t>(NS_SCAN)) - 1u;
    int32_t tw[NS_SCAN_ROWS];
#pragma unroll
    for (int j = 0; j < NS_SCAN_ROWS; ++j) {            // the loads first
        const uint32_t i = static_cast<uint32_t>(j * NS_BLOCK) + threadIdx.x;
        tw[j] = ld_at(a.time, (base + (i < last ? i : last)) * 4u);
    }
    // Small tables (the usual case) are copied into LDS by the first lanes -- one vector load each, in flight with
    // the time row -- and the obstacle pass of the re-seeding reads them from there after the barrier below,
    // four rows per wait (through the scalar path it waits once per two rows, 200 clocks each).
    // (reading small tables as SGPR operands from the quick table instead loses here: 5.14 vs 5.02 us)
    constexpr bool table_in_regs = SMALL_TABLE;
    uint32_t table_word = 0;
    if (table_in_regs && threadIdx.x < NS_TABLE_ROWS * 8 && threadIdx.x < static_cast<uint32_t>(a.K) * 8)
        table_word = ld1(reinterpret_cast<const uint32_t*>(static_cast<const char*>(a.obst_blob) + sizeof(ObstHeader)) + threadIdx.x);
    const StepConst k = make_const<QUICK_NEVER>(a, obstacle_rows(a.obst_blob));
    const uint64_t tick = launch_tick(a);
    const int32_t restart = done_code(tick - 1);
    AQUA_RTSTAMP(0);
    uint32_t n_mine = 0;
#pragma unroll
    for (int j = 0; j < NS_SCAN_ROWS; ++j) {            // wavefront-private compaction: ballot + prefix count
        const uint32_t i = static_cast<uint32_t>(j * NS_BLOCK) + threadIdx.x;
#ifdef AQUA_NS_NOMAIN
        const bool p = i < rem && tw[j] != restart && ((base + i) * 2654435761u + static_cast<uint32_t>(tick) * 40503u) % 54u == 0u;
#else
        const bool p = i < rem && tw[j] == restart;
#endif
        const uint64_t m = __ballot(p);
        if (p) sh.list[wave][n_mine + __builtin_amdgcn_mbcnt_hi(static_cast<uint32_t>(m >> 32),
                                   __builtin_amdgcn_mbcnt_lo(static_cast<uint32_t>(m), 0u))] = static_cast<uint16_t>(i);
        n_mine += static_cast<uint32_t>(__builtin_popcountll(m));
    }
    if (lane == 0) sh.count[wave] = n_mine;
    if (table_in_regs && threadIdx.x < NS_TABLE_ROWS * 8) {
        // absent rows: r2 = -3e38, which no squared distance is ever below (word 4 of a row is r2)
        const bool present = threadIdx.x < static_cast<uint32_t>(a.K) * 8;
        reinterpret_cast<uint32_t*>(sh.rows)[threadIdx.x] = present ? table_word : ((threadIdx.x & 7u) == 4u ? 0xFF61B1E6u : 0u);
    }
    __syncthreads();

    uint32_t first[NS_MAIN_WAVES + 1];
    first[0] = 0;
#pragma unroll
    for (int w = 0; w < NS_MAIN_WAVES; ++w) first[w + 1] = first[w] + sh.count[w];
    const uint32_t n_pending = uni(first[NS_MAIN_WAVES]);
    asm volatile("" ::"s"(k.touch[0]), "s"(k.touch[1]), "s"(k.touch[2]), "s"(k.touch[3]));                   // the table's lines are resident from here on
    AQUA_RTSTAMP(1);
    const int waves = args_waves(a), random_boat = (a.flags & NS_RANDOM_BOAT) != 0, random_goal = (a.flags & NS_RANDOM_GOAL) != 0;
    const uint64_t env_base = static_cast<uint64_t>(a.env_offset) + base;
    constexpr uint32_t PER_WAVE = 64 / NS_RESEED_GROUP, PER_BLOCK = NS_MAIN_WAVES * PER_WAVE;
#ifdef AQUA_NS_ONE_PASS                      // (timing experiment: no wavefront of a re-seeding block makes a second pass; results differ)
    for (uint32_t qb = static_cast<uint32_t>(wave) * PER_WAVE; qb < (n_pending < PER_BLOCK ? n_pending : PER_BLOCK); qb += PER_BLOCK) {
#else
    for (uint32_t qb = static_cast<uint32_t>(wave) * PER_WAVE; qb < n_pending; qb += PER_BLOCK) {
#endif
        const uint32_t q = qb + (lane / NS_RESEED_GROUP);
        const bool active = q < n_pending;
        uint32_t seg = 0;
#pragma unroll
        for (int w = 1; w < NS_MAIN_WAVES; ++w) seg += (active && q >= first[w]) ? 1u : 0u;
        const uint32_t i = sh.list[seg][active ? q - first[seg] : 0];
        const uint64_t env = env_base + i;
        EnvState e;
        if constexpr (SMALL_TABLE)
            e = reset_env_group<NS_RESEED_GROUP, NS_TABLE_ROWS>(active, a.seed, env, tick, waves, random_boat, random_goal, k.K, k.obst, sh.rows);
        else
            e = reset_env_group<NS_RESEED_GROUP>(active, a.seed, env, tick, waves, random_boat, random_goal, k.K, k.obst);
        if (active && (lane & (NS_RESEED_GROUP - 1)) == 0) {
            constexpr bool wb = WB;
            const uint32_t i4 = (base + i) * 4u;
            st_at(a.row[0], i4, e.x, wb); st_at(a.row[1], i4, e.y, wb); st_at(a.row[2], i4, e.th, wb);
            st_at(a.row[3], i4, e.gx, wb); st_at(a.row[4], i4, e.gy, wb);
            st_at(a.row[5], i4, e.wx, wb); st_at(a.row[6], i4, e.wy, wb);
            st_at(a.time, i4, restart_code(tick), wb);
            ns_write_norm(a, i4, e.x, e.y, e.th, e.gx, e.gy, wb);
        }
    }
    AQUA_RTSTAMP(2);
}

// Five blocks of a CU (one re-seeding, four stepping at 262 144 worlds) must be resident TOGETHER: the grid is exactly
// one round of blocks, and a kernel that needs more than 512 / 5 registers per lane leaves some blocks waiting for a
// slot (a second round: +0.5 us per launch, measured with 82 registers -> 88 allocated -> 5 wavefronts per SIMD and
// not one to spare).  Asking for at least 6 wavefronts per SIMD caps the kernel at 80 registers.
// Which role, which tile (next-step kernels).  A grid of ONE round of blocks (262 144 worlds fill the 256 CUs exactly
// once) starts its re-seeding blocks first -- theirs is the longest chain.  A grid of many rounds would run thousands of
// re-seeding blocks (latency-bound, three wavefronts of four busy, no memory traffic) before the first world is stepped,
// and the two roles would no longer overlap: there one block in every five re-seeds, next to the four stepping blocks of
// the same 1024 worlds (16.7 M worlds: 288 -> 195-215 us per step).  Returns false for the spare
// blocks of a rounded-up grid.  role_index: the re-seeding block's number, or the stepping block's tile number.
template <bool INTERLEAVE>
__device__ __forceinline__ bool ns_role(uint32_t reseed_blocks, uint32_t n_tiles, bool& reseed_role, uint32_t& role_index)
{
    if constexpr (INTERLEAVE) {
        // Group g = the re-seeding block and the four stepping blocks of worlds [1024 g, 1024 g + 1024).  Blocks are dealt
        // round-robin over the 8 XCDs (blocks b and b + 8 share one: observed, relied on for speed only), so the five
        // members of a group are blocks x + 8 (5 q + m), m = 0..4, of group 8 q + x: one XCD, one L2 -- the time row the
        // re-seeding block scans is the one its stepping neighbours read.
        constexpr uint32_t PER = NS_SCAN / NS_TILE + 1;
        const uint32_t x = blockIdx.x & 7u, j = blockIdx.x >> 3;
        const uint32_t q = j / PER, member = j - q * PER;
        const uint32_t group = q * 8u + x;
        reseed_role = member == 0;
        role_index = reseed_role ? group : group * (PER - 1) + (member - 1);
        if (group >= reseed_blocks) return false;              // the grid is rounded up to whole sets of 8 groups
        if (!reseed_role && role_index >= n_tiles) return false;   // the last group may be short of stepping tiles
    } else {
        reseed_role = blockIdx.x < reseed_blocks;      // (at the tail of the grid instead: 4.72 -> 4.75 us)
        role_index = reseed_role ? blockIdx.x : blockIdx.x - reseed_blocks;
    }
    return true;
}

// INTERLEAVE: the grid layout (chosen on the host by batch size, like SMALL_TABLE: one kernel per case keeps the SGPR
// file of the one-round case -- the benchmarked one -- exactly as tight as it was)
template <int AK, bool SMALL_TABLE, bool INTERLEAVE, bool WB = false>
__global__ __launch_bounds__(NS_BLOCK) __attribute__((amdgpu_waves_per_eu(6, 8))) void step_ns_kernel(const NsArgs a)
{
    __shared__ NsReseedShared sh;
    AQUA_RTSTAMP(3);        // wavefront started
    tick_housekeeping(a);
    bool reseed_role;
    uint32_t role_index;
    if (!ns_role<INTERLEAVE>(a.reseed_blocks, (a.N + NS_TILE - 1) / NS_TILE, reseed_role, role_index)) return;
    if (reseed_role) {
#ifndef AQUA_NS_NOWORK                       // (timing experiment: what the re-seeding blocks cost the launch)
        ns_reseed_block<SMALL_TABLE, WB>(a, role_index, sh);
#endif
        return;
    }
#ifdef AQUA_NS_NOMAIN
    return;                                  // (timing experiment: the re-seeding blocks alone)
#endif
    // ---- stepping block.  Every load goes out first: an address is a row pointer straight from the kernel arguments
    // plus the lane's byte offset (tile included), while the table header and the tick base below cost another
    // scalar-memory round trip.  One straight-line sequence for whole and ragged tiles (a lane past the end reads the
    // tile's last world and never writes).
    const uint32_t tile = role_index * NS_TILE;
    const uint32_t rem = a.N - tile;                   // >= 1
    const int lane = threadIdx.x & 63;
    const uint32_t last = (rem < NS_TILE ? rem : static_cast<uint32_t>(NS_TILE)) - 1u;
    const uint32_t off = threadIdx.x;                  // < NS_TILE
    const bool valid = off < rem;
    const uint32_t o = tile + (off < last ? off : last);       // the lane's world: a 32-bit offset (saddr + voffset accesses)
    float x[1], y[1], th[1], gx[1], gy[1], wx[1], wy[1], u0[1] = {0.0f}, u1[1] = {0.0f}, avl[1] = {0.5f}, avr[1] = {0.5f};
    int32_t tin[1];
    int64_t araw[1] = {2};
    int aidx[1] = {2};
    const uint32_t o4 = o * 4u;                         // byte offset inside a float / int row
    // late loads (NS_LATE_LOADS above): wavefronts 1-3 of the block issue theirs behind the draws -- in the one-round layout
    // only (the interleaved grids of large batches are staggered by themselves: 524 288 worlds 8.30 -> 8.67 us with them)
    const bool late = NS_LATE_LOADS && !INTERLEAVE && (threadIdx.x >> 6) != 0u;
    const auto issue_loads = [&]() {
        tin[0] = ld_at(a.time, o4);
        x[0] = ld_at(a.row[0], o4); y[0] = ld_at(a.row[1], o4); th[0] = ld_at(a.row[2], o4);
        gx[0] = ld_at(a.row[3], o4); gy[0] = ld_at(a.row[4], o4);
        wx[0] = ld_at(a.row[5], o4); wy[0] = ld_at(a.row[6], o4);
        if constexpr (AK == AQUA_ACT_U8) araw[0] = ld_at(static_cast<const uint8_t*>(a.action), o);
        else if constexpr (AK == AQUA_ACT_I32) araw[0] = ld_at(static_cast<const int32_t*>(a.action), o4);
        else if constexpr (AK == AQUA_ACT_I64) araw[0] = ld_at(static_cast<const int64_t*>(a.action), o * 8u);
        else if constexpr (AK == AQUA_ACT_F32X2) {
            avl[0] = ld_at(static_cast<const float*>(a.action), o4);
            avr[0] = ld_at(a.action_hi, o4);
        }
    };
    if (!late) issue_loads();
    constexpr int QUICK = SMALL_TABLE ? QUICK_ALWAYS : QUICK_NEVER;     // tables of up to 8 rows: the quick table (5.22 -> 5.08 us)
    const StepConst k = make_const<QUICK>(a, obstacle_rows(a.obst_blob));
    const uint64_t tick = launch_tick(a);
    AQUA_RTSTAMP(0);
    // The draws need no loaded value: they run in the shadow of the loads.
    const uint64_t env0 = (static_cast<uint64_t>(a.env_offset) + tile) + off;
    if (__builtin_expect(!(a.flags & NS_HAS_NOISE), 1)) {
        uint32_t w0[1], w1[1];
        pair_draws<1>(a.seed, env0, tick, STREAM_STEP, w0, w1);
        u0[0] = u_pm1(w0[0]); u1[0] = u_pm1(w1[0]);
    } else {                                           // injected noise (tests): loaded here, not up front, so that no
        u0[0] = ld_at(kernarg_at<const float*>(offsetof(NsArgs, noise)), o4);             // load of this rare path is
        u1[0] = ld_at(kernarg_at<const float*>(offsetof(NsArgs, noise) + sizeof(float*)), o4);   // outstanding across the draws
        asm volatile("" : "+v"(u0[0]), "+v"(u1[0]));   // waited for right here: see above
    }
    if constexpr (AK >= AQUA_ACT_SAMPLE_D) {
        uint32_t w0[1], w1[1];
        pair_draws<1, false>(a.seed, env0, tick, STREAM_ACT, w0, w1);
        if constexpr (AK == AQUA_ACT_SAMPLE_D) aidx[0] = sample_discrete(w0[0]);
        else { avl[0] = sample_thrust(w0[0]); avr[0] = sample_thrust(w1[0]); }
    }
    asm volatile("" : "+v"(u0[0]), "+v"(u1[0]));
    if (late) issue_loads();
    asm volatile("" : "+v"(tin[0]), "+v"(u0[0]), "+v"(u1[0]));   // scheduling fence, see hold_loads()
    hold_loads<1, AK>(x, y, th, gx, gy, wx, wy, araw, avl, avr);
    AQUA_RTSTAMP(4);        // draws done, loads back
    fold_actions<1, AK>(araw, aidx);
    const int32_t t0 = tin[0] == restart_code(tick - 1) ? 0 : tin[0];   // restarted last tick: steps from 0
#ifdef AQUA_NS_NOWORK
    const bool pending = false;                        // (timing experiment: nobody restarts, every world keeps stepping)
#else
    const bool pending = valid && t0 < 0;              // any other marker: the world does not step
#endif
    if constexpr (AK == AQUA_ACT_BEARING) aidx[0] = bearing_action(x[0], y[0], th[0], gx[0], gy[0]);
    const float x0 = x[0], y0 = y[0], th0 = th[0], wx0 = wx[0], wy0 = wy[0];
    EnvState e{x[0], y[0], th[0], gx[0], gy[0], wx[0], wy[0], t0};
    const Motion mo = decode_motion<AK>(k, aidx[0], avl[0], avr[0]);
    float rew;
    uint32_t code;
    const bool live = valid && !pending;
    AQUA_RTSTAMP(1);
    asm volatile("" ::"s"(k.touch[0]), "s"(k.touch[1]), "s"(k.touch[2]), "s"(k.touch[3]));                   // the table's lines are resident from here on
    const bool knife = fast_step<false, QUICK>(e, mo.h, mo.w, mo.chord, u0[0], u1[0], k, rew, code) && live;
    if (__builtin_expect(any_lane(knife), 0)) {
        if (knife) {
            const ExactOut o2 = exact_step(x0, y0, th0, gx[0], gy[0], wx0, wy0, e.t, exact_motion<AK>(mo), k.K, k.obst64,
                                           k.obst, k.band2, k.time_limit);
            e.x = o2.x; e.y = o2.y; e.th = o2.th; rew = o2.reward; code = o2.term;
        }
    }
    if (!live) { rew = 0.0f; code = 0u; }              // a restarting (or padding) world reports reward 0, term 0
    const bool done = code != 0u;
    // The stores' own copies of the lane offset.  With the loads' o4 the row addresses are common subexpressions of
    // the loads' and are kept, as 64-bit VGPR pairs formed by v_lshl_add_u64, from the top of the block to here; a
    // value the optimiser cannot match to it leaves base + offset to be formed where it is used, and there it is the
    // store's own SGPR base + 32-bit VGPR offset addressing: no instruction at all.
    uint32_t s4 = o4, s1 = o;
    asm volatile("" : "+v"(s4), "+v"(s1));
    constexpr bool wb = WB;                             // batches of millions of worlds only (STORE_WB_*)
    if (valid) {
        st_at(a.reward, s4, rew, wb);
        st_at(a.term, s1, static_cast<uint8_t>(code), wb);
    }
    if (a.done_bits != nullptr) {
        const uint64_t b = __ballot(done);
        const uint32_t word = (tile >> 6) + (uni(off) >> 6);                    // the wavefront's word: scalar
        if (word < ((a.N + 63u) >> 6) && lane == 0) {
            if (a.flags & NS_DONE_WORD_WB) a.done_bits[word] = b;               // (store_done_word(): the policy is the host's)
            else st1(a.done_bits + word, b);
        }
    }
    if (live) {                                        // pending worlds are written by the re-seeding blocks
        st_at(a.row[0], s4, e.x, wb); st_at(a.row[1], s4, e.y, wb); st_at(a.row[2], s4, e.th, wb);
        st_at(a.row[5], s4, e.wx, wb); st_at(a.row[6], s4, e.wy, wb);
#ifdef AQUA_NS_NOWORK
        st_at(a.time, s4, e.t, wb);
#else
        st_at(a.time, s4, done ? done_code(tick) : e.t, wb);
#endif
        ns_write_norm(a, s4, e.x, e.y, e.th, gx[0], gy[0], wb);
    }
    AQUA_RTSTAMP(2);
}

// ------------------------------------------------------------------ T steps in one launch
// State stays in registers for the whole rollout; per world-step only the action is read and reward/term
// are written.  Finished worlds are re-seeded once per step for the whole 256-world block: every wavefront
// publishes its lanes that need it on an LDS list, ONE wavefront (a different one every step) re-seeds them eight
// lanes per world (reset_env_group) and leaves the fresh states in LDS, the owners pick them up -- two
// barriers per step.  (Re-seeding inside each wavefront costs a full pass of ~500 instructions in three of
// four wavefronts per step for one or two worlds each; per block it is one pass for about five.)  In the
// next-step mode the other wavefronts step their worlds between the two barriers.  Both restart conventions
// give exactly the per-step kernels' results.
struct RolloutShared {
    uint32_t count[2][BLOCK_SMALL / 64];
    uint8_t list[2][BLOCK_SMALL / 64][64];
    float result[BLOCK_SMALL][8];
};

struct ReseedTicket {       // what publish_reseed() hands to collect_reseed()
    uint32_t n, slot;       // worlds of the block to re-seed (block-uniform); this lane's slot when it is one of them
};

__device__ __forceinline__ ReseedTicket publish_reseed(bool need, RolloutShared& sh, int parity)
{
    constexpr int WAVES = BLOCK_SMALL / 64;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const uint64_t m = __ballot(need);
    const uint32_t pos = __builtin_amdgcn_mbcnt_hi(static_cast<uint32_t>(m >> 32), __builtin_amdgcn_mbcnt_lo(static_cast<uint32_t>(m), 0u));
    if (need) sh.list[parity][wave][pos] = static_cast<uint8_t>(threadIdx.x);
    if (lane == 0) sh.count[parity][wave] = static_cast<uint32_t>(__builtin_popcountll(m));
    __syncthreads();
    uint32_t first = 0, total = 0;
#pragma unroll
    for (int w = 0; w < WAVES; ++w) {
        const uint32_t c = sh.count[parity][w];
        if (w < wave) first += c;
        total += c;
    }
    return ReseedTicket{uni(total), first + pos};
}

// the duty wavefront re-seeds the published worlds (tick: the tick whose draws the restart uses)
template <bool SMALL, typename A>
__device__ __forceinline__ void serve_reseed(const ReseedTicket& tk, const A& a, const StepConst& k, uint64_t tick,
                                             int64_t block_first_world, RolloutShared& sh, int parity)
{
    constexpr int WAVES = BLOCK_SMALL / 64;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    // a different wavefront every tick AND for every block: wavefront i of every block of a CU runs on SIMD i, and a duty
    // that depended on the tick alone put the re-seeding passes of all of a CU's blocks on one SIMD (fused rollout, next-step
    // restarts: 4.38 -> 4.05 us per step, profiles/r04/fused_duty/)
    if (tk.n == 0 || wave != static_cast<int>((static_cast<uint32_t>(tick) + blockIdx.x) & (WAVES - 1))) return;
#ifdef AQUA_FUSED_NOSERVE                    // (timing experiment: the protocol without the re-seeding pass)
    return;
#endif
    uint32_t first[WAVES + 1];
    first[0] = 0;
#pragma unroll
    for (int w = 0; w < WAVES; ++w) first[w + 1] = first[w] + sh.count[parity][w];
    constexpr uint32_t PER_PASS = 64 / RESET_GROUP;
    for (uint32_t qb = 0; qb < tk.n; qb += PER_PASS) {
        const uint32_t q = qb + (lane / RESET_GROUP);
        const bool active = q < tk.n;
        uint32_t seg = 0;
#pragma unroll
        for (int w = 1; w < WAVES; ++w) seg += (active && q >= first[w]) ? 1u : 0u;
        const uint32_t owner = sh.list[parity][seg][active ? q - first[seg] : 0];
        const uint64_t world = static_cast<uint64_t>(a.env_offset + block_first_world) + owner;
        EnvState f;
        if constexpr (SMALL)
            f = reset_env_group<RESET_GROUP, RESEED_QUICK>(active, a.seed, world, tick, k.waves, args_random_boat(a), args_random_goal(a),
                                                        k.K, k.obst, nullptr, k.quick, k.Kc);
        else
            f = reset_env_group<RESET_GROUP>(active, a.seed, world, tick, k.waves, args_random_boat(a), args_random_goal(a), k.K, k.obst);
        if (active && (lane & (RESET_GROUP - 1)) == 0) {
            float* r = sh.result[q];
            r[0] = f.x; r[1] = f.y; r[2] = f.th; r[3] = f.gx; r[4] = f.gy; r[5] = f.wx; r[6] = f.wy;
        }
    }
}

__device__ __forceinline__ void collect_reseed(EnvState& e, bool need, const ReseedTicket& tk, RolloutShared& sh)
{
    if (tk.n == 0) return;                               // block-uniform
    __syncthreads();
    if (need) {
        const float* r = sh.result[tk.slot];
        e.x = r[0]; e.y = r[1]; e.th = r[2]; e.gx = r[3]; e.gy = r[4]; e.wx = r[5]; e.wy = r[6]; e.t = 0;
    }
}

// The next-step fused rollout's hand-over (round 5): NO barrier, and a wavefront of its own that only re-seeds.  A world that
// finishes at step s does not move during step s + 1 (that tick reports reward 0 / term 0), and what it restarts with is a
// function of (seed, world, tick s + 1) alone.  So its lane posts it the moment `done` is known, at the END of step s, and
// collects the fresh state right before it moves again, in step s + 2; in between -- ONE step of slack: `done` is not
// known earlier, the state is due one step later -- a fifth wavefront of the block, which owns no worlds, runs the
// re-seeding pass (~0.9 us of dependent arithmetic) beside the other four.  The wavefronts meet only through sequence
// numbers in LDS.  Post p of a tile (p = 0: the worlds that came into the launch marked done; p = s + 1: the worlds that
// finished at step s), global sequence q = q0 + p:
//   post     every stepping wavefront lists its lanes that need a fresh state (list / count, double-buffered by the parity
//            of q) and sets posted[wave] = q + 1; before it overwrites the lists of q - 2 it makes sure q - 2 was served.
//   serve    the re-seeding wavefront waits for posted[*] > q, re-seeds the listed worlds with the draws of tick0 + p (eight
//            lanes per world) into result[parity][wave * 64 + position] and sets served = q + 1.
//   collect  a wavefront with lanes on list q waits for served > q right before those lanes move again.
// History: with two block-wide barriers per step and the pass on a rotating duty wavefront (the same-step mode's protocol
// below, and this mode's until round 4) three wavefronts of four idle through every pass: 3.98 us per step where the steps
// alone take 1.3; this protocol in the library 3.85 (3.65 with sampled actions).  Every wait is bounded: one that sees no progress for MAIL_SPIN_LIMIT polls gives up (the
// results are then wrong, every test compares them, and the grid drains).
#ifndef AQUA_FUSED_MAIL
#define AQUA_FUSED_MAIL 1                                   // (0: the barrier protocol for the next-step mode too -- A/B timing)
#endif
#ifndef AQUA_MAIL_SPIN_LIMIT
#define AQUA_MAIL_SPIN_LIMIT (1u << 18)                    // polls of ~0.1 us; a wait that is answered takes a few
#endif
constexpr uint32_t MAIL_SPIN_LIMIT = AQUA_MAIL_SPIN_LIMIT;
// the re-seeding wavefront runs at raised priority: its pass is one long dependent chain, and the four stepping wavefronts it
// shares a SIMD with would otherwise leave it a fifth of the issue slots (262 144 worlds, next-step restarts, us per step in
// builds with ONE action kind: priority 0 4.22, 1 3.52, 2 3.48-3.58, 3 3.50-3.52; the barrier protocol 4.01-4.03 --
// profiles/r05/fused_mail/; the library's own figures are 0.3 higher across the board)
#ifndef AQUA_MAIL_PRIO
#define AQUA_MAIL_PRIO 2
#endif
struct RolloutMail {
    uint32_t posted[BLOCK_SMALL / 64];
    uint32_t served;
    uint32_t count[2][BLOCK_SMALL / 64];
    uint8_t list[2][BLOCK_SMALL / 64][64];
    float result[2][BLOCK_SMALL][8];
};

// The flags and what they guard all live in LDS, which serves one wavefront's instructions in the order they were issued:
// data first, flag second on the writing side; flag first, data second on the reading side.  Plain accesses with compiler
// barriers are therefore enough -- and they must be all there is: an acquire / release at workgroup scope compiles to
// s_waitcnt vmcnt(0), which parks the wavefront behind its own reward / term stores to HBM at every step (measured in round
// 4: the protocol alone 1.1 us per step instead of 0.3).  ds instructions by hand: a volatile access through a generic
// pointer becomes a FLAT load, with vmcnt(0) as well.
__device__ __forceinline__ uint32_t lds_offset(const void* p)
{
    return static_cast<uint32_t>(reinterpret_cast<uintptr_t>((const __attribute__((address_space(3))) void*)p));
}
__device__ __forceinline__ void mail_wait(const uint32_t* flag, uint32_t at_least)
{
    const uint32_t at = lds_offset(flag);
    for (uint32_t polls = 0; polls < MAIL_SPIN_LIMIT; ++polls) {
        uint32_t v;
        asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(at) : "memory");
        if (uni(v) >= at_least) break;
        __builtin_amdgcn_s_sleep(1);
    }
}
__device__ __forceinline__ void mail_flag(uint32_t* flag, uint32_t value)
{
    asm volatile("s_waitcnt lgkmcnt(0)\n\tds_write_b32 %0, %1" ::"v"(lds_offset(flag)), "v"(value) : "memory");
}
// -> this lane's position on its wavefront's list (meaningful when `need`)
__device__ __forceinline__ uint32_t mail_post(bool need, RolloutMail& sh, uint32_t q)
{
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, parity = static_cast<int>(q & 1u);
    if (q >= 2u) mail_wait(&sh.served, q - 1u);          // the lists of q - 2 have been read
    const uint64_t m = __ballot(need);
    const uint32_t pos = __builtin_amdgcn_mbcnt_hi(static_cast<uint32_t>(m >> 32), __builtin_amdgcn_mbcnt_lo(static_cast<uint32_t>(m), 0u));
    if (need) sh.list[parity][wave][pos] = static_cast<uint8_t>(lane);
    if (lane == 0) {
        sh.count[parity][wave] = static_cast<uint32_t>(__builtin_popcountll(m));
        mail_flag(&sh.posted[wave], q + 1u);
    }
    return pos;
}

template <bool SMALL, typename A>
__device__ __forceinline__ void mail_serve(const A& a, const StepConst& k, uint64_t tick, int64_t block_first_world, RolloutMail& sh, uint32_t q)
{
    constexpr int WAVES = BLOCK_SMALL / 64;
    const int lane = threadIdx.x & 63, parity = static_cast<int>(q & 1u);
    uint32_t first[WAVES + 1];
    first[0] = 0;
#pragma unroll
    for (int w = 0; w < WAVES; ++w) {
        mail_wait(&sh.posted[w], q + 1u);
        first[w + 1] = first[w] + sh.count[parity][w];
    }
#ifdef AQUA_FUSED_NOSERVE                    // (timing experiment: the protocol without the re-seeding pass)
    const uint32_t n = 0;
#else
    const uint32_t n = uni(first[WAVES]);
#endif
    constexpr uint32_t PER_PASS = 64 / RESET_GROUP;
    for (uint32_t qb = 0; qb < n; qb += PER_PASS) {
        const uint32_t i = qb + (lane / RESET_GROUP);
        const bool active = i < n;
        uint32_t seg = 0;
#pragma unroll
        for (int w = 1; w < WAVES; ++w) seg += (active && i >= first[w]) ? 1u : 0u;
        const uint32_t at = active ? i - first[seg] : 0u;
        const uint32_t owner = seg * 64u + sh.list[parity][seg][at];
        const uint64_t world = static_cast<uint64_t>(a.env_offset + block_first_world) + owner;
        EnvState f;
        if constexpr (SMALL)
            f = reset_env_group<RESET_GROUP, RESEED_QUICK>(active, a.seed, world, tick, k.waves, args_random_boat(a), args_random_goal(a),
                                                        k.K, k.obst, nullptr, k.quick, k.Kc);
        else
            f = reset_env_group<RESET_GROUP>(active, a.seed, world, tick, k.waves, args_random_boat(a), args_random_goal(a), k.K, k.obst);
        if (active && (lane & (RESET_GROUP - 1)) == 0) {
            float* r = sh.result[parity][seg * 64u + at];
            r[0] = f.x; r[1] = f.y; r[2] = f.th; r[3] = f.gx; r[4] = f.gy; r[5] = f.wx; r[6] = f.wy;
        }
    }
    if (lane == 0) mail_flag(&sh.served, q + 1u);
}

// the fresh states of post q for the lanes that were on its lists (`pos`: what mail_post returned)
__device__ __forceinline__ void mail_collect(EnvState& e, bool need, uint32_t pos, RolloutMail& sh, uint32_t q)
{
    if (!any_lane(need)) return;
    mail_wait(&sh.served, q + 1u);
    if (need) {
        const float* r = sh.result[q & 1u][(threadIdx.x & ~63u) + pos];
        e.x = r[0]; e.y = r[1]; e.th = r[2]; e.gx = r[3]; e.gy = r[4]; e.wx = r[5]; e.wy = r[6];
    }
}

// ------------------------------------------------------------------ one launch per step: no restart, or restart in the same launch
// Workgroup = TILE_WORLDS lanes, tile = TILE_WORLDS consecutive worlds, one per lane; the per-lane path is the stepping
// role of step_ns_kernel without the restart markers (same arguments: NsArgs, same addressing; no late loads).
// RESTART == false (auto_reset 0, the reference's own step(), which never restarts a world): no list, no barrier, no
// re-seeding code in the kernel.  RESTART == true (auto_reset 1): worlds that finish are not re-seeded by their own lane
// (that would be 1-2 active lanes per wavefront, in every wavefront, looping over rejection attempts): their tile-local
// indices go on a list in LDS (one private segment per wavefront: ballot + prefix count) and, after ONE barrier, groups of
// RESET_GROUP lanes re-seed them densely (reset_env_group, draws of this tick) and write the fresh state straight to
// HBM; the owning lane skips its state stores for that world.  Tiles of 1024 worlds: 512 / 256 are slower here (6.46 /
// 6.66 / 7.00 us per step, profiles/r03/ab_tile_worlds.txt), and so are 256-world tiles whose restarted worlds are handed
// back to their own lanes through LDS and stored with the tile's coalesced row stores (the fused rollout's protocol:
// bit-identical, 7.5 us at 262 144 worlds, and behind this kernel at every size up to 16.7 M: profiles/r04/same_step_tile/)
// -- the lanes' own stores out ahead of the barrier and the re-seeding from scalar operands are worth more than the
// scattered words they cost.
struct TileShared {
    uint32_t count[TILE_WORLDS / 64];
    uint16_t list[TILE_WORLDS / 64][64];
};

template <int AK, bool SMALL_TABLE, bool RESTART, bool WB = false>
__global__ __launch_bounds__(TILE_WORLDS) void step_kernel(const NsArgs a)
{
    tick_housekeeping(a);
    const uint32_t tile = blockIdx.x * TILE_WORLDS;
    const uint32_t rem = a.N - tile;                   // >= 1
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint32_t last = (rem < TILE_WORLDS ? rem : static_cast<uint32_t>(TILE_WORLDS)) - 1u;
    const uint32_t off = threadIdx.x;
    const bool valid = off < rem;
    const uint32_t o = tile + (off < last ? off : last);
    float x[1], y[1], th[1], gx[1], gy[1], wx[1], wy[1], u0[1] = {0.0f}, u1[1] = {0.0f}, avl[1] = {0.5f}, avr[1] = {0.5f};
    int32_t tin[1];
    int64_t araw[1] = {2};
    int aidx[1] = {2};
    const uint32_t o4 = o * 4u;
    // every wavefront issues its loads first (step_ns_kernel's late loads lose here: same-step 5.86 -> 6.01 us per step, no
    // restart 3.56 -> 3.69: sixteen wavefronts of a block already reach the memory system spread out)
    tin[0] = ld_at(a.time, o4);
    x[0] = ld_at(a.row[0], o4); y[0] = ld_at(a.row[1], o4); th[0] = ld_at(a.row[2], o4);
    gx[0] = ld_at(a.row[3], o4); gy[0] = ld_at(a.row[4], o4);
    wx[0] = ld_at(a.row[5], o4); wy[0] = ld_at(a.row[6], o4);
    if constexpr (AK == AQUA_ACT_U8) araw[0] = ld_at(static_cast<const uint8_t*>(a.action), o);
    else if constexpr (AK == AQUA_ACT_I32) araw[0] = ld_at(static_cast<const int32_t*>(a.action), o4);
    else if constexpr (AK == AQUA_ACT_I64) araw[0] = ld_at(static_cast<const int64_t*>(a.action), o * 8u);
    else if constexpr (AK == AQUA_ACT_F32X2) {
        avl[0] = ld_at(static_cast<const float*>(a.action), o4);
        avr[0] = ld_at(a.action_hi, o4);
    }
    constexpr int QUICK = SMALL_TABLE ? QUICK_ALWAYS : QUICK_NEVER;
    const StepConst k = make_const<QUICK>(a, obstacle_rows(a.obst_blob));
    const uint64_t tick = launch_tick(a);
    const uint64_t env0 = (static_cast<uint64_t>(a.env_offset) + tile) + off;
    if (__builtin_expect(!(a.flags & NS_HAS_NOISE), 1)) {
        uint32_t w0[1], w1[1];
        pair_draws<1>(a.seed, env0, tick, STREAM_STEP, w0, w1);
        u0[0] = u_pm1(w0[0]); u1[0] = u_pm1(w1[0]);
    } else {
        u0[0] = ld_at(kernarg_at<const float*>(offsetof(NsArgs, noise)), o4);
        u1[0] = ld_at(kernarg_at<const float*>(offsetof(NsArgs, noise) + sizeof(float*)), o4);
        asm volatile("" : "+v"(u0[0]), "+v"(u1[0]));
    }
    if constexpr (AK >= AQUA_ACT_SAMPLE_D) {
        uint32_t w0[1], w1[1];
        pair_draws<1, false>(a.seed, env0, tick, STREAM_ACT, w0, w1);
        if constexpr (AK == AQUA_ACT_SAMPLE_D) aidx[0] = sample_discrete(w0[0]);
        else { avl[0] = sample_thrust(w0[0]); avr[0] = sample_thrust(w1[0]); }
    }
    asm volatile("" : "+v"(tin[0]), "+v"(u0[0]), "+v"(u1[0]));   // scheduling fence, see hold_loads()
    hold_loads<1, AK>(x, y, th, gx, gy, wx, wy, araw, avl, avr);
    fold_actions<1, AK>(araw, aidx);
    if constexpr (AK == AQUA_ACT_BEARING) aidx[0] = bearing_action(x[0], y[0], th[0], gx[0], gy[0]);
    const float x0 = x[0], y0 = y[0], th0 = th[0], wx0 = wx[0], wy0 = wy[0];
    EnvState e{x[0], y[0], th[0], gx[0], gy[0], wx[0], wy[0], tin[0]};
    const Motion mo = decode_motion<AK>(k, aidx[0], avl[0], avr[0]);
    float rew;
    uint32_t code;
    asm volatile("" ::"s"(k.touch[0]), "s"(k.touch[1]), "s"(k.touch[2]), "s"(k.touch[3]));                   // the table's lines are resident from here on
    const bool knife = fast_step<false, QUICK>(e, mo.h, mo.w, mo.chord, u0[0], u1[0], k, rew, code) && valid;
    if (__builtin_expect(any_lane(knife), 0)) {
        if (knife) {
            const ExactOut o2 = exact_step(x0, y0, th0, gx[0], gy[0], wx0, wy0, e.t, exact_motion<AK>(mo), k.K, k.obst64,
                                           k.obst, k.band2, k.time_limit);
            e.x = o2.x; e.y = o2.y; e.th = o2.th; rew = o2.reward; code = o2.term;
        }
    }
    const bool done = valid && code != 0u;
    uint32_t s4 = o4, s1 = o;                           // the stores' own copies of the lane offset (see step_ns_kernel)
    asm volatile("" : "+v"(s4), "+v"(s1));
    constexpr bool wb = WB;
    if (valid) {
        st_at(a.reward, s4, rew, wb);
        st_at(a.term, s1, static_cast<uint8_t>(code), wb);
    }
    const uint64_t done_ballot = __ballot(done);
    if (a.done_bits != nullptr) {
        const uint32_t word = (tile >> 6) + (uni(off) >> 6);
        if (word < ((a.N + 63u) >> 6) && lane == 0) {
            if (a.flags & NS_DONE_WORD_WB) a.done_bits[word] = done_ballot;
            else st1(a.done_bits + word, done_ballot);
        }
    }
    // The lane's own state stores go out AHEAD of the barrier: nothing behind it needs them, and the wavefronts that
    // re-seed end with the restart stores only.  A finished world's fresh state is written by its group below.
    if (valid && !(RESTART && done)) {
        st_at(a.row[0], s4, e.x, wb); st_at(a.row[1], s4, e.y, wb); st_at(a.row[2], s4, e.th, wb);
        st_at(a.row[5], s4, e.wx, wb); st_at(a.row[6], s4, e.wy, wb);
        st_at(a.time, s4, e.t, wb);
        ns_write_norm(a, s4, e.x, e.y, e.th, gx[0], gy[0], wb);
    }
    if constexpr (RESTART) {
        __shared__ TileShared sh;
        constexpr int WAVES = TILE_WORLDS / 64;
        if (done) sh.list[wave][__builtin_amdgcn_mbcnt_hi(static_cast<uint32_t>(done_ballot >> 32),
                                __builtin_amdgcn_mbcnt_lo(static_cast<uint32_t>(done_ballot), 0u))] = static_cast<uint16_t>(off);
        if (lane == 0) sh.count[wave] = static_cast<uint32_t>(__builtin_popcountll(done_ballot));
        __syncthreads();
        uint32_t first[WAVES + 1];
        first[0] = 0;
#pragma unroll
        for (int w = 0; w < WAVES; ++w) first[w + 1] = first[w] + sh.count[w];
        const uint32_t n_done = uni(first[WAVES]);
        const int waves = args_waves(a), random_boat = args_random_boat(a), random_goal = args_random_goal(a);
        const uint64_t env_base = static_cast<uint64_t>(a.env_offset) + tile;
        constexpr uint32_t GROUPS = TILE_WORLDS / RESET_GROUP;
        for (uint32_t qb = static_cast<uint32_t>(wave) * (64 / RESET_GROUP); qb < n_done; qb += GROUPS) {
            const uint32_t q = qb + (lane / RESET_GROUP);
            const bool active = q < n_done;
            uint32_t seg = 0;
#pragma unroll
            for (int w = 1; w < WAVES; ++w) seg += (active && q >= first[w]) ? 1u : 0u;
            const uint32_t i = sh.list[seg][active ? q - first[seg] : 0];
            EnvState f;
            if constexpr (SMALL_TABLE)
                f = reset_env_group<RESET_GROUP, RESEED_QUICK>(active, a.seed, env_base + i, tick, waves, random_boat, random_goal,
                                                            k.K, k.obst, nullptr, k.quick, k.Kc);
            else
                f = reset_env_group<RESET_GROUP>(active, a.seed, env_base + i, tick, waves, random_boat, random_goal, k.K, k.obst);
            if (active && (lane & (RESET_GROUP - 1)) == 0) {
                const uint32_t i4 = (tile + i) * 4u;
                st_at(a.row[0], i4, f.x, wb); st_at(a.row[1], i4, f.y, wb); st_at(a.row[2], i4, f.th, wb);
                st_at(a.row[3], i4, f.gx, wb); st_at(a.row[4], i4, f.gy, wb);
                st_at(a.row[5], i4, f.wx, wb); st_at(a.row[6], i4, f.wy, wb);
                st_at(a.time, i4, f.t, wb);
                ns_write_norm(a, i4, f.x, f.y, f.th, f.gx, f.gy, wb);
            }
        }
    }
}

// SMALL: the table has a quick table (host-selected, as for step_kernel): one inlined copy of the obstacle look and of
// the re-seeding instead of two behind run-time branches
// MODE: a.auto_reset, likewise host-selected: one inlined copy of the re-seeding (or none) in the loop
// the next-step mode's blocks carry a fifth wavefront that owns no worlds and only re-seeds (mail_serve)
constexpr bool rollout_mail(int mode) { return AQUA_FUSED_MAIL && mode == AQUA_RESET_NEXT_STEP; }
constexpr int rollout_threads(int mode) { return rollout_mail(mode) ? BLOCK_SMALL + 64 : BLOCK_SMALL; }
template <int AK, bool SMALL, int MODE>
__global__ __launch_bounds__(rollout_threads(MODE), rollout_threads(MODE) / 64) void rollout_kernel(const StepArgs a)   // four blocks per CU
{
    constexpr bool MAIL = rollout_mail(MODE);
    __shared__ typename std::conditional<MAIL, RolloutMail, RolloutShared>::type sh;
    if constexpr (MAIL) {
        if (threadIdx.x < BLOCK_SMALL / 64) sh.posted[threadIdx.x] = 0u;
        if (threadIdx.x == 0) sh.served = 0u;
        __syncthreads();
    }
    constexpr int QUICK = SMALL ? QUICK_ALWAYS : QUICK_NEVER;
    const uint64_t tick0 = launch_tick(a);
    const int64_t N = a.N, ld = a.ld;
    uint32_t q0 = 0;                                       // posts of this block's earlier tiles (the mailbox's sequence)
    if constexpr (MAIL) {
        if (threadIdx.x >= BLOCK_SMALL) {                  // the re-seeding wavefront follows the others through the same tiles
            // its own constants, made on its side of the split: the stepping wavefronts keep the quick table's first groups
            // in 32 SGPRs for the whole rollout, this wavefront reads the groups as it goes -- made ahead of the split, the
            // two paths' scalar registers added up, and how many of them the compiler then spilled inside the step loop
            // depended on what ELSE was in the translation unit (the library 119 lane moves per step against 72 for the
            // same source built with one action kind: 4.08 against 3.60 us per step, profiles/r05/fused_mail/fused_time_d.txt)
            StepConst k = make_const<QUICK_NEVER>(a, obstacle_rows(a.obst_blob));
            if constexpr (SMALL) k.quick = (QuickPtr)(uintptr_t)(reinterpret_cast<const char*>(a.obst_blob) + quick_offset(a.K));
            if (AQUA_MAIL_PRIO) __builtin_amdgcn_s_setprio(AQUA_MAIL_PRIO);
            for (int64_t bbase = static_cast<int64_t>(blockIdx.x) * BLOCK_SMALL; bbase < N;
                 bbase += static_cast<int64_t>(gridDim.x) * BLOCK_SMALL, q0 += static_cast<uint32_t>(a.T))
                for (int64_t p = 0; p < a.T; ++p)          // post p: the restarts of tick0 + p
                    mail_serve<SMALL>(a, k, tick0 + static_cast<uint64_t>(p), bbase, sh, q0 + static_cast<uint32_t>(p));
            return;
        }
    }
#ifdef AQUA_FUSED_QUICK_RESIDENT              // (A/B: the quick table's first groups held in SGPRs for the whole rollout)
    const StepConst k = make_const<SMALL ? QUICK_IF_PRESENT : QUICK_NEVER>(a, obstacle_rows(a.obst_blob));
#else
    // The quick table's first two groups are NOT kept in scalar registers across the steps (they are in the one-launch-per-
    // step kernels, where they are loaded once and used once): 32 SGPRs held through a loop whose body needs ~100 were
    // spilled to VGPR lanes and read back lane by lane in front of every obstacle test -- how many, the compiler decided
    // differently from build to build of the same source.  Each step re-reads the groups from the scalar cache instead.
    StepConst k = make_const<QUICK_NEVER>(a, obstacle_rows(a.obst_blob));
    if constexpr (SMALL) k.quick = (QuickPtr)(uintptr_t)(reinterpret_cast<const char*>(a.obst_blob) + quick_offset(a.K));
#endif
    // whole blocks iterate together; lanes past N are inert
    for (int64_t bbase = static_cast<int64_t>(blockIdx.x) * BLOCK_SMALL; bbase < N;
         bbase += static_cast<int64_t>(gridDim.x) * BLOCK_SMALL, q0 += static_cast<uint32_t>(a.T)) {
        const int64_t i = bbase + threadIdx.x;
        const bool valid = i < N;
        const int64_t ic = valid ? i : N - 1;
        EnvState e{a.state[0 * ld + ic], a.state[1 * ld + ic], a.state[2 * ld + ic], a.state[3 * ld + ic],
                   a.state[4 * ld + ic], a.state[5 * ld + ic], a.state[6 * ld + ic], a.time[ic]};
        const uint64_t env = static_cast<uint64_t>(a.env_offset + i);        // the lane's own index (pairs of lanes draw together)
        // mailbox: `owed` = this lane was on the lists of post (s - 1) -- it did not move during step s - 1 and collects its
        // fresh state in step s; `posted_pos` = its position on the lists of post s (the worlds that restart DURING step s)
        bool owed = false;
        uint32_t owed_pos = 0, posted_pos = 0;
        if constexpr (MAIL) posted_pos = mail_post(valid && e.t == done_code(tick0 - 1), sh, q0);      // post 0: marked on entry
        for (int64_t s = 0; s < a.T; ++s) {
            const uint64_t tick = tick0 + static_cast<uint64_t>(s);
            const int parity = static_cast<int>(s & 1);
            // next-step restart, with the markers of step_ns_kernel: a world that finished last tick does not
            // move this tick, it is re-seeded instead; any other marker waits
            bool pending = false, restart = false;
            ReseedTicket tk{0u, 0u};
            if constexpr (MODE == AQUA_RESET_NEXT_STEP) {
                if (e.t == restart_code(tick - 1)) e.t = 0;
                restart = valid && e.t == done_code(tick - 1);
                pending = valid && e.t < 0;
                if constexpr (!MAIL) {
                    tk = publish_reseed(restart, sh, parity);
                    serve_reseed<SMALL>(tk, a, k, tick, bbase, sh, parity);     // one wavefront; the others go on stepping
                }
            }
            // (the next step's action fetched one step ahead was measured: next-step 3.61 = 3.64, no restart 1.39 -> 1.51 us per
            // step, profiles/r05/fused_mail/fused_time_d.txt: not used)
            int idx = 2;
            float vl = 0.5f, vr = 0.5f;
            if constexpr (AK == AQUA_ACT_U8) idx = fold_index(static_cast<const uint8_t*>(a.action)[s * a.action_step_stride + ic]);
            if constexpr (AK == AQUA_ACT_I32) idx = fold_index(static_cast<const int32_t*>(a.action)[s * a.action_step_stride + ic]);
            if constexpr (AK == AQUA_ACT_I64) idx = fold_index(static_cast<const int64_t*>(a.action)[s * a.action_step_stride + ic]);
            if constexpr (AK == AQUA_ACT_F32X2) {
                const float* base = static_cast<const float*>(a.action) + s * a.action_step_stride;
                vl = base[ic]; vr = base[a.action_ld + ic];
            }
            // (Philox's ten round keys are loop invariants: the compiler hoists them out of the step loop, holds twenty more
            // scalar registers through it, spills them to VGPR lanes and reads them back with a v_readlane + s_nop in front of
            // every round.  Laundering the seed per step -- the keys re-made by scalar additions, 25 % fewer lane moves in the
            // loop -- was measured with all action kinds built: no restart 1.27 -> 1.32 us per step, same-step 3.79 -> 3.88,
            // next-step 3.87 = 3.87; per-world fused no restart 1.57 -> 1.75.  Lane moves are not the cost; not used.)
            uint64_t seed = a.seed;
#ifdef AQUA_FUSED_KEYS_LAUNDERED              // (A/B timing)
            asm volatile("" : "+s"(seed));
#endif
            uint32_t w0[1], w1[1];
            pair_draws<1, false>(seed, env, tick, STREAM_STEP, w0, w1);
            const float u0 = u_pm1(w0[0]), u1 = u_pm1(w1[0]);
            if constexpr (AK >= AQUA_ACT_SAMPLE_D) {
                pair_draws<1, false>(seed, env, tick, STREAM_ACT, w0, w1);
                if constexpr (AK == AQUA_ACT_SAMPLE_D) idx = sample_discrete(w0[0]);
                else { vl = sample_thrust(w0[0]); vr = sample_thrust(w1[0]); }
            }
            // the worlds re-seeded during the LAST step move again in this one: their fresh states are due here
            if constexpr (MAIL) {
                if (s > 0) mail_collect(e, owed, owed_pos, sh, q0 + static_cast<uint32_t>(s) - 1u);
            }
            if constexpr (AK == AQUA_ACT_BEARING) idx = bearing_action(e.x, e.y, e.th, e.gx, e.gy);
            const Motion m = decode_motion<AK>(k, idx, vl, vr);
            const EnvState before = e;
            EnvState after = e;
            float rew;
            uint32_t code;
#ifndef AQUA_FUSED_QUICK_RESIDENT
            if constexpr (SMALL) {
                QuickPtr q = k.quick;
                asm volatile("" : "+s"(q));                 // (laundered: the loads stay inside the step)
                k.qc0 = quick_circles(q, QUICK_C0);
                k.qr0 = quick_rects(q, QUICK_R0);
            }
#endif
            const bool knife = fast_step<false, QUICK>(after, m.h, m.w, m.chord, u0, u1, k, rew, code) && valid && !pending;
            if (__builtin_expect(any_lane(knife), 0)) {
                if (knife) {
                    const ExactOut o = exact_step(before.x, before.y, before.th, before.gx, before.gy, before.wx,
                                                  before.wy, after.t, exact_motion<AK>(m), k.K, k.obst64, k.obst, k.band2,
                                                  k.time_limit);
                    after.x = o.x; after.y = o.y; after.th = o.th; rew = o.reward; code = o.term;
                }
            }
            if (pending) { rew = 0.0f; code = 0u; }          // the restart tick reports reward 0, term 0
            else e = after;
            if (valid) {
                a.reward[s * a.out_step_stride + i] = rew;
                a.term[s * a.out_step_stride + i] = static_cast<uint8_t>(code);
            }
            const bool done = valid && code != 0u;
            if constexpr (MAIL) {
                if (restart) e.t = restart_code(tick);
                else if (done) e.t = done_code(tick);
                owed = restart; owed_pos = posted_pos;
                // the worlds that restart during step s + 1 (those that just finished; by the marker, which is what step
                // s + 1 will go by): posted NOW, two steps ahead of their next move
                if (s + 1 < a.T) posted_pos = mail_post(valid && e.t == done_code(tick), sh, q0 + static_cast<uint32_t>(s) + 1u);
            } else if constexpr (MODE == AQUA_RESET_NEXT_STEP) {
                collect_reseed(e, restart, tk, sh);
                if (restart) e.t = restart_code(tick);
                else if (done) e.t = done_code(tick);
            } else if constexpr (MODE == AQUA_RESET_SAME_STEP) {
                const ReseedTicket t1 = publish_reseed(done, sh, parity);
                serve_reseed<SMALL>(t1, a, k, tick, bbase, sh, parity);
                collect_reseed(e, done, t1, sh);
            }
        }
        if constexpr (MAIL) mail_collect(e, owed, owed_pos, sh, q0 + static_cast<uint32_t>(a.T) - 1u);      // the last step's restarts
        if (valid) {
            a.state[0 * ld + i] = e.x; a.state[1 * ld + i] = e.y; a.state[2 * ld + i] = e.th;
            a.state[3 * ld + i] = e.gx; a.state[4 * ld + i] = e.gy;
            a.state[5 * ld + i] = e.wx; a.state[6 * ld + i] = e.wy;
            a.time[i] = e.t;
        }
    }
}

// ------------------------------------------------------------------ per-world obstacle tables
// One launch per batched step, no restart (the caller resets with reset_tables_kernel): one world per lane, every
// lane reads the rows of its own table (WorldTable, struct of arrays over the worlds: coalesced), generic box
// formula for every row.  The arithmetic, the bands and the float64 path are the shared-table kernels'.
// RESTART (auto_reset 1, same-step): worlds that finish are published on an LDS list (one private segment per
// wavefront) and, after ONE barrier, re-seeded eight lanes per world against their own tables (draws of this tick,
// like step_kernel); the owning lane keeps its reward / term stores and skips its state stores.
constexpr int RESET_SCAN_ROWS = 4, RESET_SCAN = RESET_SCAN_ROWS * BLOCK_SMALL, RESET_DENSE = RESET_SCAN / 8;
struct ResetShared {
    uint32_t count[BLOCK_SMALL / 64];
    uint16_t list[BLOCK_SMALL / 64][RESET_SCAN_ROWS * 64];
};
struct TablesShared {
    uint32_t count[BLOCK_SMALL / 64];
    uint8_t list[BLOCK_SMALL / 64][64];
};
// Hand-off of a restarting world's table rows from its own lane (which loaded them with its state) to the eight lanes
// that re-seed it: HANDOFF_PER_WAVE slots per wavefront and round (1.2 worlds of a wavefront restart in a step; a wavefront
// with more takes another round, block-uniform).  The re-seeding then reads NO memory: with the rows fetched from the
// table a second time -- a dependent round trip behind the step's traffic -- the same-step launch took 13.6-14.4 us and
// the role-split next-step launch 14.4-15.0 (its re-seeding blocks need the rows of the worlds their scan finds).
constexpr int HANDOFF_PER_WAVE = 8;
template <int KREG>
struct HandoffShared {
    uint32_t count[BLOCK_SMALL / 64];
    uint8_t world[BLOCK_SMALL / 64][HANDOFF_PER_WAVE];           // the slot's world (offset in the tile)
    ObstF rows[BLOCK_SMALL / 64][HANDOFF_PER_WAVE][KREG];
};
// The same hand-off for tables too long for registers (9, 10 and 17..64 rows; next-step restart): a world marked "finished
// last tick" does not step this tick, but its lane still rides along through the wavefront's coalesced row loads -- it
// leaves every row in an LDS slot as the loop streams it (fast_step<.., SINK>), and after ONE barrier eight lanes re-seed
// the world from the slot (reset_env_group<.., RESEED_LDS5>).  What this replaces: the launch split by role, whose
// re-seeding groups fetched the rows of the worlds their scan found from memory -- one float per 128-byte line, 6 K lines
// per world, ~5 000 worlds per step: at 64 rows a quarter of a gigabyte of line traffic on top of the 419 MB the step
// itself streams (262 144 worlds: 98 us per step against 68 without restarts; 32 rows: 48 against 34).
// SINK_SLOTS slots per wavefront and round (1.2 worlds of a wavefront restart in a step); a wavefront with more takes
// another round, block-uniform, in which the lanes still waiting read their own rows once more.
#ifndef AQUA_SINK_SLOTS
#define AQUA_SINK_SLOTS 3
#endif
constexpr int SINK_SLOTS = AQUA_SINK_SLOTS;
// groups of eight lanes per restarting world (tables_step_block's SINK_SPLIT; 262 144 worlds, next-step, us per step with 1 /
// 2 / 4: 9 rows 16.5 / 16.6 / 17.3, 17 rows 24.7 / 24.7 / 24.5, 32 rows 40.2 / 38.8 / 37.5, 64 rows 81.2 / 76.5 / 75.6)
constexpr int SINK_SPLIT_SHORT = 2, SINK_SPLIT_LONG = 4, SINK_SPLIT_LONG_MIN_ROWS = 24;
struct SinkShared {
    uint32_t count[BLOCK_SMALL / 64];
    uint8_t world[BLOCK_SMALL / 64][SINK_SLOTS];                 // the slot's world (offset in the tile)
    float rows[BLOCK_SMALL / 64][SINK_SLOTS][AQUA_MAX_OBSTACLES * 5];
};
struct NoShared {};
constexpr int TABLES_NEXT_STEP_TILE = 3;
// MODE: AQUA_RESET_NONE, AQUA_RESET_SAME_STEP (restart inside the launch, below), AQUA_RESET_NEXT_STEP (the stepping
// role of step_tables_ns_kernel: worlds carrying a restart marker do not step, as in step_ns_kernel),
// TABLES_NEXT_STEP_TILE (next-step restart inside the tile, no re-seeding blocks -- KREG > 0: the lane of a world marked
// "finished last tick" holds that world's rows already and hands them over as the same-step restart does; KREG == 0: it
// leaves them in LDS as the row loop streams them, SinkShared above).
// KREG: tables of at most KREG rows (host-selected; 0: any length).  The lane's rows are then loaded WITH its state --
// uniform row base + the lane's 32-bit offset, forty loads in flight behind the nine of the state, one memory round trip
// -- instead of two rows at a time after the move is known (three dependent round trips for eight rows, and 64-bit
// per-lane addresses that cost the kernel 187 registers: two wavefronts per SIMD).
constexpr int TABLES_KREG = 8, TABLES_KREG_WIDE = 16, TABLES_KREG_WIDE_MIN = 11;      // (HandoffShared<KREG>)
template <int AK, int MODE, int KREG, int SINK_SPLIT = 2>
__device__ __forceinline__ void tables_step_block(const StepArgs& a, const float* __restrict__ t32, const double* __restrict__ t64,
                                                  int64_t tld, float band2, float band2_tight, int64_t tile)
{
    constexpr bool RESTART = MODE == AQUA_RESET_SAME_STEP;
    constexpr bool NS = MODE == AQUA_RESET_NEXT_STEP || MODE == TABLES_NEXT_STEP_TILE;
    constexpr bool SINK = MODE == TABLES_NEXT_STEP_TILE && KREG == 0;      // rows handed over as they are streamed (SinkShared)
    // Same-step restart of such tables: which worlds finish is known only after the rows have gone by, so the wavefront
    // fetches a finished world's rows again into the same slots, lane j row j (REFETCH).  From the struct of arrays that
    // was 5 K cache lines per world -- 2 M extra line requests per step at 64 rows (TCP_TCC_READ_REQ 2.69 M -> 4.66 M,
    // profiles/r05/tables/tables_pmc64.txt), a quarter of a gigabyte, as it was for the groups that read their rows from
    // memory (95 us per step against 61 without restarts) -- from the world-major copy behind it, 24 K contiguous bytes
    // (-DAQUA_TABLES_NO_REFETCH: the groups read the struct of arrays, for the A/B)
    // SINK_SPLIT == 1 (host: tables of 9 and 10 rows): the groups read the struct of arrays, two rows per round trip -- at that
    // length the slots' rounds cost more than the scattered lines (9 rows: 15.9 us per step against 17.4)
#ifndef AQUA_TABLES_NO_REFETCH
    constexpr bool REFETCH = MODE == AQUA_RESET_SAME_STEP && KREG == 0 && SINK_SPLIT > 1;
#else
    constexpr bool REFETCH = false;
#endif
    __shared__ typename std::conditional<SINK || REFETCH, SinkShared, NoShared>::type ssh;
    const int64_t ld = a.ld, rem = a.N - tile;
    const int lane = threadIdx.x & 63;
    float* const row0 = a.state + tile;
    int32_t* const trow = a.time + tile;
    const uint32_t last = static_cast<uint32_t>(rem < BLOCK_SMALL ? rem - 1 : BLOCK_SMALL - 1);
    const uint32_t off = threadIdx.x;
    const bool valid = static_cast<int64_t>(off) < rem;
    const uint32_t o = off < last ? off : last;
    const uint32_t o4 = o * 4u;
    float x[1], y[1], th[1], gx[1], gy[1], wx[1], wy[1], u0[1] = {0.0f}, u1[1] = {0.0f}, avl[1] = {0.5f}, avr[1] = {0.5f};
    int64_t araw[1] = {2};
    int aidx[1] = {2};
    const int32_t tin = ld_at(trow, o4);
    x[0] = ld_at(row0 + 0 * ld, o4); y[0] = ld_at(row0 + 1 * ld, o4); th[0] = ld_at(row0 + 2 * ld, o4);
    gx[0] = ld_at(row0 + 3 * ld, o4); gy[0] = ld_at(row0 + 4 * ld, o4);
    wx[0] = ld_at(row0 + 5 * ld, o4); wy[0] = ld_at(row0 + 6 * ld, o4);
    if constexpr (AK == AQUA_ACT_U8) araw[0] = ld_at(static_cast<const uint8_t*>(a.action) + tile, o);
    else if constexpr (AK == AQUA_ACT_I32) araw[0] = ld_at(static_cast<const int32_t*>(a.action) + tile, o4);
    else if constexpr (AK == AQUA_ACT_I64) araw[0] = ld_at(static_cast<const int64_t*>(a.action) + tile, o * 8u);
    else if constexpr (AK == AQUA_ACT_F32X2) {
        avl[0] = ld_at(static_cast<const float*>(a.action) + tile, o4);
        avr[0] = ld_at(static_cast<const float*>(a.action) + a.action_ld + tile, o4);
    }
    ObstF rows[KREG > 0 ? KREG : 1];
    if constexpr (KREG > 0) {
#pragma unroll
        for (int j = 0; j < KREG; ++j) {
            const int jj = j < a.K ? j : a.K - 1;         // uniform; a.K >= 1
            const float* const rb = t32 + (6 * jj) * tld + tile;
            rows[j].cx = ld_at(rb, o4); rows[j].cy = ld_at(rb + tld, o4); rows[j].hx = ld_at(rb + 2 * tld, o4);
            rows[j].hy = ld_at(rb + 3 * tld, o4); rows[j].r2 = ld_at(rb + 4 * tld, o4);
        }
        // (the same rows staged through an LDS tile [8][5][256] and read back behind a barrier, as north_star sketches the
        // per-world list: 14.30 -> 14.67 and 10.64 -> 11.04 us per step -- every element is used by one lane, once)
    }
    StepConst k;
    k.W = a.W; k.sigma = a.sigma; k.waves = a.waves; k.time_limit = a.time_limit; k.K = a.K; k.Kc = 0;
    k.band2 = band2; k.band2_tight = band2_tight; k.obst = nullptr; k.obst64 = nullptr;
    k.touch[0] = k.touch[1] = k.touch[2] = k.touch[3] = 0;
    const uint64_t tick = launch_tick(a);
    const uint64_t env0 = static_cast<uint64_t>(a.env_offset + tile) + off;
    if (a.noise == nullptr) {
        uint32_t w0[1], w1[1];
        pair_draws<1>(a.seed, env0, tick, STREAM_STEP, w0, w1);
        u0[0] = u_pm1(w0[0]); u1[0] = u_pm1(w1[0]);
    } else {
        u0[0] = ld_at(a.noise + tile, o4);
        u1[0] = ld_at(a.noise + a.noise_ld + tile, o4);
    }
    if constexpr (AK >= AQUA_ACT_SAMPLE_D) {
        uint32_t w0[1], w1[1];
        pair_draws<1, false>(a.seed, env0, tick, STREAM_ACT, w0, w1);
        if constexpr (AK == AQUA_ACT_SAMPLE_D) aidx[0] = sample_discrete(w0[0]);
        else { avl[0] = sample_thrust(w0[0]); avr[0] = sample_thrust(w1[0]); }
    }
    fold_actions<1, AK>(araw, aidx);
    if constexpr (AK == AQUA_ACT_BEARING) aidx[0] = bearing_action(x[0], y[0], th[0], gx[0], gy[0]);
    // next-step restart: restarted last tick -> steps from time 0; any other marker -> the world does not step
    const int32_t t0 = (NS && tin == restart_code(tick - 1)) ? 0 : tin;
    const bool live = valid && !(NS && t0 < 0);
    const float x0 = x[0], y0 = y[0], th0 = th[0], wx0 = wx[0], wy0 = wy[0];
    EnvState e{x[0], y[0], th[0], gx[0], gy[0], wx[0], wy[0], t0};
    const Motion mo = decode_motion<AK>(k, aidx[0], avl[0], avr[0]);
    const WorldTable wt{t32 + tile, t64 + tile, tld, o};
    float rew;
    uint32_t code;
    bool sink_want = false;
    uint32_t sink_mine = 0;
    uint64_t sink_ballot = 0;
    float* sink = nullptr;
    if constexpr (SINK) {
        sink_want = valid && tin == done_code(tick - 1);          // finished last tick: re-seeded during this one
        sink_ballot = __ballot(sink_want);
        sink_mine = __builtin_amdgcn_mbcnt_hi(static_cast<uint32_t>(sink_ballot >> 32),
                                              __builtin_amdgcn_mbcnt_lo(static_cast<uint32_t>(sink_ballot), 0u));
        if (sink_want && sink_mine < SINK_SLOTS) sink = &ssh.rows[threadIdx.x >> 6][sink_mine][0];
    }
    constexpr int INFLIGHT = MODE == AQUA_RESET_NONE ? 2 : AQUA_TABLE_ROWS_IN_FLIGHT_RESTART;       // (see fast_step)
    const bool knife = fast_step<true, QUICK_NEVER, KREG, SINK, INFLIGHT>(e, mo.h, mo.w, mo.chord, u0[0], u1[0], k, rew, code, &wt, rows, sink) && live;
    if (__builtin_expect(any_lane(knife), 0)) {
        if (knife) {
            const ExactOut o2 = exact_step_world(x0, y0, th0, gx[0], gy[0], wx0, wy0, e.t, exact_motion<AK>(mo), k.K, k.band2,
                                                 k.time_limit, wt);
            e.x = o2.x; e.y = o2.y; e.th = o2.th; rew = o2.reward; code = o2.term;
        }
    }
    if (!live) { rew = 0.0f; code = 0u; }                 // a restarting (or padding) world reports reward 0, term 0
    const bool done = code != 0u;
    if (valid) {
        st_at(a.reward + tile, o4, rew);
        st_at(a.term + tile, o, static_cast<uint8_t>(code));
    }
    if (live && !(RESTART && done)) {                     // same-step: a finished world's fresh state is written by its group below
        st_at(row0 + 0 * ld, o4, e.x); st_at(row0 + 1 * ld, o4, e.y); st_at(row0 + 2 * ld, o4, e.th);
        st_at(row0 + 5 * ld, o4, e.wx); st_at(row0 + 6 * ld, o4, e.wy);
        st_at(trow, o4, (NS && done) ? done_code(tick) : e.t);
        write_norm(a, tile + off, e.x, e.y, e.th, gx[0], gy[0]);
    }
    const uint64_t done_ballot = __ballot(done);
    if (a.done_bits != nullptr) {
        const int64_t word = (tile + (threadIdx.x & ~63u)) / 64;
        if (lane == 0 && word < ((a.N + 63) >> 6)) store_done_word(a.done_bits + word, done_ballot, a.N);
    }
    if constexpr ((RESTART || MODE == TABLES_NEXT_STEP_TILE) && KREG > 0) {
        // restart inside the tile, rows handed over through LDS (HandoffShared above)
        __shared__ HandoffShared<KREG> sh;
        constexpr int WAVES = BLOCK_SMALL / 64;
        const int wave = threadIdx.x >> 6;
        const bool want = RESTART ? done : (valid && tin == done_code(tick - 1));     // same-step: finished now; next-step: last tick
        const uint64_t want_ballot = __ballot(want);
        const uint32_t mine = __builtin_amdgcn_mbcnt_hi(static_cast<uint32_t>(want_ballot >> 32),
                                                        __builtin_amdgcn_mbcnt_lo(static_cast<uint32_t>(want_ballot), 0u));
        if (lane == 0) sh.count[wave] = static_cast<uint32_t>(__builtin_popcountll(want_ballot));
        for (uint32_t round = 0;; ++round) {
            const uint32_t lo = round * HANDOFF_PER_WAVE;
            if (want && mine >= lo && mine < lo + HANDOFF_PER_WAVE) {
                const uint32_t slot = mine - lo;
                sh.world[wave][slot] = static_cast<uint8_t>(threadIdx.x);
#pragma unroll
                for (int j = 0; j < KREG; ++j) sh.rows[wave][slot][j] = rows[j];
            }
            __syncthreads();
            uint32_t first[WAVES + 1], most = 0;
            first[0] = 0;
#pragma unroll
            for (int w = 0; w < WAVES; ++w) {
                const uint32_t c = sh.count[w];
                most = c > most ? c : most;
                const uint32_t left = c > lo ? c - lo : 0u;
                first[w + 1] = first[w] + (left < HANDOFF_PER_WAVE ? left : HANDOFF_PER_WAVE);
            }
            const uint32_t n_round = uni(first[WAVES]);
            constexpr uint32_t PER_WAVE = 64 / RESET_GROUP, PER_BLOCK = WAVES * PER_WAVE;
            for (uint32_t qb = static_cast<uint32_t>(wave) * PER_WAVE; qb < n_round; qb += PER_BLOCK) {
                const uint32_t q = qb + (lane / RESET_GROUP);
                const bool active = q < n_round;
                uint32_t seg = 0;
#pragma unroll
                for (int w = 1; w < WAVES; ++w) seg += (active && q >= first[w]) ? 1u : 0u;
                const uint32_t slot = active ? q - first[seg] : 0u;
                const uint32_t i = active ? sh.world[seg][slot] : 0u;                      // an idle group reads a world that exists
                const WorldTable own{t32 + tile, nullptr, tld, i};
                const EnvState f = reset_env_group<RESET_GROUP, KREG == 16 ? RESEED_HANDOFF16 : RESEED_HANDOFF8>(
                    active, a.seed, static_cast<uint64_t>(a.env_offset + tile) + i, tick, a.waves, a.random_boat, a.random_goal, a.K,
                    nullptr, &sh.rows[seg][slot][0], nullptr, 0, &own);
                if (active && (lane & (RESET_GROUP - 1)) == 0) {
                    st1(row0 + 0 * ld + i, f.x); st1(row0 + 1 * ld + i, f.y); st1(row0 + 2 * ld + i, f.th);
                    st1(row0 + 3 * ld + i, f.gx); st1(row0 + 4 * ld + i, f.gy);
                    st1(row0 + 5 * ld + i, f.wx); st1(row0 + 6 * ld + i, f.wy);
                    st1(trow + i, RESTART ? f.t : restart_code(tick));
                    write_norm(a, tile + i, f.x, f.y, f.th, f.gx, f.gy);
                }
            }
            if (uni(most) <= lo + HANDOFF_PER_WAVE) break;        // block-uniform: every wavefront's worlds have had a slot
            __syncthreads();                                      // the slots are written again
        }
    } else if constexpr (SINK || REFETCH) {
        constexpr int WAVES = BLOCK_SMALL / 64;
        const int wave = threadIdx.x >> 6;
        if constexpr (REFETCH) {                                   // the worlds that finished in this step
            sink_want = done;
            sink_ballot = done_ballot;
            sink_mine = __builtin_amdgcn_mbcnt_hi(static_cast<uint32_t>(sink_ballot >> 32),
                                                  __builtin_amdgcn_mbcnt_lo(static_cast<uint32_t>(sink_ballot), 0u));
        }
        if (lane == 0) ssh.count[wave] = static_cast<uint32_t>(__builtin_popcountll(sink_ballot));
        for (uint32_t round = 0;; ++round) {
            const uint32_t lo = round * SINK_SLOTS;
            if (sink_want && sink_mine >= lo && sink_mine < lo + SINK_SLOTS) ssh.world[wave][sink_mine - lo] = static_cast<uint8_t>(threadIdx.x);
            if (round != 0 || REFETCH) {
                // more than SINK_SLOTS worlds of this wavefront restart at once (one wavefront in eight at 64 rows): the worlds
                // still waiting have their rows fetched by the WHOLE wavefront, lane j row j -- 5 loads per lane, all in flight,
                // one memory round trip (left to the owning lane alone it was K dependent round trips at the block's tail)
                static_assert(AQUA_MAX_OBSTACLES <= 64, "one row per lane");
#pragma unroll 1
                for (uint32_t sl = 0; sl < SINK_SLOTS; ++sl) {
                    const uint64_t who = __ballot(sink_want && sink_mine == lo + sl);
                    if (who == 0) break;                                                    // wavefront-uniform
                    const uint32_t owner = (threadIdx.x & ~63u) + static_cast<uint32_t>(__builtin_ctzll(who));
                    if (lane < a.K) {
                        // from the world-major copy behind the struct of arrays (aqua_pack_tables): the world's K rows are
                        // 24 K contiguous bytes -- as 5 K scattered words of the struct of arrays they were 5 K cache lines
                        const float* const r = t32 + static_cast<int64_t>(6) * a.K * tld + ((tile + owner) * a.K + lane) * 6;
                        float* const d = &ssh.rows[wave][sl][5 * lane];
                        d[0] = ld1(r); d[1] = ld1(r + 1); d[2] = ld1(r + 2); d[3] = ld1(r + 3); d[4] = ld1(r + 4);
                    }
                }
            }
            __syncthreads();
            uint32_t first[WAVES + 1], most = 0;
            first[0] = 0;
#pragma unroll
            for (int w = 0; w < WAVES; ++w) {
                const uint32_t c = ssh.count[w];
                most = c > most ? c : most;
                const uint32_t left = c > lo ? c - lo : 0u;
                first[w + 1] = first[w] + (left < SINK_SLOTS ? left : SINK_SLOTS);
            }
            const uint32_t n_round = uni(first[WAVES]);
            // SINK_SPLIT groups of eight lanes per world, each with its share of the rows (reset_env_group): a round holds at
            // most WAVES * SINK_SLOTS = 12 worlds, the block's 256 lanes are 32 groups -- without the split two wavefronts in
            // three sat out the pass that the whole block waits for
            constexpr uint32_t LANES = RESET_GROUP * SINK_SPLIT, PER_WAVE = 64 / LANES, PER_BLOCK = WAVES * PER_WAVE;
            for (uint32_t qb = static_cast<uint32_t>(wave) * PER_WAVE; qb < n_round; qb += PER_BLOCK) {
                const uint32_t q = qb + (lane / LANES);
                const bool active = q < n_round;
                uint32_t seg = 0;
#pragma unroll
                for (int w = 1; w < WAVES; ++w) seg += (active && q >= first[w]) ? 1u : 0u;
                const uint32_t slot = active ? q - first[seg] : 0u;
                const uint32_t i = active ? ssh.world[seg][slot] : 0u;                     // an idle group reads a world that exists
                const WorldTable own{t32 + tile, nullptr, tld, i};
                const EnvState f = reset_env_group<RESET_GROUP, RESEED_LDS5, 8, 256, SINK_SPLIT>(
                    active, a.seed, static_cast<uint64_t>(a.env_offset + tile) + i, tick, a.waves, a.random_boat, a.random_goal, a.K,
                    nullptr, reinterpret_cast<const ObstF*>(&ssh.rows[seg][slot][0]), nullptr, 0, &own);
                if (active && (lane & (LANES - 1)) == 0) {
                    st1(row0 + 0 * ld + i, f.x); st1(row0 + 1 * ld + i, f.y); st1(row0 + 2 * ld + i, f.th);
                    st1(row0 + 3 * ld + i, f.gx); st1(row0 + 4 * ld + i, f.gy);
                    st1(row0 + 5 * ld + i, f.wx); st1(row0 + 6 * ld + i, f.wy);
                    st1(trow + i, REFETCH ? f.t : restart_code(tick));
                    write_norm(a, tile + i, f.x, f.y, f.th, f.gx, f.gy);
                }
            }
            if (uni(most) <= lo + SINK_SLOTS) break;              // block-uniform: every wavefront's worlds have had a slot
            __syncthreads();                                      // the slots are written again
        }
    } else if constexpr (RESTART) {
        // tables of more than eight rows: the groups read their world's rows from memory
        __shared__ TablesShared sh;
        constexpr int WAVES = BLOCK_SMALL / 64;
        const int wave = threadIdx.x >> 6;
        if (done) sh.list[wave][__builtin_amdgcn_mbcnt_hi(static_cast<uint32_t>(done_ballot >> 32),
                                __builtin_amdgcn_mbcnt_lo(static_cast<uint32_t>(done_ballot), 0u))] = static_cast<uint8_t>(threadIdx.x);
        if (lane == 0) sh.count[wave] = static_cast<uint32_t>(__builtin_popcountll(done_ballot));
        __syncthreads();
        uint32_t first[WAVES + 1];
        first[0] = 0;
#pragma unroll
        for (int w = 0; w < WAVES; ++w) first[w + 1] = first[w] + sh.count[w];
        const uint32_t n_done = uni(first[WAVES]);
        // SINK_SPLIT groups of eight lanes per world here too (see RESEED_WORLD): ~8 worlds of a tile finish in a step, the
        // block's 256 lanes are 32 groups
        constexpr uint32_t LANES = RESET_GROUP * SINK_SPLIT, PER_WAVE = 64 / LANES, PER_BLOCK = WAVES * PER_WAVE;
        for (uint32_t qb = static_cast<uint32_t>(wave) * PER_WAVE; qb < n_done; qb += PER_BLOCK) {
            const uint32_t q = qb + (lane / LANES);
            const bool active = q < n_done;
            uint32_t seg = 0;
#pragma unroll
            for (int w = 1; w < WAVES; ++w) seg += (active && q >= first[w]) ? 1u : 0u;
            const uint32_t i = active ? sh.list[seg][q - first[seg]] : 0u;             // an idle group reads a world that exists
            const WorldTable own{t32 + tile, nullptr, tld, i};
            const EnvState f = reset_env_group<RESET_GROUP, RESEED_WORLD, 8, 256, SINK_SPLIT>(
                active, a.seed, static_cast<uint64_t>(a.env_offset + tile) + i, tick, a.waves, a.random_boat, a.random_goal, a.K,
                nullptr, nullptr, nullptr, 0, &own);
            if (active && (lane & (LANES - 1)) == 0) {
                st1(row0 + 0 * ld + i, f.x); st1(row0 + 1 * ld + i, f.y); st1(row0 + 2 * ld + i, f.th);
                st1(row0 + 3 * ld + i, f.gx); st1(row0 + 4 * ld + i, f.gy);
                st1(row0 + 5 * ld + i, f.wx); st1(row0 + 6 * ld + i, f.wy);
                st1(trow + i, f.t);
                write_norm(a, tile + i, f.x, f.y, f.th, f.gx, f.gy);
            }
        }
    }
}

// (registers: 78-114 by instantiation; the attribute only states the floor of two wavefronts per SIMD.  Asking for more
// makes the compiler spill to scratch; what brought the count down from 187-242 was not unrolling the cold loops,
// aqua_device.hpp)
#ifndef AQUA_TABLES_MAX_WAVES                // (timing experiment: the no-restart kernel at the restart kernels' occupancy)
#define AQUA_TABLES_MAX_WAVES 8
#endif
template <int AK, int MODE, int KREG, int SINK_SPLIT = SINK_SPLIT_SHORT>
__global__ __launch_bounds__(BLOCK_SMALL) __attribute__((amdgpu_waves_per_eu(2, AQUA_TABLES_MAX_WAVES))) void step_tables_kernel(const StepArgs a, const float* __restrict__ t32,
                                                                  const double* __restrict__ t64, int64_t tld,
                                                                  float band2, float band2_tight)
{
    static_assert(MODE == AQUA_RESET_NONE || MODE == AQUA_RESET_SAME_STEP || MODE == TABLES_NEXT_STEP_TILE, "one tile per block");
    tick_housekeeping(a);
    tables_step_block<AK, MODE, KREG, SINK_SPLIT>(a, t32, t64, tld, band2, band2_tight, static_cast<int64_t>(blockIdx.x) * BLOCK_SMALL);
}

// Next-step restart with per-world tables: step_ns_kernel's launch split by role (same markers in the time row, same
// grid layouts), the stepping blocks being tables_step_block<AK, NEXT_STEP> and the re-seeding blocks re-seeding each
// finished world eight lanes per world against ITS OWN table (the masked reset's groups, with the tick of this step).
// Nothing of the re-seeding chain -- four dependent global round trips for the rows of a group's table -- is on the
// step's path any more, which is what the same-step form pays for (17.5 us per step at 262 144 worlds, 8 rows).
static_assert(NS_TILE == BLOCK_SMALL, "the stepping role of the per-world next-step kernel is one tables_step_block per block");
template <int AK, bool INTERLEAVE>
__global__ __launch_bounds__(NS_BLOCK) __attribute__((amdgpu_waves_per_eu(2, 8))) void step_tables_ns_kernel(const StepArgs a, const float* __restrict__ t32,
                                                                  const double* __restrict__ t64, int64_t tld,
                                                                  float band2, float band2_tight)
{
    __shared__ ResetShared sh;
    tick_housekeeping(a);
    bool reseed_role;
    uint32_t role;
    if (!ns_role<INTERLEAVE>(static_cast<uint32_t>(a.reseed_blocks), static_cast<uint32_t>((a.N + NS_TILE - 1) / NS_TILE), reseed_role, role)) return;
    const int64_t role_index = role;
    if (!reseed_role) {
#ifndef AQUA_NS_NOMAIN                       // (timing experiment: the re-seeding blocks alone, on a synthetic pending set)
        tables_step_block<AK, AQUA_RESET_NEXT_STEP, 0>(a, t32, t64, tld, band2, band2_tight, role_index * NS_TILE);
#endif
        return;
    }
#ifdef AQUA_NS_NOWORK                        // (timing experiment: the stepping blocks alone)
    return;
#endif
    __builtin_amdgcn_s_setprio(3);
    static_assert(RESET_SCAN == NS_SCAN && RESET_SCAN_ROWS == NS_SCAN_ROWS, "one scan shape for both re-seeding kernels");
    constexpr int WAVES = NS_BLOCK / 64;
    const int64_t base = role_index * NS_SCAN, ld = a.ld, rem = a.N - base;        // rem > 0
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const uint32_t last = static_cast<uint32_t>(rem < NS_SCAN ? rem - 1 : NS_SCAN - 1);
    int32_t tw[NS_SCAN_ROWS];
#pragma unroll
    for (int j = 0; j < NS_SCAN_ROWS; ++j) {              // the loads first
        const uint32_t i = static_cast<uint32_t>(j * NS_BLOCK) + threadIdx.x;
        tw[j] = ld1(a.time + base + (i < last ? i : last));
    }
    const uint64_t tick = launch_tick(a);
    const int32_t restart = done_code(tick - 1);
    uint32_t n_mine = 0;
#pragma unroll
    for (int j = 0; j < NS_SCAN_ROWS; ++j) {              // wavefront-private compaction: ballot + prefix count
        const uint32_t i = static_cast<uint32_t>(j * NS_BLOCK) + threadIdx.x;
#ifdef AQUA_NS_NOMAIN
        const bool p = static_cast<int64_t>(i) < rem && tw[j] != restart &&
                       ((static_cast<uint32_t>(base) + i) * 2654435761u + static_cast<uint32_t>(tick) * 40503u) % 54u == 0u;
#else
        const bool p = static_cast<int64_t>(i) < rem && tw[j] == restart;
#endif
        const uint64_t m = __ballot(p);
        if (p) sh.list[wave][n_mine + __builtin_amdgcn_mbcnt_hi(static_cast<uint32_t>(m >> 32),
                                   __builtin_amdgcn_mbcnt_lo(static_cast<uint32_t>(m), 0u))] = static_cast<uint16_t>(i);
        n_mine += static_cast<uint32_t>(__builtin_popcountll(m));
    }
    if (lane == 0) sh.count[wave] = n_mine;
    __syncthreads();
    uint32_t first[WAVES + 1];
    first[0] = 0;
#pragma unroll
    for (int w = 0; w < WAVES; ++w) first[w + 1] = first[w] + sh.count[w];
    const uint32_t n = uni(first[WAVES]);
    constexpr uint32_t PER_WAVE = 64 / RESET_GROUP, PER_BLOCK = WAVES * PER_WAVE;
    for (uint32_t qb = static_cast<uint32_t>(wave) * PER_WAVE; qb < n; qb += PER_BLOCK) {
        const uint32_t q = qb + (lane / RESET_GROUP);
        const bool active = q < n;
        uint32_t seg = 0;
#pragma unroll
        for (int w = 1; w < WAVES; ++w) seg += (active && q >= first[w]) ? 1u : 0u;
        const uint32_t local = active ? sh.list[seg][q - first[seg]] : 0u;         // an idle group reads a world that exists
        const int64_t i = base + local;
        const WorldTable own{t32 + base, nullptr, tld, local};
        const EnvState f = reset_env_group<RESET_GROUP, RESEED_WORLD>(
            active, a.seed, static_cast<uint64_t>(a.env_offset + i), tick, a.waves, a.random_boat, a.random_goal, a.K, nullptr,
            nullptr, nullptr, 0, &own);
        if (active && (lane & (RESET_GROUP - 1)) == 0) {
            st1(a.state + 0 * ld + i, f.x); st1(a.state + 1 * ld + i, f.y); st1(a.state + 2 * ld + i, f.th);
            st1(a.state + 3 * ld + i, f.gx); st1(a.state + 4 * ld + i, f.gy);
            st1(a.state + 5 * ld + i, f.wx); st1(a.state + 6 * ld + i, f.wy);
            st1(a.time + i, restart_code(tick));
            write_norm(a, i, f.x, f.y, f.th, f.gx, f.gy);
        }
    }
}

// ------------------------------------------------------------------ per-world tables: T steps in one launch
// rollout_kernel for batches in which every world has its own table (of at most KT = 8, 16, 32 or 64 rows): the state stays
// in registers for the whole rollout and the block's tables stay in LDS -- 5 KT floats per world as [row][field][world],
// read conflict-free by the world's own lane in every step and, as a broadcast, by the eight lanes that re-seed it.  Here
// the LDS tile pays (it did not for the one-launch-per-step kernel): every row is used T times.  HBM traffic
// per world-step: the action in, reward and term out.  Restart protocol, markers and results: rollout_kernel's, i.e. T
// launches of the per-step kernels bit for bit.
// The tile is 40 KB (KT = 8) or 80 KB per block.  Up to 16 rows a block of 256 lanes holds 256 worlds; longer tables keep
// the 80 KB by giving the block fewer WORLDS (WPB = 128 for up to 32 rows, 64 for up to 64) -- the block stays 256 lanes,
// the lanes without a world of their own idle through the steps and take their turn at re-seeding: the restart protocol
// (lists, tickets, the serving wavefront rotating with the tick) is then the one of the full block, unchanged.
constexpr int FUSED_TABLE_ROWS_MAX = 64;
static_assert(FUSED_TABLE_ROWS_MAX == AQUA_MAX_OBSTACLES, "every table the library accepts has a fused rollout");
template <int KT, int WPB>
struct RolloutTablesShared {
    RolloutShared r;
    float rows[KT * 5][WPB];
};

template <int KT, int WPB>
__device__ __forceinline__ void serve_reseed_tables(const ReseedTicket& tk, const StepArgs& a, uint64_t tick, int64_t block_first_world,
                                                    RolloutTablesShared<KT, WPB>& sh, int parity, const float* __restrict__ t32_tile, int64_t tld)
{
    constexpr int WAVES = BLOCK_SMALL / 64;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    // Who re-seeds.  A full block (256 worlds): ONE wavefront, a different one every tick, while the others go on to their
    // step.  A block of 128 or 64 worlds has wavefronts without worlds: all of those, sharing the list -- at 64 rows per
    // table and every sixth world restarting per step a single server was the whole step (126 us per step; the
    // one-launch-per-step kernels 113).
    constexpr int WORLD_WAVES = WPB / 64, SERVERS = WPB == BLOCK_SMALL ? 1 : WAVES - WORLD_WAVES;
    // (by tick alone: rotating it by block as serve_reseed() does measured 1-2 % slower here, profiles/r04/fused_duty/)
    const int server = WPB == BLOCK_SMALL ? (wave == static_cast<int>(tick & (WAVES - 1)) ? 0 : -1) : wave - WORLD_WAVES;
    if (tk.n == 0 || server < 0) return;
    uint32_t first[WAVES + 1];
    first[0] = 0;
#pragma unroll
    for (int w = 0; w < WAVES; ++w) first[w + 1] = first[w] + sh.r.count[parity][w];
    // a long table's rows are shared out over 4 or 8 groups of eight lanes (8 rows per lane and pass either way): one world
    // of 64 rows, or two of 32, per wavefront and pass instead of eight -- a block of 64 worlds restarts one or two per step
    constexpr int SPLIT = KT / 8 >= 4 ? KT / 8 : 1, WORLD_LANES = RESET_GROUP * SPLIT;
    constexpr uint32_t PER_PASS = 64 / WORLD_LANES;
    for (uint32_t qb = static_cast<uint32_t>(server) * PER_PASS; qb < tk.n; qb += SERVERS * PER_PASS) {
        const uint32_t q = qb + (lane / WORLD_LANES);
        const bool active = q < tk.n;
        uint32_t seg = 0;
#pragma unroll
        for (int w = 1; w < WAVES; ++w) seg += (active && q >= first[w]) ? 1u : 0u;
        const uint32_t owner = sh.r.list[parity][seg][active ? q - first[seg] : 0];
        const uint64_t world = static_cast<uint64_t>(a.env_offset + block_first_world) + owner;
        const WorldTable own{t32_tile, nullptr, tld, owner};
        const EnvState f = reset_env_group<RESET_GROUP, RESEED_SOA, KT, WPB, SPLIT>(active, a.seed, world, tick, a.waves, a.random_boat, a.random_goal, a.K,
                                                                      nullptr, reinterpret_cast<const ObstF*>(&sh.rows[0][owner]), nullptr, 0, &own);
        if (active && (lane & (WORLD_LANES - 1)) == 0) {
            float* r = sh.r.result[q];
            r[0] = f.x; r[1] = f.y; r[2] = f.th; r[3] = f.gx; r[4] = f.gy; r[5] = f.wx; r[6] = f.wy;
        }
    }
}

template <int AK, int MODE, int KT, int WPB>
__device__ __forceinline__ void rollout_tables_body(const StepArgs& a, const float* __restrict__ t32, const double* __restrict__ t64,
                                                    int64_t tld, float band2, float band2_tight, RolloutTablesShared<KT, WPB>& sh)
{
    static_assert((KT == 8 || KT == 16 || KT == 32 || KT == 64) && KT * WPB <= 16 * BLOCK_SMALL && WPB <= BLOCK_SMALL && WPB % 64 == 0,
                  "an LDS tile of at most 80 KB, whole wavefronts of worlds");
    StepConst k;
    k.W = a.W; k.sigma = a.sigma; k.waves = a.waves; k.time_limit = a.time_limit; k.K = a.K; k.Kc = 0;
    k.band2 = band2; k.band2_tight = band2_tight; k.obst = nullptr; k.obst64 = nullptr; k.quick = nullptr;
    k.touch[0] = k.touch[1] = k.touch[2] = k.touch[3] = 0;
    const uint64_t tick0 = launch_tick(a);
    const int64_t N = a.N, ld = a.ld;
    for (int64_t bbase = static_cast<int64_t>(blockIdx.x) * WPB; bbase < N;
         bbase += static_cast<int64_t>(gridDim.x) * WPB) {
        const int64_t i = bbase + threadIdx.x;
        const bool valid = (WPB == BLOCK_SMALL || threadIdx.x < WPB) && i < N;     // a lane with a world of its own
        const int64_t tile_last = (bbase + WPB <= N ? bbase + WPB : N) - 1;
        const int64_t ic = valid ? i : tile_last;          // the others shadow the tile's last world and write nothing
        const uint32_t off = static_cast<uint32_t>(ic - bbase);                     // < WPB
        EnvState e{a.state[0 * ld + ic], a.state[1 * ld + ic], a.state[2 * ld + ic], a.state[3 * ld + ic],
                   a.state[4 * ld + ic], a.state[5 * ld + ic], a.state[6 * ld + ic], a.time[ic]};
        __syncthreads();                                   // (a block that iterates: the last tile's columns are no longer read)
        if (WPB == BLOCK_SMALL || threadIdx.x < WPB) {
#pragma unroll 8
            for (int j = 0; j < KT; ++j) {
                const int jj = j < a.K ? j : a.K - 1;      // uniform; 1 <= a.K <= KT
                const float* const rb = t32 + (6 * jj) * tld + bbase;
#pragma unroll
                for (int f = 0; f < 5; ++f) sh.rows[j * 5 + f][threadIdx.x] = rb[f * tld + off];
            }
        }
        __syncthreads();
        const WorldTable wt{t32 + bbase, t64 + bbase, tld, off};
        const uint64_t env = static_cast<uint64_t>(a.env_offset + i);
        for (int64_t s = 0; s < a.T; ++s) {
            const uint64_t tick = tick0 + static_cast<uint64_t>(s);
            const int parity = static_cast<int>(s & 1);
            bool pending = false, restart = false;
            ReseedTicket tk{0u, 0u};
            if constexpr (MODE == AQUA_RESET_NEXT_STEP) {
                if (e.t == restart_code(tick - 1)) e.t = 0;
                restart = valid && e.t == done_code(tick - 1);
                pending = valid && e.t < 0;
                tk = publish_reseed(restart, sh.r, parity);
                serve_reseed_tables(tk, a, tick, bbase, sh, parity, t32 + bbase, tld);
            }
            int idx = 2;
            float vl = 0.5f, vr = 0.5f;
            if constexpr (AK == AQUA_ACT_U8) idx = fold_index(static_cast<const uint8_t*>(a.action)[s * a.action_step_stride + ic]);
            if constexpr (AK == AQUA_ACT_I32) idx = fold_index(static_cast<const int32_t*>(a.action)[s * a.action_step_stride + ic]);
            if constexpr (AK == AQUA_ACT_I64) idx = fold_index(static_cast<const int64_t*>(a.action)[s * a.action_step_stride + ic]);
            if constexpr (AK == AQUA_ACT_F32X2) {
                const float* base = static_cast<const float*>(a.action) + s * a.action_step_stride;
                vl = base[ic]; vr = base[a.action_ld + ic];
            }
            const uint64_t seed = a.seed;
            uint32_t w0[1], w1[1];
            pair_draws<1, false>(seed, env, tick, STREAM_STEP, w0, w1);
            const float u0 = u_pm1(w0[0]), u1 = u_pm1(w1[0]);
            if constexpr (AK >= AQUA_ACT_SAMPLE_D) {
                pair_draws<1, false>(seed, env, tick, STREAM_ACT, w0, w1);
                if constexpr (AK == AQUA_ACT_SAMPLE_D) idx = sample_discrete(w0[0]);
                else { vl = sample_thrust(w0[0]); vr = sample_thrust(w1[0]); }
            }
            if constexpr (AK == AQUA_ACT_BEARING) idx = bearing_action(e.x, e.y, e.th, e.gx, e.gy);
            const Motion m = decode_motion<AK>(k, idx, vl, vr);
            const EnvState before = e;
            EnvState after = e;
            float rew;
            uint32_t code;
            ObstF rows[KT];                                 // this step's copy of the lane's column
#pragma unroll
            for (int j = 0; j < KT; ++j) {
                rows[j].cx = sh.rows[j * 5 + 0][off]; rows[j].cy = sh.rows[j * 5 + 1][off];
                rows[j].hx = sh.rows[j * 5 + 2][off]; rows[j].hy = sh.rows[j * 5 + 3][off];
                rows[j].r2 = sh.rows[j * 5 + 4][off]; rows[j].w = 1.0f;
            }
            const bool knife = fast_step<true, QUICK_NEVER, KT>(after, m.h, m.w, m.chord, u0, u1, k, rew, code, &wt, rows) && valid && !pending;
            if (any_lane(knife)) {
                if (knife) {
                    const ExactOut o = exact_step_world(before.x, before.y, before.th, before.gx, before.gy, before.wx, before.wy,
                                                        after.t, exact_motion<AK>(m), k.K, k.band2, k.time_limit, wt);
                    after.x = o.x; after.y = o.y; after.th = o.th; rew = o.reward; code = o.term;
                }
            }
            if (pending) { rew = 0.0f; code = 0u; }
            else e = after;
            if (valid) {
                a.reward[s * a.out_step_stride + i] = rew;
                a.term[s * a.out_step_stride + i] = static_cast<uint8_t>(code);
            }
            const bool done = valid && code != 0u;
            if constexpr (MODE == AQUA_RESET_NEXT_STEP) {
                collect_reseed(e, restart, tk, sh.r);
                if (restart) e.t = restart_code(tick);
                else if (done) e.t = done_code(tick);
            } else if constexpr (MODE == AQUA_RESET_SAME_STEP) {
                const ReseedTicket t1 = publish_reseed(done, sh.r, parity);
                serve_reseed_tables(t1, a, tick, bbase, sh, parity, t32 + bbase, tld);
                collect_reseed(e, done, t1, sh.r);
            }
        }
        if (valid) {
            a.state[0 * ld + i] = e.x; a.state[1 * ld + i] = e.y; a.state[2 * ld + i] = e.th;
            a.state[3 * ld + i] = e.gx; a.state[4 * ld + i] = e.gy;
            a.state[5 * ld + i] = e.wx; a.state[6 * ld + i] = e.wy;
            a.time[i] = e.t;
        }
    }
}

template <int AK, int MODE>
__global__ __launch_bounds__(BLOCK_SMALL) __attribute__((amdgpu_waves_per_eu(2, 3))) void rollout_tables_kernel(
    const StepArgs a, const float* __restrict__ t32, const double* __restrict__ t64, int64_t tld, float band2, float band2_tight)
{
    __shared__ RolloutTablesShared<8, BLOCK_SMALL> sh;
    rollout_tables_body<AK, MODE, 8, BLOCK_SMALL>(a, t32, t64, tld, band2, band2_tight, sh);
}

// tables of 9..16 rows: 80 KB of LDS per block, two blocks per CU
template <int AK, int MODE>
__global__ __launch_bounds__(BLOCK_SMALL) __attribute__((amdgpu_waves_per_eu(1, 2))) void rollout_tables16_kernel(
    const StepArgs a, const float* __restrict__ t32, const double* __restrict__ t64, int64_t tld, float band2, float band2_tight)
{
    __shared__ RolloutTablesShared<16, BLOCK_SMALL> sh;
    rollout_tables_body<AK, MODE, 16, BLOCK_SMALL>(a, t32, t64, tld, band2, band2_tight, sh);
}

// tables of 17..32 rows: 128 worlds per block; 33..64 rows: 64 worlds per block (one block per CU)
template <int AK, int MODE>
__global__ __launch_bounds__(BLOCK_SMALL) __attribute__((amdgpu_waves_per_eu(1, 2))) void rollout_tables32_kernel(
    const StepArgs a, const float* __restrict__ t32, const double* __restrict__ t64, int64_t tld, float band2, float band2_tight)
{
    __shared__ RolloutTablesShared<32, 128> sh;
    rollout_tables_body<AK, MODE, 32, 128>(a, t32, t64, tld, band2, band2_tight, sh);
}

template <int AK, int MODE>
__global__ __launch_bounds__(BLOCK_SMALL) __attribute__((amdgpu_waves_per_eu(1, 2))) void rollout_tables64_kernel(
    const StepArgs a, const float* __restrict__ t32, const double* __restrict__ t64, int64_t tld, float band2, float band2_tight)
{
    __shared__ RolloutTablesShared<64, 64> sh;
    rollout_tables_body<AK, MODE, 64, 64>(a, t32, t64, tld, band2, band2_tight, sh);
}

// Masked reset against per-world tables.  A block reads the mask of RESET_SCAN worlds and compacts the selected ones
// into an LDS list (ballot + prefix count per wavefront, one barrier).  Few selected (the restart after a step:
// ~2 % of the worlds): they are re-seeded eight lanes per world like everywhere else -- with one world per lane,
// three wavefronts of four would each loop over rejection attempts for one or two live lanes.  Many selected (a
// whole-batch reset): one world per lane, the serial specification.  Bit-identical either way.
__global__ __launch_bounds__(BLOCK_SMALL) void reset_tables_kernel(const StepArgs a, const uint8_t* __restrict__ mask,
                                                                   const float* __restrict__ t32, int64_t tld)
{
    __shared__ ResetShared sh;
    constexpr int WAVES = BLOCK_SMALL / 64;
    const uint64_t tick = launch_tick(a);
    const int64_t base = static_cast<int64_t>(blockIdx.x) * RESET_SCAN, ld = a.ld, rem = a.N - base;   // rem > 0
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    bool sel[RESET_SCAN_ROWS];
    uint32_t n_mine = 0;
#pragma unroll
    for (int j = 0; j < RESET_SCAN_ROWS; ++j) {
        const uint32_t i = static_cast<uint32_t>(j * BLOCK_SMALL) + threadIdx.x;
        sel[j] = static_cast<int64_t>(i) < rem && (mask == nullptr || mask[base + i] != 0);
    }
#pragma unroll
    for (int j = 0; j < RESET_SCAN_ROWS; ++j) {
        const uint32_t i = static_cast<uint32_t>(j * BLOCK_SMALL) + threadIdx.x;
        const uint64_t m = __ballot(sel[j]);
        if (sel[j]) sh.list[wave][n_mine + __builtin_amdgcn_mbcnt_hi(static_cast<uint32_t>(m >> 32),
                                       __builtin_amdgcn_mbcnt_lo(static_cast<uint32_t>(m), 0u))] = static_cast<uint16_t>(i);
        n_mine += static_cast<uint32_t>(__builtin_popcountll(m));
    }
    if (lane == 0) sh.count[wave] = n_mine;
    __syncthreads();
    uint32_t first[WAVES + 1];
    first[0] = 0;
#pragma unroll
    for (int w = 0; w < WAVES; ++w) first[w + 1] = first[w] + sh.count[w];
    const uint32_t n = uni(first[WAVES]);
    const auto store = [&](int64_t i, const EnvState& e) {
        a.state[0 * ld + i] = e.x; a.state[1 * ld + i] = e.y; a.state[2 * ld + i] = e.th;
        a.state[3 * ld + i] = e.gx; a.state[4 * ld + i] = e.gy;
        a.state[5 * ld + i] = e.wx; a.state[6 * ld + i] = e.wy;
        a.time[i] = e.t;
    };
    if (n > static_cast<uint32_t>(RESET_DENSE)) {
#pragma unroll 1
        for (int j = 0; j < RESET_SCAN_ROWS; ++j) {
            if (!sel[j]) continue;
            const uint32_t local = static_cast<uint32_t>(j * BLOCK_SMALL) + threadIdx.x;
            const int64_t i = base + local;
            const WorldTable wt{t32 + base, nullptr, tld, local};
            store(i, reset_env_world(a.seed, static_cast<uint64_t>(a.env_offset + i), tick, a.waves, a.random_boat,
                                     a.random_goal, a.K, wt));
        }
        return;
    }
    constexpr uint32_t PER_WAVE = 64 / RESET_GROUP, PER_BLOCK = WAVES * PER_WAVE;
    for (uint32_t qb = static_cast<uint32_t>(wave) * PER_WAVE; qb < n; qb += PER_BLOCK) {
        const uint32_t q = qb + (lane / RESET_GROUP);
        const bool active = q < n;
        uint32_t seg = 0;
#pragma unroll
        for (int w = 1; w < WAVES; ++w) seg += (active && q >= first[w]) ? 1u : 0u;
        const uint32_t local = active ? sh.list[seg][q - first[seg]] : 0u;         // an idle group reads a world that exists
        const int64_t i = base + local;
        const WorldTable wt{t32 + base, nullptr, tld, local};
        const EnvState e = reset_env_group<RESET_GROUP, RESEED_WORLD>(active, a.seed, static_cast<uint64_t>(a.env_offset + i), tick,
                                                                      a.waves, a.random_boat, a.random_goal, a.K, nullptr, nullptr,
                                                                      nullptr, 0, &wt);
        if (active && (lane & (RESET_GROUP - 1)) == 0) store(i, e);
    }
}

// ------------------------------------------------------------------ masked reset
__global__ __launch_bounds__(BLOCK_SMALL) void reset_kernel(const StepArgs a, const uint8_t* __restrict__ mask)
{
    const ObstPtr obst = obstacle_rows(a.obst_blob);
    const uint64_t tick = launch_tick(a);
    const int64_t N = a.N, ld = a.ld;
    for (int64_t i = static_cast<int64_t>(blockIdx.x) * BLOCK_SMALL + threadIdx.x; i < N;
         i += static_cast<int64_t>(gridDim.x) * BLOCK_SMALL) {
        if (mask != nullptr && mask[i] == 0) continue;
        const EnvState e = reset_env(a.seed, static_cast<uint64_t>(a.env_offset + i), tick, a.waves, a.random_boat,
                                     a.random_goal, a.K, obst);
        a.state[0 * ld + i] = e.x; a.state[1 * ld + i] = e.y; a.state[2 * ld + i] = e.th;
        a.state[3 * ld + i] = e.gx; a.state[4 * ld + i] = e.gy;
        a.state[5 * ld + i] = e.wx; a.state[6 * ld + i] = e.wy;
        a.time[i] = e.t;
    }
}

__global__ void tick_kernel(uint64_t* tick_base, uint64_t delta) { *tick_base += delta; }

// The done-mask blocks' way into a receive buffer (the own one, or another GPU's through its IPC mapping): single-wavefront
// workgroups that copy 16-byte words, four loads in flight per lane; 64 threads, a few registers, no LDS.  The grid grows
// with the block (one workgroup per 32 KiB, 8 to 256 of them) so that the copy is SHORT: measured on one rank
// (profiles/r03/exchange_overhead_one_rank.txt) the step kernels of the other stream lose ~2 us per launch for as long as
// ANY other kernel is resident -- 8 wavefronts that take a millisecond for a 16 MB block cost the step stream as much as
// RCCL's all-gather does (18 %), the same bytes moved in ~20 us cost nothing measurable.
constexpr unsigned COPY_BLOCKS_MIN = 8, COPY_BLOCKS_MAX = 256;
constexpr size_t COPY_BYTES_PER_BLOCK = 32768;
struct alignas(16) Word16 { unsigned long long lo, hi; };
template <typename W>
__global__ __launch_bounds__(64) void copy_words_kernel(W* __restrict__ dst, const W* __restrict__ src, size_t n)
{
    const size_t stride = static_cast<size_t>(gridDim.x) * 64;
    size_t i = static_cast<size_t>(blockIdx.x) * 64 + threadIdx.x;
    for (; i + 3 * stride < n; i += 4 * stride) {
        const W a = src[i], b = src[i + stride], c = src[i + 2 * stride], d = src[i + 3 * stride];
        dst[i] = a; dst[i + stride] = b; dst[i + 2 * stride] = c; dst[i + 3 * stride] = d;
    }
    for (; i < n; i += stride) dst[i] = src[i];
}

// The same copy into up to eight receive buffers at once (blockIdx.y picks the destination): ONE launch, in stream order
// behind the last step of a region, delivers the region's last block to every GPU of the node -- no host round trip and
// no second stream between that step and its copies; its duration is the slowest link's.
constexpr int COPY_FANOUT_MAX = 8;
struct CopyFanout { void* dst[COPY_FANOUT_MAX]; };
template <typename W>
__global__ __launch_bounds__(64) void copy_fanout_kernel(CopyFanout f, const W* __restrict__ src, size_t n)
{
    W* const dst = static_cast<W*>(f.dst[blockIdx.y]);
    const size_t stride = static_cast<size_t>(gridDim.x) * 64;
    size_t i = static_cast<size_t>(blockIdx.x) * 64 + threadIdx.x;
    for (; i + 3 * stride < n; i += 4 * stride) {
        const W a = src[i], b = src[i + stride], c = src[i + 2 * stride], d = src[i + 3 * stride];
        dst[i] = a; dst[i + stride] = b; dst[i + 2 * stride] = c; dst[i + 3 * stride] = d;
    }
    for (; i < n; i += stride) dst[i] = src[i];
}

// ring[r][(cursor + i) % capacity] = src[r][i]: the batch lands in consecutive slots, so both sides are coalesced
// (the wrap splits at most one wavefront's store)
template <typename T>
__global__ __launch_bounds__(BLOCK_SMALL) void ring_write_kernel(T* __restrict__ ring, int64_t ring_ld, int64_t capacity,
                                                                 int64_t cursor, const T* __restrict__ src, int64_t src_ld,
                                                                 int rows, int64_t N)
{
    for (int64_t i = static_cast<int64_t>(blockIdx.x) * BLOCK_SMALL + threadIdx.x; i < N;
         i += static_cast<int64_t>(gridDim.x) * BLOCK_SMALL) {
        int64_t slot = cursor + i;                       // cursor < capacity, i < N <= capacity
        if (slot >= capacity) slot -= capacity;
        for (int r = 0; r < rows; ++r) ring[r * ring_ld + slot] = src[r * src_ld + i];
    }
}

__global__ __launch_bounds__(BLOCK_SMALL) void obs_norm_kernel(const float* __restrict__ state, int64_t ld, int64_t N,
                                                               const uint8_t* __restrict__ mask, float* __restrict__ out)
{
    for (int64_t i = static_cast<int64_t>(blockIdx.x) * BLOCK_SMALL + threadIdx.x; i < N;
         i += static_cast<int64_t>(gridDim.x) * BLOCK_SMALL) {
        if (mask != nullptr && mask[i] == 0) continue;
        out[0 * ld + i] = state[0 * ld + i] * 0.01f;
        out[1 * ld + i] = state[1 * ld + i] * 0.01f;
        out[2 * ld + i] = fmaf(state[2 * ld + i], 0.15915494309189535f, 0.5f);
        out[3 * ld + i] = state[3 * ld + i] * 0.01f;
        out[4 * ld + i] = state[4 * ld + i] * 0.01f;
    }
}

// ------------------------------------------------------------------ host side
thread_local char g_err[512] = "";

// The launch tables below name every action kind after AQUA_ACT_U8 through this macro.  Development builds only
// (-DAQUA_DEV_U8_ONLY, never the shipped library): one action kind instead of seven, a sixth of the compile time while a
// kernel is being worked on; the other kinds then fail with hipErrorInvalidValue.
#ifdef AQUA_DEV_U8_ONLY
#define AQUA_DEV_OTHER_KINDS(LAUNCH)
#elif defined(AQUA_DEV_TWO_KINDS)
#define AQUA_DEV_OTHER_KINDS(LAUNCH) LAUNCH(AQUA_ACT_SAMPLE_D)
#else
#define AQUA_DEV_OTHER_KINDS(LAUNCH)                                                                       \
    LAUNCH(AQUA_ACT_I32) LAUNCH(AQUA_ACT_I64) LAUNCH(AQUA_ACT_F32X2) LAUNCH(AQUA_ACT_SAMPLE_D) LAUNCH(AQUA_ACT_SAMPLE_C) \
    LAUNCH(AQUA_ACT_BEARING)
#endif

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

int grid_for(int64_t items, int block, int max_grid)
{
    int64_t g = (items + block - 1) / block;
    if (g < 1) g = 1;
    if (g > max_grid) g = max_grid;
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
    a.state = state; a.ld = ld; a.time = time; a.obst_blob = K > 0 ? blob : nullptr; a.tick_base = tick_base;
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

// Events attached to a LAUNCH (aqua_rollout_events_f32): `start` takes the kernel's own start time, `stop` its end time --
// no marker packet of their own in the queue (a recorded stream event is one, and a pair of them around a region costs
// the region 12-14 us: profiles/r03/burst_timeline.txt).
struct LaunchEvents {
    hipEvent_t start = nullptr, stop = nullptr;
};

template <typename Kernel, typename... Args>
void launch_kernel(Kernel kernel, dim3 grid, dim3 block, hipStream_t s, const LaunchEvents& ev, Args... args)
{
    if (ev.start == nullptr && ev.stop == nullptr) hipLaunchKernelGGL(kernel, grid, block, 0, s, args...);
    else hipExtLaunchKernelGGL(kernel, grid, block, 0, s, ev.start, ev.stop, 0, args...);
}

void fill_ns_args(NsArgs& a, const StepArgs& a0, int kind, int64_t first, int64_t n, bool housekeeping_head, bool housekeeping_tail);

// auto_reset 0 / 1: step_kernel, one launch per NS_LAUNCH_MAX_WORLDS worlds
hipError_t launch_step(const StepArgs& a0, int kind, hipStream_t s, const LaunchEvents& ev = {})
{
    const bool plain = a0.auto_reset == 0;              // no restart: no list, no barrier, no re-seeding code in the kernel
    const bool wb = !plain && a0.N >= STORE_WB_SAME_STEP_MIN;
    const bool small = NS_TABLE_ROWS > 0 && a0.K <= NS_TABLE_ROWS;
    for (int64_t first = 0; first < a0.N; first += NS_LAUNCH_MAX_WORLDS) {
        const int64_t n = a0.N - first < NS_LAUNCH_MAX_WORLDS ? a0.N - first : NS_LAUNCH_MAX_WORLDS;
        const bool head = first == 0, tail = first + n == a0.N;
        LaunchEvents e;
        if (head) e.start = ev.start;
        if (tail) e.stop = ev.stop;
        NsArgs a;
        fill_ns_args(a, a0, kind, first, n, head, tail);
        const dim3 grid(static_cast<unsigned>((n + TILE_WORLDS - 1) / TILE_WORLDS)), block(TILE_WORLDS);
#define AQUA_STEP_LAUNCH(AK)                                                                                                     \
    case AK:                                                                                                                     \
        if (plain && small) launch_kernel((step_kernel<AK, true, false>), grid, block, s, e, a);                                 \
        else if (plain) launch_kernel((step_kernel<AK, false, false>), grid, block, s, e, a);                                    \
        else if (small && wb) launch_kernel((step_kernel<AK, true, true, true>), grid, block, s, e, a);                          \
        else if (wb) launch_kernel((step_kernel<AK, false, true, true>), grid, block, s, e, a);                                  \
        else if (small) launch_kernel((step_kernel<AK, true, true>), grid, block, s, e, a);                                      \
        else launch_kernel((step_kernel<AK, false, true>), grid, block, s, e, a);                                                \
        break;
        switch (kind) {
            AQUA_STEP_LAUNCH(AQUA_ACT_U8)
            AQUA_DEV_OTHER_KINDS(AQUA_STEP_LAUNCH)
            default: return hipErrorInvalidValue;
        }
#undef AQUA_STEP_LAUNCH
        const hipError_t rc = hipGetLastError();
        if (rc != hipSuccess) return rc;
    }
    return hipSuccess;
}

// StepArgs -> the NsArgs of ONE launch over worlds [first, first + n) of the caller's batch (n <= NS_LAUNCH_MAX_WORLDS,
// first a multiple of NS_SCAN): pointers advanced to the range
void fill_ns_args(NsArgs& a, const StepArgs& a0, int kind, int64_t first, int64_t n, bool housekeeping_head, bool housekeeping_tail)
{
    std::memset(&a, 0, sizeof(a));
    for (int r = 0; r < 7; ++r) a.row[r] = a0.state + r * a0.ld + first;
    a.time = a0.time + first;
    const size_t esz = action_elem_bytes(kind);
    a.action = a0.action ? static_cast<const char*>(a0.action) + static_cast<size_t>(first) * esz : nullptr;
    a.action_hi = (kind == AQUA_ACT_F32X2 && a0.action) ? static_cast<const float*>(a0.action) + a0.action_ld + first : nullptr;
    a.reward = a0.reward + first;
    a.term = a0.term + first;
    a.done_bits = a0.done_bits ? a0.done_bits + first / 64 : nullptr;
    a.obst_blob = a0.obst_blob; a.tick_base = a0.tick_base; a.seed = a0.seed; a.tick = a0.tick;
    a.env_offset = a0.env_offset + first;
    a.N = static_cast<uint32_t>(n);
    a.reseed_blocks = static_cast<uint32_t>((n + NS_SCAN - 1) / NS_SCAN);
    a.K = a0.K; a.time_limit = a0.time_limit; a.W = a0.W; a.sigma = a0.sigma;
    a.flags = (static_cast<uint32_t>(a0.waves) & NS_WAVES_MASK) | (a0.random_boat ? NS_RANDOM_BOAT : 0u) | (a0.random_goal ? NS_RANDOM_GOAL : 0u);
    if (a0.noise != nullptr) { a.flags |= NS_HAS_NOISE; a.noise[0] = a0.noise + first; a.noise[1] = a0.noise + a0.noise_ld + first; }
    if (a0.obs_norm != nullptr) { a.flags |= NS_HAS_NORM; for (int r = 0; r < 5; ++r) a.norm[r] = a0.obs_norm + r * a0.ld + first; }
    if (a0.N > DONE_WORD_WRITE_THROUGH_MAX_WORLDS) a.flags |= NS_DONE_WORD_WB;       // (store_done_word(): by the BATCH's size)
    if (housekeeping_head) a.tick_copy_to = a0.tick_copy_to;         // see tick_housekeeping(): block 0 of ONE launch per step
    if (housekeeping_tail) { a.tick_bump_to = a0.tick_bump_to; a.tick_bump = a0.tick_bump; }
}

hipError_t launch_step_ns_range(const StepArgs& a0, int kind, int64_t first, int64_t n, bool interleave, bool wb, bool housekeeping_head,
                                bool housekeeping_tail, hipStream_t s, const LaunchEvents& ev)
{
    NsArgs a;
    fill_ns_args(a, a0, kind, first, n, housekeeping_head, housekeeping_tail);
    const int64_t tiles = interleave ? (static_cast<int64_t>(a.reseed_blocks) + 7) / 8 * 8 * (NS_SCAN / NS_TILE + 1)
                                     : (n + NS_TILE - 1) / NS_TILE + a.reseed_blocks;
    const dim3 grid(static_cast<unsigned>(tiles)), block(NS_BLOCK);
    const bool small = NS_TABLE_ROWS > 0 && a.K <= NS_TABLE_ROWS;
#define AQUA_NS_LAUNCH(AK)                                                                           \
    case AK:                                                                                         \
        if (small && wb) launch_kernel((step_ns_kernel<AK, true, true, true>), grid, block, s, ev, a);          \
        else if (wb) launch_kernel((step_ns_kernel<AK, false, true, true>), grid, block, s, ev, a);             \
        else if (small && interleave) launch_kernel((step_ns_kernel<AK, true, true>), grid, block, s, ev, a);   \
        else if (small) launch_kernel((step_ns_kernel<AK, true, false>), grid, block, s, ev, a);                \
        else if (interleave) launch_kernel((step_ns_kernel<AK, false, true>), grid, block, s, ev, a);           \
        else launch_kernel((step_ns_kernel<AK, false, false>), grid, block, s, ev, a);                          \
        break;
    switch (kind) {
        AQUA_NS_LAUNCH(AQUA_ACT_U8)
        AQUA_DEV_OTHER_KINDS(AQUA_NS_LAUNCH)
        default: return hipErrorInvalidValue;
    }
#undef AQUA_NS_LAUNCH
    return hipGetLastError();
}

hipError_t launch_step_ns(const StepArgs& a, int kind, hipStream_t s, const LaunchEvents& ev = {})
{
    // layout and store policy follow the BATCH, not the launch: a batch of more than NS_LAUNCH_MAX_WORLDS worlds is a few
    // launches of that many (their offsets inside a row must fit 32 bits), each with the kernels of its whole
    const bool interleave = a.N >= NS_INTERLEAVE_MIN;
    const bool wb = a.N >= STORE_WB_NEXT_STEP_MIN && a.N <= STORE_WB_NEXT_STEP_MAX;      // implies interleave
    for (int64_t first = 0; first < a.N; first += NS_LAUNCH_MAX_WORLDS) {
        const int64_t n = a.N - first < NS_LAUNCH_MAX_WORLDS ? a.N - first : NS_LAUNCH_MAX_WORLDS;
        const bool head = first == 0, tail = first + n == a.N;
        LaunchEvents e;
        if (head) e.start = ev.start;
        if (tail) e.stop = ev.stop;
        const hipError_t rc = launch_step_ns_range(a, kind, first, n, interleave, wb, head, tail, s, e);
        if (rc != hipSuccess) return rc;
    }
    return hipSuccess;
}

hipError_t launch_step_any(const StepArgs& a, int kind, hipStream_t s, const LaunchEvents& ev = {})
{
    if (a.auto_reset == AQUA_RESET_NEXT_STEP) return launch_step_ns(a, kind, s, ev);
    return launch_step(a, kind, s, ev);
}

int check_step_buffers(int64_t N, const void* action, int action_kind, int64_t action_ld, const float* noise,
                       int64_t noise_ld, const float* reward, const uint8_t* term)
{
    if (action_kind < AQUA_ACT_U8 || action_kind > AQUA_ACT_BEARING)
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

template <typename T>
int ring_write(T* ring, int64_t ring_ld, int64_t capacity, int64_t cursor, const T* src, int64_t src_ld, int rows, int64_t N,
               void* stream)
{
    if (capacity <= 0 || ring_ld < capacity) return fail(AQUA_E_INVALID, "bad ring: capacity=%lld ring_ld=%lld", (long long)capacity, (long long)ring_ld);
    if (N < 0 || N > capacity || src_ld < N) return fail(AQUA_E_INVALID, "bad sizes: N=%lld capacity=%lld src_ld=%lld", (long long)N, (long long)capacity, (long long)src_ld);
    if (cursor < 0 || cursor >= capacity) return fail(AQUA_E_INVALID, "cursor %lld outside [0, capacity)", (long long)cursor);
    if (rows < 0 || rows > 64) return fail(AQUA_E_INVALID, "rows=%d outside [0, 64]", rows);
    if (N == 0 || rows == 0) return 0;
    if (ring == nullptr || src == nullptr) return fail(AQUA_E_INVALID, "ring/src is NULL");
    if (!aligned(ring, sizeof(T)) || !aligned(src, sizeof(T))) return fail(AQUA_E_ALIGN, "ring/src not aligned to the element size");
    hipLaunchKernelGGL((ring_write_kernel<T>), dim3(grid_for(N, BLOCK_SMALL, 4096)), dim3(BLOCK_SMALL), 0,
                       static_cast<hipStream_t>(stream), ring, ring_ld, capacity, cursor, src, src_ld, rows, N);
    const hipError_t e = hipGetLastError();
    return e == hipSuccess ? 0 : hip_fail(e, "aqua_ring_write launch");
}

}  // namespace

struct AquaGraph {
    hipGraph_t graph;
    hipGraphExec_t exec;
};

struct AquaEvent {
    hipEvent_t event;
};

extern "C" {

int aqua_version(void) { return AQUA_ABI_VERSION; }

const char* aqua_last_error(void) { return g_err; }

size_t aqua_obstacle_blob_bytes(int K)
{
    if (K <= 0) return 0;
    const size_t used = sizeof(ObstHeader) + static_cast<size_t>(K) * (sizeof(ObstF) + 5 * sizeof(double));
    if (K <= QUICK_MAX) return quick_offset(K) + QUICK_VECS * sizeof(f32x4);   // header, rows, float64 rows, quick table
    return used < AQUA_BLOB_MIN_BYTES ? AQUA_BLOB_MIN_BYTES : used;     // the kernels touch the first five 64-byte lines
}

int aqua_pack_obstacles(const double* rows, int K, void* blob_host, size_t blob_bytes)
{
    if (K < 0 || K > AQUA_MAX_OBSTACLES) return fail(AQUA_E_INVALID, "K=%d outside [0, %d]", K, AQUA_MAX_OBSTACLES);
    if (K == 0) return 0;
    if (rows == nullptr || blob_host == nullptr) return fail(AQUA_E_INVALID, "rows/blob is NULL");
    if (blob_bytes < aqua_obstacle_blob_bytes(K)) return fail(AQUA_E_INVALID, "blob too small");
    ObstHeader* h = static_cast<ObstHeader*>(blob_host);
    ObstF* f = reinterpret_cast<ObstF*>(static_cast<char*>(blob_host) + sizeof(ObstHeader));
    double* d = reinterpret_cast<double*>(static_cast<char*>(blob_host) + sizeof(ObstHeader) + sizeof(ObstF) * K);
    std::memset(blob_host, 0, aqua_obstacle_blob_bytes(K));
    int n_circles = 0;
    for (int k = 0; k < K; ++k) {
        const double* o = rows + 5 * k;
        if (!(o[2] == 0.0 || o[2] == 1.0)) return fail(AQUA_E_INVALID, "obstacle %d: kind must be 0 or 1", k);
        n_circles += o[2] == 0.0;
    }
    double r_max = 0.0;
    int ic = 0, ir = n_circles;                          // circles first, rectangles after (original order kept)
    for (int k = 0; k < K; ++k) {
        const double* o = rows + 5 * k;
        double hx = 0, hy = 0, R = 2.5;                 // boat radius (aqua.py:75)
        if (o[2] == 0.0) R += o[3]; else { hx = o[3] / 2; hy = o[4] / 2; }
        if (!(R > 0.0) || hx < 0.0 || hy < 0.0 || (o[2] == 0.0 && o[3] < 0.0))
            return fail(AQUA_E_INVALID, "obstacle %d: negative size", k);
        const int slot = o[2] == 0.0 ? ic++ : ir++;
        f[slot].cx = static_cast<float>(o[0]); f[slot].cy = static_cast<float>(o[1]);
        f[slot].hx = static_cast<float>(hx); f[slot].hy = static_cast<float>(hy);
        f[slot].r2 = static_cast<float>(R * R);
        if (R > r_max) r_max = R;
        for (int j = 0; j < 5; ++j) d[5 * slot + j] = o[j];
    }
    h->n_obstacles = K;
    h->n_circles = n_circles;
    h->r_max = static_cast<float>(r_max);
    h->band2 = static_cast<float>(2.5 * (r_max + static_cast<double>(BAND)) * static_cast<double>(BAND));
    const auto tight = [](double R) {
        return 2.5 * (R + static_cast<double>(BAND)) * static_cast<double>(BAND_TIGHT) + 4.0 * 1.1920929e-7 * R * R;
    };
    h->band2_tight = static_cast<float>(tight(r_max));
    for (int k = 0; k < K; ++k)                          // per-obstacle scale of the compensated margin (ObstF::w)
        f[k].w = static_cast<float>(static_cast<double>(h->band2_tight) / tight(std::sqrt(static_cast<double>(f[k].r2))));
    if (K <= QUICK_MAX) {                                // quick table: the first look's operands, four obstacles per group
        float* q = reinterpret_cast<float*>(static_cast<char*>(blob_host) + quick_offset(K));
        const auto group = [&](int vec0, int j) { return q + 4 * vec0 + (j & 3); };      // + 4 * field
        for (int j = 0; j < QUICK_MAX; ++j) {
            float* c = group(j < 4 ? QUICK_C0 : QUICK_C1, j);
            const bool used = j < n_circles;
            c[0] = used ? f[j].cx : 0.0f; c[4] = used ? f[j].cy : 0.0f; c[8] = used ? -f[j].r2 : QUICK_EMPTY_NR2;
            float* r = group(j < 4 ? QUICK_R0 : QUICK_R1, j);
            const int row = n_circles + j;
            const bool have = row < K;
            r[0] = have ? f[row].cx : 0.0f; r[4] = have ? f[row].cy : 0.0f;
            r[8] = have ? f[row].hx : 0.0f; r[12] = have ? f[row].hy : 0.0f; r[16] = have ? -f[row].r2 : QUICK_EMPTY_NR2;
        }
        h->reserved[0] = static_cast<int32_t>(quick_offset(K));
    }
    return 0;
}

void aqua_discrete_constants(float out[9])
{
    const float v[9] = {ACT_H_TURN, -ACT_H_TURN, ACT_H_LINE, ACT_W_TURN, -ACT_W_TURN, ACT_W_LINE,
                        ACT_C_TURN, ACT_C_TURN, ACT_C_LINE};
    for (int j = 0; j < 9; ++j) out[j] = v[j];
}

#if AQUA_STAMPS
#include "aqua_tuning.inc"               // diagnostic builds only
#endif

int aqua_step_f32(const AquaParams* p, const void* obst_blob_dev, int K, int64_t N, int64_t env_offset,
                  float* state, int64_t ld, int32_t* time, const void* action, int action_kind,
                  int64_t action_ld, const float* noise, int64_t noise_ld, uint64_t seed, uint64_t tick,
                  const uint64_t* tick_base_dev, float* reward, uint8_t* term, uint64_t* done_bits,
                  float* obs_norm, int auto_reset, void* stream)
{
    StepArgs a;
    int rc = fill_args(a, p, obst_blob_dev, K, N, env_offset, state, ld, time, seed, tick, tick_base_dev);
    if (rc) return rc;
    rc = check_step_buffers(N, action, action_kind, action_ld, noise, noise_ld, reward, term);
    if (rc) return rc;
    if (done_bits != nullptr && !aligned(done_bits, 8)) return fail(AQUA_E_ALIGN, "done_bits must be 8-byte aligned");
    if (auto_reset < 0 || auto_reset > 2) return fail(AQUA_E_INVALID, "auto_reset must be 0, 1 or 2");
    if (N == 0) return 0;
    a.action = action; a.action_ld = action_ld; a.noise = noise; a.noise_ld = noise_ld;
    a.reward = reward; a.term = term; a.done_bits = done_bits; a.obs_norm = obs_norm; a.auto_reset = auto_reset;
    if (obs_norm != nullptr && !aligned(obs_norm, 4)) return fail(AQUA_E_ALIGN, "obs_norm must be 4-byte aligned");
    const hipError_t e = launch_step_any(a, action_kind, static_cast<hipStream_t>(stream));
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
    hipLaunchKernelGGL(reset_kernel, dim3(grid_for(N, BLOCK_SMALL, 2048)), dim3(BLOCK_SMALL), 0, static_cast<hipStream_t>(stream), a, mask);
    const hipError_t e = hipGetLastError();
    return e == hipSuccess ? 0 : hip_fail(e, "aqua_reset_f32 launch");
}

int aqua_rollout_f32(const AquaParams* p, const void* obst_blob_dev, int K, int64_t N, int64_t env_offset,
                     float* state, int64_t ld, int32_t* time, int64_t T, const void* actions, int action_kind,
                     int64_t action_ld, int64_t action_step_stride, uint64_t seed, uint64_t tick,
                     const uint64_t* tick_base_dev, float* reward, uint8_t* term, int64_t out_step_stride,
                     uint64_t* done_bits, int64_t done_step_stride, float* obs_norm, int auto_reset, int advance_tick,
                     void* stream)
{
    return aqua_rollout_events_f32(p, obst_blob_dev, K, N, env_offset, state, ld, time, T, actions, action_kind, action_ld,
                                   action_step_stride, seed, tick, tick_base_dev, reward, term, out_step_stride, done_bits,
                                   done_step_stride, obs_norm, auto_reset, advance_tick, nullptr, nullptr, stream);
}

int aqua_rollout_events_f32(const AquaParams* p, const void* obst_blob_dev, int K, int64_t N, int64_t env_offset,
                            float* state, int64_t ld, int32_t* time, int64_t T, const void* actions, int action_kind,
                            int64_t action_ld, int64_t action_step_stride, uint64_t seed, uint64_t tick,
                            const uint64_t* tick_base_dev, float* reward, uint8_t* term, int64_t out_step_stride,
                            uint64_t* done_bits, int64_t done_step_stride, float* obs_norm, int auto_reset, int advance_tick,
                            AquaEvent* first_start, AquaEvent* last_stop, void* stream)
{
    StepArgs a;
    int rc = fill_args(a, p, obst_blob_dev, K, N, env_offset, state, ld, time, seed, tick, tick_base_dev);
    if (rc) return rc;
    rc = check_step_buffers(N, actions, action_kind, action_ld, nullptr, 0, reward, term);
    if (rc) return rc;
    if (T < 0 || action_step_stride < 0 || out_step_stride < 0 || done_step_stride < 0)
        return fail(AQUA_E_INVALID, "negative T or stride");
    if (auto_reset < 0 || auto_reset > 2) return fail(AQUA_E_INVALID, "auto_reset must be 0, 1 or 2");
    if (advance_tick && tick_base_dev == nullptr) return fail(AQUA_E_INVALID, "advance_tick needs tick_base_dev");
    if (N == 0 || T == 0) return 0;
    a.action_ld = action_ld; a.auto_reset = auto_reset; a.obs_norm = obs_norm;
    const size_t esz = action_elem_bytes(action_kind);
    uint64_t* const tick_words = const_cast<uint64_t*>(tick_base_dev);       // [0] the base, [1] scratch (advance_tick)
    for (int64_t t = 0; t < T; ++t) {
        a.tick = tick + static_cast<uint64_t>(t);
        if (advance_tick && T >= 2) {                    // see tick_housekeeping()
            a.tick_base = (t == T - 1) ? tick_words + 1 : tick_words;
            a.tick_copy_to = (t == 0) ? tick_words + 1 : nullptr;
            a.tick_bump_to = (t == T - 1) ? tick_words : nullptr;
            a.tick_bump = static_cast<uint64_t>(T);
        }
        a.action = actions ? static_cast<const char*>(actions) + static_cast<size_t>(t * action_step_stride) * esz : nullptr;
        a.reward = reward + t * out_step_stride;
        a.term = term + t * out_step_stride;
        a.done_bits = done_bits ? done_bits + t * done_step_stride : nullptr;
        LaunchEvents ev;
        if (t == 0 && first_start != nullptr) ev.start = first_start->event;
        if (t == T - 1 && last_stop != nullptr) ev.stop = last_stop->event;
        const hipError_t e = launch_step_any(a, action_kind, static_cast<hipStream_t>(stream), ev);
        if (e != hipSuccess) return hip_fail(e, "aqua_rollout_f32 launch");
    }
    if (advance_tick && T == 1) return aqua_tick_advance(tick_words, 1, stream);   // one launch cannot do both halves
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
    if (auto_reset < 0 || auto_reset > 2) return fail(AQUA_E_INVALID, "auto_reset must be 0, 1 or 2");
    if (N == 0 || T == 0) return 0;
    a.action = actions; a.action_ld = action_ld; a.action_step_stride = action_step_stride;
    a.reward = reward; a.term = term; a.out_step_stride = out_step_stride; a.T = T; a.auto_reset = auto_reset;
    const dim3 grid(grid_for(N, BLOCK_SMALL, 2048)), block(rollout_threads(auto_reset));
    hipStream_t s = static_cast<hipStream_t>(stream);
    const bool small = K > 0 && K <= QUICK_MAX;
#define AQUA_ROLLOUT_MODE(AK, SM)                                                                                         \
    do {                                                                                                                  \
        if (auto_reset == AQUA_RESET_NEXT_STEP) hipLaunchKernelGGL((rollout_kernel<AK, SM, AQUA_RESET_NEXT_STEP>), grid, block, 0, s, a); \
        else if (auto_reset == AQUA_RESET_SAME_STEP) hipLaunchKernelGGL((rollout_kernel<AK, SM, AQUA_RESET_SAME_STEP>), grid, block, 0, s, a); \
        else hipLaunchKernelGGL((rollout_kernel<AK, SM, 0>), grid, block, 0, s, a);                                       \
    } while (0)
#define AQUA_ROLLOUT_LAUNCH(AK)                                                                      \
    case AK:                                                                                         \
        if (small) AQUA_ROLLOUT_MODE(AK, true);                                                      \
        else AQUA_ROLLOUT_MODE(AK, false);                                                           \
        break;
    switch (action_kind) {
        AQUA_ROLLOUT_LAUNCH(AQUA_ACT_U8)
        AQUA_DEV_OTHER_KINDS(AQUA_ROLLOUT_LAUNCH)
#undef AQUA_ROLLOUT_LAUNCH
#undef AQUA_ROLLOUT_MODE
        default: return fail(AQUA_E_INVALID, "unknown action_kind %d", action_kind);
    }
    const hipError_t e = hipGetLastError();
    return e == hipSuccess ? 0 : hip_fail(e, "aqua_rollout_fused_f32 launch");
}

size_t aqua_tables32_floats(int K, int64_t tld)
{
    return (K < 1 || tld < 0) ? 0 : static_cast<size_t>(12) * static_cast<size_t>(K) * static_cast<size_t>(tld);
}

int aqua_pack_tables(const double* rows, int K, int64_t N, int64_t tld, float* tab32_host, double* tab64_host, float* r_max_out)
{
    if (K < 1 || K > AQUA_MAX_OBSTACLES) return fail(AQUA_E_INVALID, "K=%d outside [1, %d]", K, AQUA_MAX_OBSTACLES);
    if (N < 0 || tld < N) return fail(AQUA_E_INVALID, "bad sizes: N=%lld tld=%lld", (long long)N, (long long)tld);
    if (rows == nullptr || tab32_host == nullptr || tab64_host == nullptr || r_max_out == nullptr)
        return fail(AQUA_E_INVALID, "rows/tab32/tab64/r_max is NULL");
    const auto tight = [](double R) {
        return 2.5 * (R + static_cast<double>(BAND)) * static_cast<double>(BAND_TIGHT) + 4.0 * 1.1920929e-7 * R * R;
    };
    double r_max = 2.5;                                  // a rectangle's (and the smallest possible) collision radius
    for (int64_t i = 0; i < N; ++i)
        for (int k = 0; k < K; ++k) {
            const double* o = rows + (i * K + k) * 5;
            if (o[2] < 0.0) continue;
            if (!(o[2] == 0.0 || o[2] == 1.0)) return fail(AQUA_E_INVALID, "world %lld obstacle %d: kind must be 0, 1 or < 0", (long long)i, k);
            if (o[3] < 0.0 || (o[2] == 1.0 && o[4] < 0.0)) return fail(AQUA_E_INVALID, "world %lld obstacle %d: negative size", (long long)i, k);
            if (o[2] == 0.0 && 2.5 + o[3] > r_max) r_max = 2.5 + o[3];
        }
    const float b_max = static_cast<float>(tight(r_max));
    for (int64_t i = 0; i < N; ++i)
        for (int k = 0; k < K; ++k) {
            const double* o = rows + (i * K + k) * 5;
            float* f = tab32_host + (6 * k) * tld + i;
            double* d = tab64_host + (5 * k) * tld + i;
            for (int c = 0; c < 5; ++c) d[c * tld] = o[c];
            if (o[2] < 0.0) {                            // absent: never hit, never in band
                f[0] = f[tld] = f[2 * tld] = f[3 * tld] = 0.0f; f[4 * tld] = -3.0e38f; f[5 * tld] = 1.0f;
                continue;
            }
            double hx = 0, hy = 0, R = 2.5;
            if (o[2] == 0.0) R += o[3]; else { hx = o[3] / 2; hy = o[4] / 2; }
            f[0] = static_cast<float>(o[0]); f[tld] = static_cast<float>(o[1]);
            f[2 * tld] = static_cast<float>(hx); f[3 * tld] = static_cast<float>(hy);
            const float r2 = static_cast<float>(R * R);
            f[4 * tld] = r2;
            f[5 * tld] = static_cast<float>(static_cast<double>(b_max) / tight(std::sqrt(static_cast<double>(r2))));
        }
    // the world-major copy behind the struct of arrays: [tld][K][6], a world's table contiguous (tables_world_major())
    float* const aos = tab32_host + static_cast<size_t>(6) * K * tld;
    for (int64_t i = 0; i < N; ++i)
        for (int k = 0; k < K; ++k)
            for (int c = 0; c < 6; ++c) aos[(i * K + k) * 6 + c] = tab32_host[(6 * k + c) * tld + i];
    *r_max_out = static_cast<float>(r_max);
    return 0;
}

namespace {
int fill_table_args(StepArgs& a, const AquaParams* p, const float* tab32, int K, int64_t tld, int64_t N, int64_t env_offset,
                    float* state, int64_t ld, int32_t* time, uint64_t seed, uint64_t tick, const uint64_t* tick_base)
{
    if (K < 1) return fail(AQUA_E_INVALID, "K=%d: per-world tables need at least one row", K);
    if (tab32 == nullptr) return fail(AQUA_E_INVALID, "tab32 is NULL");
    if (tld < N) return fail(AQUA_E_INVALID, "tld=%lld < N=%lld", (long long)tld, (long long)N);
    if (!aligned(tab32, 4)) return fail(AQUA_E_ALIGN, "tab32 must be 4-byte aligned");
    const int rc = fill_args(a, p, nullptr, 0, N, env_offset, state, ld, time, seed, tick, tick_base);
    if (rc) return rc;
    a.K = K;
    return 0;
}
}  // namespace

namespace {
struct TableArgs {
    const float* t32;
    const double* t64;
    int64_t tld;
    float band2, band2_tight;
};

hipError_t launch_step_tables(const StepArgs& a0, const TableArgs& t, int kind, hipStream_t s)
{
    StepArgs a = a0;
    // Tables of 11..16 rows with restarts take the 16-row instantiation (rows in registers, handed over through LDS like the
    // eight-row one): us per step at 262 144 worlds, rows fetched again by the re-seeding groups -> handed over
    // (profiles/r03/tables_kreg16.txt): 11 rows 18.9 -> 17.9, 12: 20.2 -> 18.6, 13: 21.9 -> 19.4, 14: 22.5 -> 20.4, 16: 24.7 ->
    // 22.2 (same-step 26.0 -> 21.5).  It loads 16 rows whatever the table holds: 9 and 10 rows lose 2-4 %, and without
    // restarts it is within 3 % either way -- both stay on the kernels that read the rows as they go.
    const bool wide = a.K >= TABLES_KREG_WIDE_MIN && a.K <= TABLES_KREG_WIDE && a.auto_reset != 0;
    const bool regs = a.K <= TABLES_KREG || wide;       // the rows fit the lanes' registers (see tables_step_block)
    // next-step restart: tables whose rows are in registers restart inside the tile (rows handed over through LDS); the
    // others keep the launch split by role (their re-seeding groups read the rows from memory)
    const bool ns_tile = a.auto_reset == AQUA_RESET_NEXT_STEP && regs;
#ifdef AQUA_TABLES_ROLE_SPLIT                 // (A/B timing: round 4's launch split by role for the long tables)
    const bool ns_sink = false;
#else
    const bool ns_sink = a.auto_reset == AQUA_RESET_NEXT_STEP && !regs;     // rows handed over as they are streamed
#endif
    const bool ns = a.auto_reset == AQUA_RESET_NEXT_STEP && !regs && !ns_sink;
    const bool interleave = ns && a.N >= NS_INTERLEAVE_MIN;
    int64_t blocks = (a.N + BLOCK_SMALL - 1) / BLOCK_SMALL;
    if (ns) {
        a.reseed_blocks = (a.N + NS_SCAN - 1) / NS_SCAN;
        blocks = interleave ? (a.reseed_blocks + 7) / 8 * 8 * (NS_SCAN / NS_TILE + 1) : blocks + a.reseed_blocks;
    }
    if (blocks > MAX_GRID) return hipErrorInvalidValue;
    const dim3 grid(static_cast<unsigned>(blocks)), block(BLOCK_SMALL);
#define AQUA_TAB_ARGS grid, block, 0, s, a, t.t32, t.t64, t.tld, t.band2, t.band2_tight
#define AQUA_TAB_LAUNCH(AK)                                                                                        \
    case AK:                                                                                                        \
        if (ns_tile && wide) hipLaunchKernelGGL((step_tables_kernel<AK, TABLES_NEXT_STEP_TILE, TABLES_KREG_WIDE>), AQUA_TAB_ARGS); \
        else if (ns_tile) hipLaunchKernelGGL((step_tables_kernel<AK, TABLES_NEXT_STEP_TILE, TABLES_KREG>), AQUA_TAB_ARGS);   \
        else if (ns_sink && a.K >= SINK_SPLIT_LONG_MIN_ROWS) hipLaunchKernelGGL((step_tables_kernel<AK, TABLES_NEXT_STEP_TILE, 0, SINK_SPLIT_LONG>), AQUA_TAB_ARGS); \
        else if (ns_sink) hipLaunchKernelGGL((step_tables_kernel<AK, TABLES_NEXT_STEP_TILE, 0>), AQUA_TAB_ARGS);             \
        else if (regs && wide && a.auto_reset) hipLaunchKernelGGL((step_tables_kernel<AK, AQUA_RESET_SAME_STEP, TABLES_KREG_WIDE>), AQUA_TAB_ARGS); \
        else if (regs && a.auto_reset) hipLaunchKernelGGL((step_tables_kernel<AK, AQUA_RESET_SAME_STEP, TABLES_KREG>), AQUA_TAB_ARGS); \
        else if (regs) hipLaunchKernelGGL((step_tables_kernel<AK, AQUA_RESET_NONE, TABLES_KREG>), AQUA_TAB_ARGS);            \
        else if (interleave) hipLaunchKernelGGL((step_tables_ns_kernel<AK, true>), AQUA_TAB_ARGS);                           \
        else if (ns) hipLaunchKernelGGL((step_tables_ns_kernel<AK, false>), AQUA_TAB_ARGS);                                  \
        else if (a.auto_reset && a.K >= SINK_SPLIT_LONG_MIN_ROWS) hipLaunchKernelGGL((step_tables_kernel<AK, AQUA_RESET_SAME_STEP, 0, SINK_SPLIT_LONG>), AQUA_TAB_ARGS); \
        else if (a.auto_reset && a.K > TABLES_KREG_WIDE) hipLaunchKernelGGL((step_tables_kernel<AK, AQUA_RESET_SAME_STEP, 0>), AQUA_TAB_ARGS); \
        else if (a.auto_reset) hipLaunchKernelGGL((step_tables_kernel<AK, AQUA_RESET_SAME_STEP, 0, 1>), AQUA_TAB_ARGS);      \
        else hipLaunchKernelGGL((step_tables_kernel<AK, AQUA_RESET_NONE, 0>), AQUA_TAB_ARGS);                                \
        break;
    switch (kind) {
        AQUA_TAB_LAUNCH(AQUA_ACT_U8)
        AQUA_DEV_OTHER_KINDS(AQUA_TAB_LAUNCH)
        default: return hipErrorInvalidValue;
    }
#undef AQUA_TAB_LAUNCH
#undef AQUA_TAB_ARGS
    return hipGetLastError();
}

int check_tables(TableArgs& t, const float* tab32_dev, const double* tab64_dev, int64_t tld, float r_max)
{
    if (tab64_dev == nullptr || !aligned(tab64_dev, 8)) return fail(AQUA_E_INVALID, "tab64 is NULL or not 8-byte aligned");
    if (!(r_max >= 2.5f)) return fail(AQUA_E_INVALID, "r_max=%g: use the value aqua_pack_tables returned", (double)r_max);
    const double R = static_cast<double>(r_max);
    t.t32 = tab32_dev; t.t64 = tab64_dev; t.tld = tld;
    t.band2 = static_cast<float>(2.5 * (R + static_cast<double>(BAND)) * static_cast<double>(BAND));
    t.band2_tight = static_cast<float>(2.5 * (R + static_cast<double>(BAND)) * static_cast<double>(BAND_TIGHT) +
                                       4.0 * 1.1920929e-7 * R * R);
    return 0;
}
}  // namespace

int aqua_step_tables_f32(const AquaParams* p, const float* tab32_dev, const double* tab64_dev, int K, int64_t tld,
                         float r_max, int64_t N, int64_t env_offset, float* state, int64_t ld, int32_t* time,
                         const void* action, int action_kind, int64_t action_ld, const float* noise, int64_t noise_ld,
                         uint64_t seed, uint64_t tick, const uint64_t* tick_base_dev, float* reward, uint8_t* term,
                         uint64_t* done_bits, float* obs_norm, int auto_reset, void* stream)
{
    StepArgs a;
    if (auto_reset < 0 || auto_reset > 2) return fail(AQUA_E_INVALID, "auto_reset must be 0, 1 or 2");
    int rc = fill_table_args(a, p, tab32_dev, K, tld, N, env_offset, state, ld, time, seed, tick, tick_base_dev);
    if (rc) return rc;
    TableArgs t;
    rc = check_tables(t, tab32_dev, tab64_dev, tld, r_max);
    if (rc) return rc;
    rc = check_step_buffers(N, action, action_kind, action_ld, noise, noise_ld, reward, term);
    if (rc) return rc;
    if (done_bits != nullptr && !aligned(done_bits, 8)) return fail(AQUA_E_ALIGN, "done_bits must be 8-byte aligned");
    if (N == 0) return 0;
    a.action = action; a.action_ld = action_ld; a.noise = noise; a.noise_ld = noise_ld;
    a.reward = reward; a.term = term; a.done_bits = done_bits; a.obs_norm = obs_norm; a.auto_reset = auto_reset;
    const hipError_t e = launch_step_tables(a, t, action_kind, static_cast<hipStream_t>(stream));
    return e == hipSuccess ? 0 : hip_fail(e, "aqua_step_tables_f32 launch");
}

int aqua_rollout_tables_f32(const AquaParams* p, const float* tab32_dev, const double* tab64_dev, int K, int64_t tld,
                            float r_max, int64_t N, int64_t env_offset, float* state, int64_t ld, int32_t* time, int64_t T,
                            const void* actions, int action_kind, int64_t action_ld, int64_t action_step_stride,
                            uint64_t seed, uint64_t tick, const uint64_t* tick_base_dev, float* reward, uint8_t* term,
                            int64_t out_step_stride, uint64_t* done_bits, int64_t done_step_stride, float* obs_norm,
                            int auto_reset, int advance_tick, void* stream)
{
    StepArgs a;
    if (auto_reset < 0 || auto_reset > 2) return fail(AQUA_E_INVALID, "auto_reset must be 0, 1 or 2");
    int rc = fill_table_args(a, p, tab32_dev, K, tld, N, env_offset, state, ld, time, seed, tick, tick_base_dev);
    if (rc) return rc;
    TableArgs t;
    rc = check_tables(t, tab32_dev, tab64_dev, tld, r_max);
    if (rc) return rc;
    rc = check_step_buffers(N, actions, action_kind, action_ld, nullptr, 0, reward, term);
    if (rc) return rc;
    if (T < 0 || action_step_stride < 0 || out_step_stride < 0 || done_step_stride < 0)
        return fail(AQUA_E_INVALID, "negative T or stride");
    if (advance_tick && tick_base_dev == nullptr) return fail(AQUA_E_INVALID, "advance_tick needs tick_base_dev");
    if (N == 0 || T == 0) return 0;
    a.action_ld = action_ld; a.auto_reset = auto_reset; a.obs_norm = obs_norm;
    const size_t esz = action_elem_bytes(action_kind);
    uint64_t* const tick_words = const_cast<uint64_t*>(tick_base_dev);
    for (int64_t s = 0; s < T; ++s) {
        a.tick = tick + static_cast<uint64_t>(s);
        if (advance_tick && T >= 2) {                    // see tick_housekeeping()
            a.tick_base = (s == T - 1) ? tick_words + 1 : tick_words;
            a.tick_copy_to = (s == 0) ? tick_words + 1 : nullptr;
            a.tick_bump_to = (s == T - 1) ? tick_words : nullptr;
            a.tick_bump = static_cast<uint64_t>(T);
        }
        a.action = actions ? static_cast<const char*>(actions) + static_cast<size_t>(s * action_step_stride) * esz : nullptr;
        a.reward = reward + s * out_step_stride;
        a.term = term + s * out_step_stride;
        a.done_bits = done_bits ? done_bits + s * done_step_stride : nullptr;
        const hipError_t e = launch_step_tables(a, t, action_kind, static_cast<hipStream_t>(stream));
        if (e != hipSuccess) return hip_fail(e, "aqua_rollout_tables_f32 launch");
    }
    if (advance_tick && T == 1) return aqua_tick_advance(tick_words, 1, stream);
    return 0;
}

int aqua_rollout_tables_fused_f32(const AquaParams* p, const float* tab32_dev, const double* tab64_dev, int K, int64_t tld,
                                  float r_max, int64_t N, int64_t env_offset, float* state, int64_t ld, int32_t* time, int64_t T,
                                  const void* actions, int action_kind, int64_t action_ld, int64_t action_step_stride,
                                  uint64_t seed, uint64_t tick, const uint64_t* tick_base_dev, float* reward, uint8_t* term,
                                  int64_t out_step_stride, int auto_reset, void* stream)
{
    StepArgs a;
    if (auto_reset < 0 || auto_reset > 2) return fail(AQUA_E_INVALID, "auto_reset must be 0, 1 or 2");
    if (K > FUSED_TABLE_ROWS_MAX)
        return fail(AQUA_E_INVALID, "the fused per-world rollout keeps tables of at most %d rows in LDS (K=%d): use aqua_rollout_tables_f32", FUSED_TABLE_ROWS_MAX, K);
    const bool wide = K > TABLES_KREG;                   // 9..16 rows: the 80 KB tile
    int rc = fill_table_args(a, p, tab32_dev, K, tld, N, env_offset, state, ld, time, seed, tick, tick_base_dev);
    if (rc) return rc;
    TableArgs t;
    rc = check_tables(t, tab32_dev, tab64_dev, tld, r_max);
    if (rc) return rc;
    rc = check_step_buffers(N, actions, action_kind, action_ld, nullptr, 0, reward, term);
    if (rc) return rc;
    if (T < 0 || action_step_stride < 0 || out_step_stride < 0) return fail(AQUA_E_INVALID, "negative T or stride");
    if (N == 0 || T == 0) return 0;
    a.action = actions; a.action_ld = action_ld; a.action_step_stride = action_step_stride;
    a.reward = reward; a.term = term; a.out_step_stride = out_step_stride; a.T = T; a.auto_reset = auto_reset;
    const int worlds_per_block = K <= 16 ? BLOCK_SMALL : (K <= 32 ? 128 : 64);
    const dim3 grid(grid_for(N, worlds_per_block, 4096)), block(BLOCK_SMALL);
    hipStream_t s = static_cast<hipStream_t>(stream);
#define AQUA_ROLLOUT_TABLES_MODE(KERNEL, AK)                                                                                 \
        if (auto_reset == AQUA_RESET_NEXT_STEP)                                                                              \
            hipLaunchKernelGGL((KERNEL<AK, AQUA_RESET_NEXT_STEP>), grid, block, 0, s, a, t.t32, t.t64, t.tld, t.band2, t.band2_tight); \
        else if (auto_reset == AQUA_RESET_SAME_STEP)                                                                         \
            hipLaunchKernelGGL((KERNEL<AK, AQUA_RESET_SAME_STEP>), grid, block, 0, s, a, t.t32, t.t64, t.tld, t.band2, t.band2_tight); \
        else hipLaunchKernelGGL((KERNEL<AK, 0>), grid, block, 0, s, a, t.t32, t.t64, t.tld, t.band2, t.band2_tight);
#define AQUA_ROLLOUT_TABLES(AK)                                                                                              \
    case AK:                                                                                                                 \
        if (K > 32) { AQUA_ROLLOUT_TABLES_MODE(rollout_tables64_kernel, AK) }                                                \
        else if (K > 16) { AQUA_ROLLOUT_TABLES_MODE(rollout_tables32_kernel, AK) }                                           \
        else if (wide) { AQUA_ROLLOUT_TABLES_MODE(rollout_tables16_kernel, AK) }                                             \
        else { AQUA_ROLLOUT_TABLES_MODE(rollout_tables_kernel, AK) }                                                         \
        break;
    switch (action_kind) {
        AQUA_ROLLOUT_TABLES(AQUA_ACT_U8)
        AQUA_DEV_OTHER_KINDS(AQUA_ROLLOUT_TABLES)
#undef AQUA_ROLLOUT_TABLES
#undef AQUA_ROLLOUT_TABLES_MODE
        default: return fail(AQUA_E_INVALID, "unknown action_kind %d", action_kind);
    }
    const hipError_t e = hipGetLastError();
    return e == hipSuccess ? 0 : hip_fail(e, "aqua_rollout_tables_fused_f32 launch");
}

int aqua_reset_tables_f32(const AquaParams* p, const float* tab32_dev, int K, int64_t tld, int64_t N, int64_t env_offset,
                          float* state, int64_t ld, int32_t* time, const uint8_t* mask, uint64_t seed, uint64_t tick,
                          const uint64_t* tick_base_dev, void* stream)
{
    StepArgs a;
    const int rc = fill_table_args(a, p, tab32_dev, K, tld, N, env_offset, state, ld, time, seed, tick, tick_base_dev);
    if (rc) return rc;
    if (N == 0) return 0;
    if ((N + RESET_SCAN - 1) / RESET_SCAN > MAX_GRID) return fail(AQUA_E_INVALID, "N=%lld too large for one launch", (long long)N);
    hipLaunchKernelGGL(reset_tables_kernel, dim3(static_cast<unsigned>((N + RESET_SCAN - 1) / RESET_SCAN)), dim3(BLOCK_SMALL), 0,
                       static_cast<hipStream_t>(stream), a, mask, tab32_dev, tld);
    const hipError_t e = hipGetLastError();
    return e == hipSuccess ? 0 : hip_fail(e, "aqua_reset_tables_f32 launch");
}

int aqua_obs_norm_f32(const float* state, int64_t ld, int64_t N, const uint8_t* mask, float* obs_norm, void* stream)
{
    if (N < 0 || ld < N) return fail(AQUA_E_INVALID, "bad sizes: N=%lld ld=%lld", (long long)N, (long long)ld);
    if (N == 0) return 0;
    if (state == nullptr || obs_norm == nullptr) return fail(AQUA_E_INVALID, "state/obs_norm is NULL");
    if (!aligned(state, 4) || !aligned(obs_norm, 4)) return fail(AQUA_E_ALIGN, "state/obs_norm must be 4-byte aligned");
    hipLaunchKernelGGL(obs_norm_kernel, dim3(grid_for(N, BLOCK_SMALL, 2048)), dim3(BLOCK_SMALL), 0,
                       static_cast<hipStream_t>(stream), state, ld, N, mask, obs_norm);
    const hipError_t e = hipGetLastError();
    return e == hipSuccess ? 0 : hip_fail(e, "aqua_obs_norm_f32 launch");
}

int aqua_ring_write_f32(float* ring, int64_t ring_ld, int64_t capacity, int64_t cursor, const float* src, int64_t src_ld,
                        int rows, int64_t N, void* stream)
{
    return ring_write<float>(ring, ring_ld, capacity, cursor, src, src_ld, rows, N, stream);
}

int aqua_ring_write_u8(uint8_t* ring, int64_t ring_ld, int64_t capacity, int64_t cursor, const uint8_t* src,
                       int64_t src_ld, int rows, int64_t N, void* stream)
{
    return ring_write<uint8_t>(ring, ring_ld, capacity, cursor, src, src_ld, rows, N, stream);
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

int aqua_graph_upload(AquaGraph* g, void* stream)
{
    if (g == nullptr) return fail(AQUA_E_INVALID, "graph is NULL");
    const hipError_t e = hipGraphUpload(g->exec, static_cast<hipStream_t>(stream));
    return e == hipSuccess ? 0 : hip_fail(e, "hipGraphUpload");
}

int aqua_graph_destroy(AquaGraph* g)
{
    if (g == nullptr) return 0;
    (void)hipGraphExecDestroy(g->exec);
    (void)hipGraphDestroy(g->graph);
    delete g;
    return 0;
}

int aqua_event_create(AquaEvent** out)
{
    if (out == nullptr) return fail(AQUA_E_INVALID, "out is NULL");
    hipEvent_t e = nullptr;
    const hipError_t rc = hipEventCreateWithFlags(&e, hipEventDefault);
    if (rc != hipSuccess) return hip_fail(rc, "hipEventCreateWithFlags");
    *out = new AquaEvent{e};
    return 0;
}

int aqua_event_record(AquaEvent* e, void* stream)
{
    if (e == nullptr) return fail(AQUA_E_INVALID, "event is NULL");
    const hipError_t rc = hipEventRecord(e->event, static_cast<hipStream_t>(stream));
    return rc == hipSuccess ? 0 : hip_fail(rc, "hipEventRecord");
}

int aqua_graph_end_timed(void* stream, AquaGraph** out, AquaEvent* start, AquaEvent* stop)
{
    if (out == nullptr || start == nullptr || stop == nullptr) return fail(AQUA_E_INVALID, "out/start/stop is NULL");
    hipGraph_t g = nullptr;
    hipError_t e = hipStreamEndCapture(static_cast<hipStream_t>(stream), &g);
    if (e != hipSuccess) return hip_fail(e, "hipStreamEndCapture");
    // event-record nodes around what was captured: `start` ahead of every root, `stop` behind every leaf
    const auto bail = [&](hipError_t err, const char* what) { (void)hipGraphDestroy(g); return hip_fail(err, what); };
    size_t n_nodes = 0, n_roots = 0;
    if ((e = hipGraphGetNodes(g, nullptr, &n_nodes)) != hipSuccess) return bail(e, "hipGraphGetNodes");
    std::vector<hipGraphNode_t> nodes(n_nodes);
    if (n_nodes && (e = hipGraphGetNodes(g, nodes.data(), &n_nodes)) != hipSuccess) return bail(e, "hipGraphGetNodes");
    if ((e = hipGraphGetRootNodes(g, nullptr, &n_roots)) != hipSuccess) return bail(e, "hipGraphGetRootNodes");
    std::vector<hipGraphNode_t> roots(n_roots), leaves;
    if (n_roots && (e = hipGraphGetRootNodes(g, roots.data(), &n_roots)) != hipSuccess) return bail(e, "hipGraphGetRootNodes");
    for (hipGraphNode_t node : nodes) {
        size_t n_dep = 0;
        if ((e = hipGraphNodeGetDependentNodes(node, nullptr, &n_dep)) != hipSuccess) return bail(e, "hipGraphNodeGetDependentNodes");
        if (n_dep == 0) leaves.push_back(node);
    }
    hipGraphNode_t first = nullptr, last = nullptr;
    if ((e = hipGraphAddEventRecordNode(&first, g, nullptr, 0, start->event)) != hipSuccess) return bail(e, "hipGraphAddEventRecordNode");
    for (hipGraphNode_t root : roots)
        if ((e = hipGraphAddDependencies(g, &first, &root, 1)) != hipSuccess) return bail(e, "hipGraphAddDependencies");
    if ((e = hipGraphAddEventRecordNode(&last, g, leaves.data(), leaves.size(), stop->event)) != hipSuccess)
        return bail(e, "hipGraphAddEventRecordNode");
    hipGraphExec_t x = nullptr;
    if ((e = hipGraphInstantiate(&x, g, nullptr, nullptr, 0)) != hipSuccess) return bail(e, "hipGraphInstantiate");
    *out = new AquaGraph{g, x};
    return 0;
}

int aqua_event_elapsed_ms(AquaEvent* start, AquaEvent* stop, float* ms)
{
    if (start == nullptr || stop == nullptr || ms == nullptr) return fail(AQUA_E_INVALID, "event/ms is NULL");
    const hipError_t rc = hipEventElapsedTime(ms, start->event, stop->event);
    return rc == hipSuccess ? 0 : hip_fail(rc, "hipEventElapsedTime");
}

int aqua_event_destroy(AquaEvent* e)
{
    if (e == nullptr) return 0;
    (void)hipEventDestroy(e->event);
    delete e;
    return 0;
}

// ------------------------------------------------------------------ buffers other processes of the node can write (done-mask exchange)
struct AquaIpcBuffer {
    void* ptr;
    size_t bytes;
};

int aqua_ipc_buffer_create(size_t bytes, AquaIpcBuffer** out)
{
    if (out == nullptr || bytes == 0) return fail(AQUA_E_INVALID, "out is NULL or bytes == 0");
    void* p = nullptr;
    hipError_t e = hipMalloc(&p, bytes);
    if (e != hipSuccess) return hip_fail(e, "hipMalloc (ipc buffer)");
    if ((e = hipMemset(p, 0, bytes)) != hipSuccess) { (void)hipFree(p); return hip_fail(e, "hipMemset (ipc buffer)"); }
    // the handle is exported next and other PROCESSES then write through streams nothing orders behind this memset on the
    // null stream: it must have run before anybody can know the buffer (set-up code, not the hot path)
    if ((e = hipStreamSynchronize(nullptr)) != hipSuccess) { (void)hipFree(p); return hip_fail(e, "hipStreamSynchronize (ipc buffer)"); }
    *out = new AquaIpcBuffer{p, bytes};
    return 0;
}

void* aqua_ipc_buffer_ptr(AquaIpcBuffer* b) { return b ? b->ptr : nullptr; }

int aqua_ipc_buffer_handle(AquaIpcBuffer* b, unsigned char handle[AQUA_IPC_HANDLE_BYTES])
{
    static_assert(sizeof(hipIpcMemHandle_t) == AQUA_IPC_HANDLE_BYTES, "hipIpcMemHandle_t size");
    if (b == nullptr || handle == nullptr) return fail(AQUA_E_INVALID, "buffer/handle is NULL");
    hipIpcMemHandle_t h;
    const hipError_t e = hipIpcGetMemHandle(&h, b->ptr);
    if (e != hipSuccess) return hip_fail(e, "hipIpcGetMemHandle");
    std::memcpy(handle, &h, sizeof(h));
    return 0;
}

int aqua_ipc_buffer_destroy(AquaIpcBuffer* b)
{
    if (b == nullptr) return 0;
    (void)hipFree(b->ptr);
    delete b;
    return 0;
}

int aqua_ipc_open(const unsigned char handle[AQUA_IPC_HANDLE_BYTES], void** peer_ptr)
{
    if (handle == nullptr || peer_ptr == nullptr) return fail(AQUA_E_INVALID, "handle/peer_ptr is NULL");
    hipIpcMemHandle_t h;
    std::memcpy(&h, handle, sizeof(h));
    const hipError_t e = hipIpcOpenMemHandle(peer_ptr, h, hipIpcMemLazyEnablePeerAccess);
    return e == hipSuccess ? 0 : hip_fail(e, "hipIpcOpenMemHandle");
}

int aqua_ipc_close(void* peer_ptr)
{
    if (peer_ptr == nullptr) return 0;
    const hipError_t e = hipIpcCloseMemHandle(peer_ptr);
    return e == hipSuccess ? 0 : hip_fail(e, "hipIpcCloseMemHandle");
}

int aqua_copy_fanout_async(void* const* dsts, int n_dst, const void* src, size_t bytes, void* stream)
{
    if (bytes == 0 || n_dst == 0) return 0;
    if (dsts == nullptr || src == nullptr || n_dst < 0) return fail(AQUA_E_INVALID, "dsts/src is NULL or n_dst < 0");
    if (bytes % 8 != 0 || !aligned(src, 8)) return fail(AQUA_E_ALIGN, "aqua_copy_fanout_async moves whole 8-byte words");
    bool wide = bytes % 16 == 0 && aligned(src, 16);
    for (int j = 0; j < n_dst; ++j) {
        if (dsts[j] == nullptr) return fail(AQUA_E_INVALID, "destination %d is NULL", j);
        if (!aligned(dsts[j], 8)) return fail(AQUA_E_ALIGN, "destination %d is not 8-byte aligned", j);
        wide = wide && aligned(dsts[j], 16);
    }
    size_t blocks = (bytes + COPY_BYTES_PER_BLOCK - 1) / COPY_BYTES_PER_BLOCK;
    blocks = blocks < COPY_BLOCKS_MIN ? COPY_BLOCKS_MIN : (blocks > COPY_BLOCKS_MAX ? COPY_BLOCKS_MAX : blocks);
    for (int first = 0; first < n_dst; first += COPY_FANOUT_MAX) {
        CopyFanout f;
        const int m = n_dst - first < COPY_FANOUT_MAX ? n_dst - first : COPY_FANOUT_MAX;
        for (int j = 0; j < COPY_FANOUT_MAX; ++j) f.dst[j] = dsts[first + (j < m ? j : 0)];
        const dim3 grid(static_cast<unsigned>(blocks), static_cast<unsigned>(m));
        if (wide)
            hipLaunchKernelGGL((copy_fanout_kernel<Word16>), grid, dim3(64), 0, static_cast<hipStream_t>(stream), f,
                               static_cast<const Word16*>(src), bytes / 16);
        else
            hipLaunchKernelGGL((copy_fanout_kernel<unsigned long long>), grid, dim3(64), 0, static_cast<hipStream_t>(stream), f,
                               static_cast<const unsigned long long*>(src), bytes / 8);
        const hipError_t e = hipGetLastError();
        if (e != hipSuccess) return hip_fail(e, "aqua_copy_fanout_async launch");
    }
    return 0;
}

int aqua_copy_async(void* dst, const void* src, size_t bytes, int engine, void* stream)
{
    if (bytes == 0) return 0;
    if (dst == nullptr || src == nullptr) return fail(AQUA_E_INVALID, "dst/src is NULL");
    if (engine == AQUA_COPY_ENGINE_DMA) {
        const hipError_t e = hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, static_cast<hipStream_t>(stream));
        return e == hipSuccess ? 0 : hip_fail(e, "hipMemcpyAsync (device to device)");
    }
    if (engine != AQUA_COPY_ENGINE_WAVES) return fail(AQUA_E_INVALID, "unknown copy engine %d", engine);
    if (bytes % 8 != 0 || !aligned(dst, 8) || !aligned(src, 8)) return fail(AQUA_E_ALIGN, "aqua_copy_async moves whole 8-byte words");
    size_t blocks = (bytes + COPY_BYTES_PER_BLOCK - 1) / COPY_BYTES_PER_BLOCK;
    blocks = blocks < COPY_BLOCKS_MIN ? COPY_BLOCKS_MIN : (blocks > COPY_BLOCKS_MAX ? COPY_BLOCKS_MAX : blocks);
    if (bytes % 16 == 0 && aligned(dst, 16) && aligned(src, 16))
        hipLaunchKernelGGL((copy_words_kernel<Word16>), dim3(static_cast<unsigned>(blocks)), dim3(64), 0, static_cast<hipStream_t>(stream),
                           static_cast<Word16*>(dst), static_cast<const Word16*>(src), bytes / 16);
    else
        hipLaunchKernelGGL((copy_words_kernel<unsigned long long>), dim3(static_cast<unsigned>(blocks)), dim3(64), 0,
                           static_cast<hipStream_t>(stream), static_cast<unsigned long long*>(dst),
                           static_cast<const unsigned long long*>(src), bytes / 8);
    const hipError_t e = hipGetLastError();
    return e == hipSuccess ? 0 : hip_fail(e, "aqua_copy_async launch");
}

}  // extern "C"
