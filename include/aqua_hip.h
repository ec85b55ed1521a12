/*
 * aqua_hip.h -- C ABI of libaqua_hip.so: the batched AquaEnv step()/reset() hot path on MI355X (gfx950).
 *
 * The reference (ilVecc/AquaticGymEnv) is pure Python and has no native layer, so there is no
 * existing FFI to mirror; every entry point below names the reference interface it replaces
 * (file:line under the reference root).  The host-side binding is ctypes
 * (aquaticgymenv_amd/_capi.py); INTEGRATION.md shows the stub a maintainer of the reference would add.
 *
 * Conventions
 *  - plain pointers and sizes only; no torch / HIP types in signatures (streams are void*).
 *  - every DEVICE buffer is owned by the caller; the library borrows the pointers for the duration
 *    of the call (asynchronously: until the work queued on `stream` has run), allocates nothing,
 *    frees nothing and keeps no pointer afterwards.  The only objects the library owns are the
 *    AquaGraph handles returned by aqua_graph_end(), the AquaEvent handles and the AquaIpcBuffer receive buffers.
 *  - launches are asynchronous on `stream`; no entry point synchronises the device, so all of
 *    them may be captured into a HIP graph.
 *  - return value: 0 = ok; > 0 = hipError_t; < 0 = AQUA_E_*.  aqua_last_error() returns a
 *    thread-local message for the last failing call on this thread.
 *  - thread safety: no global mutable state; calls on different streams may come from
 *    different host threads.
 *
 * State layout (struct of arrays, float32): `state` points at 7 rows of `ld` floats:
 *      row 0 x, 1 y, 2 theta, 3 goal_x, 4 goal_y, 5 wave_x, 6 wave_y        (ld >= N)
 * Rows 0..4 ARE the observation of aqua.py:213 (obs[:, k] = state[k*ld + i]); nothing is copied.
 * `time` is int32[N] (aqua.py:82,141).  Rows are read and written one 4-byte element per lane (coalesced 256-byte
 * wavefront accesses); 4-byte alignment is all the kernels need.
 */
#ifndef AQUA_HIP_H
#define AQUA_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define AQUA_ABI_VERSION 8   /* 2: the obstacle blob of tables of up to 8 rows ends with the quick table;
                                3: aqua_rollout_f32 takes advance_tick, timing events, aqua_graph_end_timed;
                                4: aqua_rollout_tables_fused_f32;
                                5: aqua_ipc_*, aqua_copy_async (done-mask exchange by peer copies);
                                6: aqua_rollout_events_f32 (events attached to the first / last launch);
                                7: aqua_graph_upload;
                                8: tab32 of the per-world tables carries a world-major copy behind the struct of arrays
                                   (aqua_tables32_floats) */

/* library error codes (negative) */
#define AQUA_E_INVALID   (-1)   /* bad argument (null pointer, negative size, K too large ...) */
#define AQUA_E_ALIGN     (-2)   /* pointer / leading dimension not usable */
#define AQUA_E_NODEVICE  (-3)   /* no HIP device / wrong architecture */

/* action encodings accepted by aqua_step_f32 (aqua.py:33-43,137-154) */
#define AQUA_ACT_U8        0    /* uint8[N]   discrete index 0..2 */
#define AQUA_ACT_I32       1    /* int32[N]   discrete; -3..-1 wrap like a Python list index */
#define AQUA_ACT_I64       2    /* int64[N]   discrete */
#define AQUA_ACT_F32X2     3    /* float32 [2][action_ld]: row 0 vL, row 1 vR (continuous, clipped to [0.2, 0.5]) */
#define AQUA_ACT_SAMPLE_D  4    /* no buffer: uniform discrete action from Philox stream 4 */
#define AQUA_ACT_SAMPLE_C  5    /* no buffer: vL, vR ~ U[0.2, 0.5) from Philox stream 4 */
#define AQUA_ACT_BEARING   6    /* no buffer: the hand-coded bearing policy of main/testing/test_optimal.py:8-28,
                                   evaluated on the device from the world's own observation (discrete action) */

/* termination codes written to `term` (aqua.py:194-211, exactly one info flag is True when done) */
#define AQUA_TERM_NONE     0
#define AQUA_TERM_COLLIDED 1    /* 'Termination.collided' */
#define AQUA_TERM_TIME     2    /* 'Termination.time'     */
#define AQUA_TERM_SUCCESS  3    /* 'Termination.success'  */

/* auto_reset modes of aqua_step_f32 / aqua_rollout_f32 */
#define AQUA_RESET_NONE      0  /* the reference's behaviour: no freeze, no restart (the caller resets) */
#define AQUA_RESET_SAME_STEP 1  /* a finished world is restarted inside the launch that finished it */
#define AQUA_RESET_NEXT_STEP 2  /* ... during the NEXT step, in which it does not move (reward 0, term 0) */

#define AQUA_MAX_OBSTACLES 64

/* Constructor arguments of the reference's AquaEnv (aqua.py:13-31) that change the arithmetic. */
typedef struct AquaParams {
    int32_t waves;          /* int(waves): scales wave bound 0.05 and wave step 0.001 (aqua.py:15,23-25) */
    int32_t continuous;     /* AquaContinuousEnv (aqua.py:458-459); only used by reset/sampling paths      */
    int32_t random_boat;    /* aqua.py:110-117 */
    int32_t random_goal;    /* aqua.py:102-107 */
    int32_t time_limit;     /* aqua.py:91 (1000) */
    int32_t reserved[3];
} AquaParams;

int aqua_version(void);                 /* AQUA_ABI_VERSION */
const char* aqua_last_error(void);

/*
 * Obstacle table.  Replaces the per-object Python list of aqua.py:56-68.
 * rows: host float64 [K][5] = cx, cy, kind (0 circle | 1 rectangle), a (radius | width), b (0 | height).
 * aqua_pack_obstacles() writes the device-format blob (float32 clamp boxes + squared thresholds
 * for the fast path, then the float64 rows for the exact path, then -- for K <= 8 -- the 384-byte
 * "quick table": the fast path's operands once more as struct of arrays in groups of four obstacles)
 * into HOST memory; the caller uploads it and passes the device copy to the calls below.
 * Always size the buffer with aqua_obstacle_blob_bytes(K).  K == 0 -> blob of 0 bytes, pass NULL.
 */
size_t aqua_obstacle_blob_bytes(int K);
int aqua_pack_obstacles(const double* rows, int K, void* blob_host, size_t blob_bytes);

/*
 * One batched step: replaces AquaEnv.step(action) (aqua.py:135-213) for N worlds.
 *   N, env_offset : this call advances global worlds [env_offset, env_offset + N); the Philox
 *                   stream is keyed by the GLOBAL index so any range partition gives the same result.
 *   action        : see AQUA_ACT_*; action_ld is the row stride of AQUA_ACT_F32X2 (ignored otherwise).
 *   noise         : NULL -> the two wave draws of aqua.py:188 come from Philox4x32-10
 *                   (key = seed, counter = (global env, tick)); else float32 [2][noise_ld] of
 *                   uniforms in [-1, 1) that are multiplied by 0.001*waves (injection, for parity tests).
 *   tick          : step counter of the caller; tick_base_dev (nullable, device uint64) is added to
 *                   it on the device so a captured graph can be replayed with fresh noise.
 *   reward        : float32[N]; term: uint8[N] (AQUA_TERM_*); done_bits: uint64[ceil(N/64)] or NULL,
 *                   bit (i % 64) of word (i / 64) = done flag of local world i (wavefront ballot).
 *   obs_norm      : NULL, or float32 [5][ld]: fused epilogue that also writes the new observation scaled as the
 *                   reference's AquaStateNormalizer does before the DQN sees it (main/impl/utils.py:15-33):
 *                   x/100, y/100, theta/(2 pi) + 0.5, gx/100, gy/100 (+20 B written per world-step).
 *   auto_reset    : AQUA_RESET_NONE      -> the reference's behaviour: no freeze, no reset.
 *                   AQUA_RESET_SAME_STEP -> worlds that finished are re-initialised in the same launch exactly as
 *                   aqua_reset_f32(mask = term != 0) would; reward/term/done_bits still describe the step
 *                   that finished; the observation is already the new episode's.
 *                   AQUA_RESET_NEXT_STEP -> (Gymnasium >= 1.0 convention, the fastest mode) a world that finishes
 *                   at tick t keeps its terminal state and is marked in time[]; the launch of tick t + 1
 *                   re-initialises it instead of stepping it: that tick reports the fresh observation with
 *                   reward 0 and term 0.  Negative time[] values are this bookkeeping, tagged with the parity of
 *                   the tick that wrote them (which is what lets the re-initialisation run beside the stepping
 *                   inside one launch without any synchronisation): -1 - (t & 1) "finished at tick t, awaiting
 *                   restart", -3 - (t & 1) "restarted during tick t, steps from time 0 at tick t + 1".  Ticks must
 *                   advance by one per step in this mode (a marked world otherwise waits one extra step).
 *   N             : any size the buffers hold.  The step kernels address a world by a 32-bit byte offset from the row
 *                   pointers of their launch, so the library queues a batch of more than 2^28 worlds as several launches
 *                   of that many (same stream, same results as one: a world's step depends on its global index only).
 */
int aqua_step_f32(const AquaParams* p, const void* obst_blob_dev, int K, int64_t N, int64_t env_offset,
                  float* state, int64_t ld, int32_t* time, const void* action, int action_kind,
                  int64_t action_ld, const float* noise, int64_t noise_ld, uint64_t seed, uint64_t tick,
                  const uint64_t* tick_base_dev, float* reward, uint8_t* term, uint64_t* done_bits,
                  float* obs_norm, int auto_reset, void* stream);

/*
 * Masked reset: replaces AquaEnv.reset() (aqua.py:100-126) for the worlds with mask[i] != 0
 * (mask == NULL: all N).  Rejection sampling in float32 from Philox streams 1..3 (DESIGN.md "RNG").
 */
int aqua_reset_f32(const AquaParams* p, const void* obst_blob_dev, int K, int64_t N, int64_t env_offset,
                   float* state, int64_t ld, int32_t* time, const uint8_t* mask, uint64_t seed, uint64_t tick,
                   const uint64_t* tick_base_dev, void* stream);

/*
 * T consecutive steps, one launch per step (the rollout loop of main/testing/__init__.py:25-34 and
 * main/impl/dqn.py:159-179 with the policy replaced by a pre-generated or sampled action stream).
 *   actions       : AQUA_ACT_U8/I32/I64: [T][action_step_stride elements]; AQUA_ACT_F32X2: T blocks of
 *                   [2][action_ld]; AQUA_ACT_SAMPLE_*: NULL.  action_step_stride is in ELEMENTS.
 *   reward, term  : per-step outputs, step t writes at + t*out_step_stride elements (0 = overwrite).
 *   done_bits     : likewise with done_step_stride (in uint64 words; 0 = overwrite; NULL = skip).
 *   advance_tick  : != 0 (needs tick_base_dev, which then points at TWO uint64 words: the base and a scratch word):
 *                   after the T steps the device-resident base has advanced by T, so that a captured graph of this call
 *                   draws fresh noise at every replay.  For T >= 2 the first and the last step launch do it themselves
 *                   (no extra kernel in the graph); for T == 1 it is aqua_tick_advance() behind the step.
 */
int aqua_rollout_f32(const AquaParams* p, const void* obst_blob_dev, int K, int64_t N, int64_t env_offset,
                     float* state, int64_t ld, int32_t* time, int64_t T, const void* actions, int action_kind,
                     int64_t action_ld, int64_t action_step_stride, uint64_t seed, uint64_t tick,
                     const uint64_t* tick_base_dev, float* reward, uint8_t* term, int64_t out_step_stride,
                     uint64_t* done_bits, int64_t done_step_stride, float* obs_norm, int auto_reset, int advance_tick,
                     void* stream);

/*
 * aqua_rollout_f32 with a clock on it: `first_start` (nullable) receives the START time of the first step launch,
 * `last_stop` (nullable) the END time of the last one (hipExtLaunchKernel's start / stop events: the timestamps of the
 * kernels' own dispatch packets).  aqua_event_elapsed_ms(first_start, last_stop) is then the time from the first
 * wavefront's start to the last kernel's end -- launches and the boundaries between them, nothing else.  Events RECORDED
 * on the stream around the same launches (aqua_event_record) are marker packets of their own and read 12-14 us more per
 * pair (profiles/r03/burst_timeline.txt).  Not capturable into a graph (a graph node carries no events); with N == 0 or
 * T == 0 the events are left untouched.  Both events must be complete before they are read (synchronise the stream).
 */
typedef struct AquaEvent AquaEvent;
int aqua_rollout_events_f32(const AquaParams* p, const void* obst_blob_dev, int K, int64_t N, int64_t env_offset,
                            float* state, int64_t ld, int32_t* time, int64_t T, const void* actions, int action_kind,
                            int64_t action_ld, int64_t action_step_stride, uint64_t seed, uint64_t tick,
                            const uint64_t* tick_base_dev, float* reward, uint8_t* term, int64_t out_step_stride,
                            uint64_t* done_bits, int64_t done_step_stride, float* obs_norm, int auto_reset, int advance_tick,
                            AquaEvent* first_start, AquaEvent* last_stop, void* stream);

/*
 * The same T steps fused into ONE launch: pose, goal, wave and time stay in registers between
 * steps, so HBM traffic per world-step drops to the action read and the reward/term writes.
 * Results are identical to aqua_rollout_f32 with the same arguments (tests/test_hip_parity.py).
 */
int aqua_rollout_fused_f32(const AquaParams* p, const void* obst_blob_dev, int K, int64_t N, int64_t env_offset,
                           float* state, int64_t ld, int32_t* time, int64_t T, const void* actions,
                           int action_kind, int64_t action_ld, int64_t action_step_stride, uint64_t seed,
                           uint64_t tick, const uint64_t* tick_base_dev, float* reward, uint8_t* term,
                           int64_t out_step_stride, int auto_reset, void* stream);

/*
 * Per-world obstacle tables: every world of the batch has its own obstacle list, as every env object of the
 * reference has (AquaEnv(obstacles=[...]), aqua.py:13,56-68) -- domain randomisation over a batch.
 *
 * aqua_pack_tables() converts rows[N][K][5] float64 (the reference's format, cx, cy, kind 0 circle | 1 rectangle,
 * a, b; kind < 0 marks an absent row, so worlds may hold fewer than K obstacles) into the two device-format arrays,
 * on the HOST; the caller uploads them:
 *   tab32  float32 [K][6][tld]  (cx, cy, hx, hy, R^2, w) per row, struct of arrays over the worlds (coalesced: what the
 *          step streams), FOLLOWED BY the same rows world-major, float32 [tld][K][6]: a world's whole table contiguous --
 *          what a restart re-reads (one world's K rows are 6 K different cache lines of the struct of arrays, 1.5 K / 128
 *          of the copy).  aqua_tables32_floats(K, tld) = 12 K tld is the size of the buffer in floats.
 *   tab64  float64 [K][5][tld]  the rows as given, for the float64 knife-edge path
 * and returns the largest collision radius of the batch in *r_max (sizes the knife-edge bands).
 *
 * aqua_step_tables_f32 / aqua_rollout_tables_f32 / aqua_reset_tables_f32 are aqua_step_f32 / aqua_rollout_f32 /
 * aqua_reset_f32 with these tables: every restart mode (AQUA_RESET_NEXT_STEP keeps the same markers in time[]; a finished world
 * restarts inside the stepping tile, its own lane handing its rows to the re-seeding lanes through LDS -- from registers
 * for tables of up to 16 rows, as the row loop streams them for longer ones), every action kind, and all other arguments mean what they mean there.  Results for a batch whose
 * worlds all hold the same list are bit-identical to the shared-table calls.  Algorithmic bytes per world-step:
 * 62 + 24 K (discrete).
 */
size_t aqua_tables32_floats(int K, int64_t tld);
int aqua_pack_tables(const double* rows, int K, int64_t N, int64_t tld, float* tab32_host, double* tab64_host,
                     float* r_max);
int aqua_step_tables_f32(const AquaParams* p, const float* tab32_dev, const double* tab64_dev, int K, int64_t tld,
                         float r_max, int64_t N, int64_t env_offset, float* state, int64_t ld, int32_t* time,
                         const void* action, int action_kind, int64_t action_ld, const float* noise, int64_t noise_ld,
                         uint64_t seed, uint64_t tick, const uint64_t* tick_base_dev, float* reward, uint8_t* term,
                         uint64_t* done_bits, float* obs_norm, int auto_reset, void* stream);
int aqua_rollout_tables_f32(const AquaParams* p, const float* tab32_dev, const double* tab64_dev, int K, int64_t tld,
                            float r_max, int64_t N, int64_t env_offset, float* state, int64_t ld, int32_t* time, int64_t T,
                            const void* actions, int action_kind, int64_t action_ld, int64_t action_step_stride,
                            uint64_t seed, uint64_t tick, const uint64_t* tick_base_dev, float* reward, uint8_t* term,
                            int64_t out_step_stride, uint64_t* done_bits, int64_t done_step_stride, float* obs_norm,
                            int auto_reset, int advance_tick, void* stream);
/* aqua_rollout_fused_f32 with per-world tables: T steps in ONE launch, the state in registers and the block's tables in
 * LDS for the whole rollout (40 KB per block of 256 worlds for up to 8 rows; 80 KB for 9..16 rows, for 17..32 rows with
 * 128 worlds per block and for 33..64 rows with 64).  Results are identical to aqua_rollout_tables_f32 with the same
 * arguments. */
int aqua_rollout_tables_fused_f32(const AquaParams* p, const float* tab32_dev, const double* tab64_dev, int K, int64_t tld,
                                  float r_max, int64_t N, int64_t env_offset, float* state, int64_t ld, int32_t* time, int64_t T,
                                  const void* actions, int action_kind, int64_t action_ld, int64_t action_step_stride,
                                  uint64_t seed, uint64_t tick, const uint64_t* tick_base_dev, float* reward, uint8_t* term,
                                  int64_t out_step_stride, int auto_reset, void* stream);
int aqua_reset_tables_f32(const AquaParams* p, const float* tab32_dev, int K, int64_t tld, int64_t N, int64_t env_offset,
                          float* state, int64_t ld, int32_t* time, const uint8_t* mask, uint64_t seed, uint64_t tick,
                          const uint64_t* tick_base_dev, void* stream);

/*
 * obs_norm[5][ld] <- the observation of `state` scaled as the reference's AquaStateNormalizer
 * (main/impl/utils.py:15-33): what the step kernels' obs_norm epilogue writes, for the worlds a reset just
 * placed (mask as in aqua_reset_f32; NULL: all N).
 */
int aqua_obs_norm_f32(const float* state, int64_t ld, int64_t N, const uint8_t* mask, float* obs_norm, void* stream);

/*
 * Device-side replay ring (the experience buffer of main/impl/dqn.py:174, `exp_buffer.append([s, a, r, s', d])`,
 * for a batch of worlds): ring[r][(cursor + i) % capacity] = src[r][i] for r < rows, i < N.  Rows are
 * float32 (aqua_ring_write_f32) or bytes (aqua_ring_write_u8); `ring_ld`/`src_ld` are the row pitches in
 * elements (ring_ld >= capacity).  Coalesced on both sides; N <= capacity.  A batched step is recorded as two
 * calls around aqua_step_f32 into the same slots -- (s, a) before it, (r, s', d) after it.
 */
int aqua_ring_write_f32(float* ring, int64_t ring_ld, int64_t capacity, int64_t cursor, const float* src, int64_t src_ld,
                        int rows, int64_t N, void* stream);
int aqua_ring_write_u8(uint8_t* ring, int64_t ring_ld, int64_t capacity, int64_t cursor, const uint8_t* src,
                       int64_t src_ld, int rows, int64_t N, void* stream);

/* *tick_base_dev += delta, as a 1-thread kernel on `stream` (the last node of a captured rollout graph). */
int aqua_tick_advance(uint64_t* tick_base_dev, uint64_t delta, void* stream);

/* HIP-graph helpers: capture whatever is launched on `stream` between begin and end, replay it. */
typedef struct AquaGraph AquaGraph;
int aqua_graph_begin(void* stream);
int aqua_graph_end(void* stream, AquaGraph** out);
int aqua_graph_launch(AquaGraph* g, void* stream);
/* hipGraphUpload: the executable graph's device-side resources are set up on `stream` now instead of inside its first
 * launch (bench.py: a timed graph that the warm-up never replayed would otherwise pay for that inside region 0). */
int aqua_graph_upload(AquaGraph* g, void* stream);
int aqua_graph_destroy(AquaGraph* g);

/*
 * Timing events (HIP events).  aqua_event_record() stamps the event when `stream` reaches it.
 * aqua_graph_end_timed() is aqua_graph_end() with two event-record NODES added to the captured graph -- `start` ahead
 * of everything captured, `stop` behind it -- so that every replay stamps both and their distance is exactly the
 * captured launches (bench.py's roofline.launch_us: no host launch latency inside the interval).
 * aqua_event_elapsed_ms() needs both events recorded and complete (synchronise the stream first).
 */
/* (AquaEvent is declared above, with aqua_rollout_events_f32) */
int aqua_event_create(AquaEvent** out);
int aqua_event_record(AquaEvent* e, void* stream);
int aqua_graph_end_timed(void* stream, AquaGraph** out, AquaEvent* start, AquaEvent* stop);
int aqua_event_elapsed_ms(AquaEvent* start, AquaEvent* stop, float* ms);
int aqua_event_destroy(AquaEvent* e);

/*
 * Done-mask exchange without a collective kernel (peer writes through hipIpcMemHandle).
 * The one exchange of the sharded path is the episodic done mask (32 KiB of ballot words per step and GPU).  RCCL's
 * all-gather runs on the compute units and takes them from the step kernels (one rank: 5.07 -> 6.15 us per step); a
 * device-to-device copy does not.  So every rank owns a receive buffer that the other ranks of the node map and
 * write into with asynchronous copies over xGMI:
 *   aqua_ipc_buffer_create/ptr/handle/destroy   the receive buffer (the one device allocation the library makes: it has to
 *                                               be an allocation of its own to be exported) and its 64-byte handle, which
 *                                               the host side hands to the other ranks over its process group;
 *   aqua_ipc_open / aqua_ipc_close              map / unmap another rank's buffer (hipIpcOpenMemHandle, lazy peer access);
 *   aqua_copy_async                             device-to-device copy on `stream` (the side streams of aquaticgymenv_amd/
 *                                               sharded.py DoneMaskExchange(kind="ipc")): AQUA_COPY_ENGINE_WAVES = eight
 *                                               single-wavefront workgroups (whole 8-byte words; they fit beside the step
 *                                               kernel's one round of blocks), AQUA_COPY_ENGINE_DMA = hipMemcpyAsync.
 * The replaced reference interface is none: the reference is one process, one env object (main/testing/__init__.py:17-36).
 */
#define AQUA_IPC_HANDLE_BYTES 64
typedef struct AquaIpcBuffer AquaIpcBuffer;
int aqua_ipc_buffer_create(size_t bytes, AquaIpcBuffer** out);
void* aqua_ipc_buffer_ptr(AquaIpcBuffer* b);
int aqua_ipc_buffer_handle(AquaIpcBuffer* b, unsigned char handle[AQUA_IPC_HANDLE_BYTES]);
int aqua_ipc_buffer_destroy(AquaIpcBuffer* b);
int aqua_ipc_open(const unsigned char handle[AQUA_IPC_HANDLE_BYTES], void** peer_ptr);
int aqua_ipc_close(void* peer_ptr);
#define AQUA_COPY_ENGINE_WAVES 0
#define AQUA_COPY_ENGINE_DMA   1
int aqua_copy_async(void* dst, const void* src, size_t bytes, int engine, void* stream);
/* the same bytes into n_dst buffers with ONE launch (wavefronts; blockIdx.y = destination): the last block of a region,
 * queued in stream order right behind the region's last step */
int aqua_copy_fanout_async(void* const* dsts, int n_dst, const void* src, size_t bytes, void* stream);

/* the float32 constants the kernels use for the three discrete actions (aqua.py:33-42 folded through
 * aqua.py:159-170): out = h[3] (w/2), w[3], chord[3]; for tests. */
void aqua_discrete_constants(float out[9]);
#ifdef __cplusplus
}
#endif
#endif /* AQUA_HIP_H */
