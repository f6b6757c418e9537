// taco_step.hpp -- the fused VecTask.step() kernel for gfx950 (MI355X).
//
// All persistent per-env state lives in HBM as 16 float4 CHUNKS per env, TILE-major (tiles of 64 consecutive envs: chunk c of env i at
// byte (((i >> 6) * 16 + c) * 64 + (i & 63)) * 16, see "workspace layout" below), so every state load / store of a wavefront is ONE
// fully coalesced 16-byte-per-lane access and everything a wavefront touches of an array is one contiguous block.  One lane = one env
// in the throughput forms, four lanes = one env ("quad layout") in the latency forms (see the instantiation table above the kernel).
// One launch performs, for every env: reset + domain randomisation (if its reset flag is set), the pending-action delay line,
// 10 x [euler unwrap, body rates, rate PID, allocator, battery sag, rotor first-order lag, aero, rigid-body integrate], then the relative
// state, the 26-D observation / state frames (transposed through a per-wavefront LDS tile so the [env][frame][26] rows are written with
// coalesced stores), the task reward and the done / time-out flags.  No MFMA: this is element-wise ODE integration, the largest
// contraction is 4x4 . 4.
//
// Reference lines restated by each block are cited inline (FA = isaacgymenvs/tasks/fpv_asymmetry.py,
// VT = tasks/base/vec_task_asymmetry.py, CTRL = tasks/control/, TU = python/isaacgym/torch_utils.py,
// TJ = isaacgymenvs/utils/torch_jit_utils.py).
#pragma once
#include <type_traits>
#include "taco_math.hpp"
#include "../../include/taco_env.h"

#ifndef TACO_AB_SERVE_ALWAYS
#define TACO_AB_SERVE_ALWAYS 0  // A/B builds only: the battery server in every four-role quad launch, whatever its size
#endif

namespace taco {

struct StepParams {
    // device pointers
    float *S;            // [tiles][NUM_CHUNKS][64] float4 state chunks (tile = 64 consecutive envs, see "workspace layout" below)
    float *ring;         // [tiles][100][64] float4 pending-action ring, PHYSICAL slot order (logical slot s lives at (head+s)%100)
    float *hist;         // [tiles][16][64] float4 action history, row = step count & 15
    const float *act_in; // [n][4]
    float *obs;          // [n][len_obs][26]
    float *states;       // [n][len_states][26]
    float *rew;          // [n]
    long long *reset;    // [n] in/out
    uint8_t *timeout;    // [n]
    const float *obs_prev;    // frame stacks the len>1 shift reads (== obs / states for the in-place step; the previous
    const float *states_prev; //   replay-buffer slot for taco_step_rollout)
    float *done_f32;     // optional [n] f32 image of the new reset flags (PPOReplayBuffer.done_buf, buffer_asymmetry.py:67)
    unsigned long long *stamps;  // optional [8]: shader-clock stamps of workgroup 0 at the phase boundaries (taco_bind_phase_stamps)
    float *gather;       // optional [n][len_obs*26 + 3] f32: obs stack | reward | done | time-out (one all-gather block per rank)
    float *obs_out;      // optional [n][len_obs][26]: clamp(new obs stack, +-clip_obs), the copy VecTask.step returns (VT:331); NULL = none
    float *states_out;   // optional [n][len_states][26]: clamp(new states stack, +-clip_states) (VT:332)
    uint32_t *ctl;       // control block in the workspace (kCtl*): the device-resident step clock (the copy the HOST reads back) + the sticky status word
    uint32_t *wclk;      // [ceil(npad / 16)][2] the device-resident step clock AS THE KERNELS READ IT: one {step, aux} pair per 16 envs (see "wclk" below)
    int use_ctl;         // 1: take step / head / hh / ring phase from wclk (launch captured into a HIP graph: kernel arguments are frozen); 0: from below
    int stamps_on;       // stamps != NULL, as a flag the kernels can test without touching the pointer (StepKernelArgs: it arrives in a register)
    // geometry / cfg
    int n, npad, env_offset, task_mode, mix_n1, mix_n2, len_obs, len_states, substeps, max_len, delay_time, head;
    uint32_t flags, seed_lo, seed_hi, step;
    int hh;              // history row this step writes (= number of steps taken so far, mod 16)
    int hw_rows;         // how many of the most recent history rows the up-front loads fetch (1..4): the deepest run queue this configuration
                         // can reach (taco_capi.hip derive()); lanes with a deeper queue gather per lane -- a performance hint, never semantics
    uint32_t obs_bytes, states_bytes, gather_bytes, gather_row;  // gather_row = floats per env in the gather block (multiple of 32)
    // FRAME RING behind VecTask.step()'s state stacks (taco_bind_states_ring; st_period == 0: off).  `states` is then the ring
    // [st_front + st_period][n][26] (st_front = len_states - 1), states_bytes one ROW of it, len_states == 1 for the launch: the step with ring
    // phase ph (part of the step clock) writes its ONE new frame to row st_front + ph -- and, while ph >= st_period - st_front, also to row
    // st_front + ph - st_period, so that rows [ph, ph + st_front] always hold the stack of that step, oldest frame first: the reference's
    // [n][len][26] stack is the strided view ring[ph : ph + len].permute(1, 0, 2), nothing is ever shifted (FA:392, :413 shift by one per step
    // and never clear: exactly a ring).
    int st_period, st_phase, st_front;
    // fp32 images of the Python doubles the reference feeds into tensor ops
    float dt, rdt, clip_act, df;  // rdt = RN(1/dt); taco_create refuses a dt for which div_const(x, dt, rdt) != x / dt
    float clip_obs, clip_states;  // bounds of obs_out / states_out (+inf = none)
    float h, half_h, inv_m, g, J0, J1, J2, hJi0, hJi1, hJi2, arm_x, arm_y;
    float flip_xy_sc, flip_xy_lo, flip_v_sc, flip_v_lo, dr_sc, dr_lo, tau_sc, tau_lo, tau_fixed, nq_sc, nq_lo;
};

// KERNEL-ARGUMENT PRELOAD.  gfx950 delivers the first user SGPRs' worth of kernel arguments IN REGISTERS when a wavefront starts (the build passes
// -amdgpu-kernarg-preload-count; byref aggregates are not eligible, hence the leading scalars).  Everything the up-front loads of a step need -- the
// state, flag and action pointers (history and device clock follow the state in the workspace), the env count, the flags, the history row --
// and what the role wavefronts' reset draws need (Philox key, step, task) leads the argument list, so those loads and draws are issued
// without the round trip to the argument segment (a launch's argument block is fresh memory) that every wavefront used to sit out first; the
// parameter block behind them (a complete StepParams: the leading scalars repeat its fields) is read while the state loads are in flight.
struct StepKernelArgs {   // (the layout of the kernel-argument segment: what lazy_arg and the line touches address)
    float *S; long long *reset; const float *act_in; uint32_t seed_lo, seed_hi, step; int task_mode, n; uint32_t flags; uint32_t early; int env_offset;
    StepParams P;
};
// (history and device clock are not among them: they lie at fixed offsets behind the state in the handle's workspace, taco_capi.hip taco_create)
static_assert(sizeof(StepKernelArgs) - sizeof(StepParams) == 56, "fourteen dwords: what gfx950 preloads beside the segment pointer");
constexpr size_t kParamsOffset = offsetof(StepKernelArgs, P);
// `early`: hh | hw_rows << 8 | stamps bound << 30 | use_ctl << 31
inline __host__ __device__ uint32_t pack_early(int hh, int hw_rows, bool stamps, int use_ctl) { return (uint32_t)(hh & 15) | ((uint32_t)(hw_rows & 7) << 8) | (stamps ? 1u << 30 : 0u) | (use_ctl ? 1u << 31 : 0u); }

// ---- workspace layout (TILE-major float4 rows): the 67 state words of an env are packed into 17 float4 CHUNKS; the envs are
// grouped in TILES of 64 consecutive envs and chunk c of env i lives at byte ((i / 64) * NUM_CHUNKS + c) * 1024 + (i % 64) * 16.
// A wavefront still moves a chunk with ONE 16-byte-per-lane access (1 KiB, fully coalesced), and everything a wavefront touches of
// an array is ONE contiguous block (17 KiB of state, 16 KiB of history) whatever num_envs is.  Round 1 kept whole rows
// ([chunk][npad]): with num_envs a power of two the 17 + 4 streams of a wavefront were then exactly 2^k bytes apart and fell on the
// same HBM channels (tools/ubench/layout: 4.2-4.7 TB/s for the step's access pattern at 1 M envs against 5.3-5.7 TB/s tile-major).
// Words that travel together share a chunk.
// Pending actions: a run-length queue (packed into the spare bits of C_POS.w / C_LINVEL.w, see below) over the action history
// hist[tile][16][64] float4 while an env is in the common regime, the literal slot ring ring[tile][100][64] float4 (PHYSICAL slot order)
// once it is DENSE (see the kernel).
enum Chunk : uint32_t {
    C_POS = 0,      // p.x p.y p.z | progress (16 bits) + run-length codes 0..7 (see "queue words")
    C_QUAT = 1,     // q.x q.y q.z q.w
    C_LINVEL = 2,   // v.x v.y v.z | delay word (see "queue words")
    C_ANGVEL = 3,   // w.x w.y w.z | battery voltage
    C_PID_PREV = 4, // previous_error xyz | battery E_c
    C_PID_INT = 5,  // integral xyz | battery u_1
    C_OMEGA = 6,    // rotor speeds
    C_MISC = 7,     // battery time | command[0] | command[1] | flip_radian
    C_RPY_OLD = 8,  // xyz | -
    C_RPY_CONT = 9,
    C_TGT_POS = 10, // xyz | -
    C_TGT_QUAT = 11,
    C_TAU = 12,     // rotor response times
    C_OPARA = 13,   // omega_para[0..3]
    C_AERO0 = 14,   // omega_para[4] | cf | ct | dx
    C_AERO1 = 15,   // dy | kt | - | -
    NUM_CHUNKS = 16
};
constexpr int HIST_ROWS = 16;  // action history: hist[tile][k][64] float4, the action of the step whose (count & 15) == k
// control block: the last 256 bytes of the workspace.  Words: 0 step count, 1 ring head, 2 history row, 3 phase of the states frame ring
// (taco_bind_states_ring) -- the step CLOCK as the host reads it back (refresh_clock): workgroup 0 of every launch leaves the NEXT values
// here --, 4 sticky status bits.
// kCtlDevSteps counts the launches that took their clock from the DEVICE (use_ctl: every launch of a handle in graph mode, and every replay of a
// graph captured from it, whenever that happens); eager launches outside graph mode never touch it -- what lets taco_check see a replay made after
// taco_release_graphs even once later eager steps have overwritten kCtlStep
enum : uint32_t { kCtlStep = 0, kCtlHead = 1, kCtlHh = 2, kCtlPhase = 3, kCtlStatus = 4, kCtlDevSteps = 5, kCtlBytes = 256 };
// wclk: the same clock where the KERNELS take it from when their arguments are frozen in a HIP graph: one {step, aux} pair per 16 envs
// (aux = ring head | history row << 8 | frame-ring phase << 16), all pairs equal between launches.  A step wavefront reads the pair of its own
// envs and leaves the next values there when it is done -- every word has ONE reader-writer per launch, so a captured step advances the clock
// with no synchronisation between workgroups at all and is ONE graph node.  (Round 2-3 kept one global clock and appended a one-thread
// kernel to every captured step: +1.2 us per step; a last-ticket scheme on the global words cost more -- 256 agent-scope atomics on one
// address take ~10 us on an eight-XCD part, and vmcnt retires in order.)  Eager launches take the clock from their arguments and write the
// pairs just the same, so that a capture can start at any time.
constexpr uint32_t kWclkWordsPerGroup = 2;
__host__ __device__ constexpr uint32_t wclk_aux(int head, int hh, int phase) { return (uint32_t)head | ((uint32_t)hh << 8) | ((uint32_t)phase << 16); }
__host__ __device__ constexpr int wclk_head(uint32_t aux) { return (int)(aux & 0xffu); }
__host__ __device__ constexpr int wclk_hh(uint32_t aux) { return (int)((aux >> 8) & 0xffu); }
__host__ __device__ constexpr int wclk_phase(uint32_t aux) { return (int)(aux >> 16); }
__host__ __device__ constexpr size_t wclk_words(int npad) { return (size_t)(npad / 16) * kWclkWordsPerGroup; }
constexpr uint32_t kStatusMailboxTimeout = 1u;  // a battery-mailbox wait gave up (the affected envs' voltage was poisoned with NaN)
constexpr uint32_t kStatusCriticClamped = 2u;   // taco_rollout_run's split-f16 critic saturated a finite frame word beyond +-65 504 (taco_policy.hpp split_frames)
constexpr uint32_t kRowBytes = 1024;  // one row (chunk, history row or ring slot) of one tile: 64 lanes x 16 B
// float index of word 0 of (row r, env i) in an array with `rows` rows per tile (host-side kernels of taco_capi.hip)
__host__ __device__ constexpr size_t tile_word(int rows, int r, int i) { return (((size_t)(i >> 6) * rows + r) * 64 + (size_t)(i & 63)) * 4; }
// blob row (include/taco_env.h "State blob") -> chunk * 4 + component; -1 for the rows that are not stored as such
// (actions / actions_old live in the action history, see export_state_kernel)
__host__ __device__ constexpr int field_slot(int f) {
    return f < 3 ? C_POS * 4 + f : f < 7 ? C_QUAT * 4 + (f - 3) : f < 10 ? C_LINVEL * 4 + (f - 7) : f < 13 ? C_ANGVEL * 4 + (f - 10)
         : f < 16 ? C_TGT_POS * 4 + (f - 13) : f < 20 ? C_TGT_QUAT * 4 + (f - 16) : f < 23 ? C_RPY_OLD * 4 + (f - 20)
         : f < 26 ? C_RPY_CONT * 4 + (f - 23) : f < 29 ? C_PID_PREV * 4 + (f - 26) : f < 32 ? C_PID_INT * 4 + (f - 29)
         : f == 32 ? C_PID_PREV * 4 + 3 : f == 33 ? C_PID_INT * 4 + 3 : f == 34 ? C_MISC * 4 + 0 : f == 35 ? C_ANGVEL * 4 + 3
         : f < 40 ? C_OMEGA * 4 + (f - 36) : f < 48 ? -1
         : f == 48 ? C_MISC * 4 + 1 : f == 49 ? C_MISC * 4 + 2 : f == 50 ? C_MISC * 4 + 3 : f < 55 ? C_TAU * 4 + (f - 51)
         : f < 59 ? C_OPARA * 4 + (f - 55) : f == 59 ? C_AERO0 * 4 + 0 : f == 60 ? C_AERO0 * 4 + 1 : f == 61 ? C_AERO0 * 4 + 2
         : f == 62 ? C_AERO0 * 4 + 3 : f == 63 ? C_AERO1 * 4 + 0 : f == 64 ? C_AERO1 * 4 + 1 : f == 65 ? C_POS * 4 + 3 : C_LINVEL * 4 + 3;
}
// queue words.  The pending-action bookkeeping of an env is 43 bits and rides in the fourth words of two chunks the step rewrites anyway
// (round 1 gave it a chunk of its own: 16 B read + 16 B written per env-step for 8 B of content):
//   C_LINVEL.w  bits 0..6 L (pending slots) | 7..13 Z (leading reset-zeros) | 14..17 m (queued runs) | 18..21 rem0 (slots left in the oldest
//               run) | 22..25 run-length codes 8, 9 | 31 DENSE
//   C_POS.w     bits 0..15 progress_buf (saturates at 65 535; taco_create refuses max_episode_length > 65 535) | 16..31 run-length codes 0..7
// run-length code = T - 9, two bits per run, oldest run first (at most 10 runs are ever queued: L + T <= 90 and every full run >= 9 slots)
constexpr int kProgressMax = 0xffff;
__host__ __device__ constexpr int dw_L(uint32_t a) { return (int)(a & 0x7fu); }
__host__ __device__ constexpr int dw_Z(uint32_t a) { return (int)((a >> 7) & 0x7fu); }
__host__ __device__ constexpr int dw_m(uint32_t a) { return (int)((a >> 14) & 0xfu); }
__host__ __device__ constexpr int dw_rem0(uint32_t a) { return (int)((a >> 18) & 0xfu); }
__host__ __device__ constexpr bool dw_dense(uint32_t a) { return (a >> 31) != 0u; }
__host__ __device__ constexpr int qw_lens(uint32_t a, uint32_t pw) { return (int)((pw >> 16) | (((a >> 22) & 0xfu) << 16)); }
__host__ __device__ constexpr int pw_progress(uint32_t pw) { return (int)(pw & 0xffffu); }
__host__ __device__ constexpr uint32_t make_dw(int L, int Z, int m, int rem0, int lens, bool dense) {
    return (uint32_t)L | ((uint32_t)Z << 7) | ((uint32_t)m << 14) | ((uint32_t)rem0 << 18) | ((((uint32_t)lens >> 16) & 0xfu) << 22) | (dense ? 0x80000000u : 0u);
}
__host__ __device__ constexpr uint32_t make_pw(int progress, int lens) {
    return (uint32_t)(progress > kProgressMax ? kProgressMax : progress) | (((uint32_t)lens & 0xffffu) << 16);
}
__host__ __device__ constexpr int run_len(int lens, int j) { return 9 + ((lens >> (2 * j)) & 3); }

// The fp32 constants of the PID / battery / rigid-body arithmetic.  As kernel arguments they are SGPR operands, and a VALU instruction with an
// SGPR source costs TWO issue passes on gfx950 once the SIMD is shared (2.0-2.1 ns against 1.05 ns for VGPR / literal / inline operands:
// tools/ubench/bank2, profiles/r02_a_ubench_bank2.txt) -- 52 of the 520 VALU instructions of a substep.  The PLAIN loop forms therefore
// exist for the DEFAULT airframe and time step only (dt = 0.001, two sub-iterations, assets/xml/fpv_without_duct.xml, g = -9.81), with
// these values as literals; taco_capi.hip derive() sets kPlainConsts when the handle's derived constants equal them bit for bit, any
// other configuration runs the general forms.
struct Consts { float dt, rdt, h, half_h, inv_m, g, J0, J1, J2, hJi0, hJi1, hJi2, arm_x, arm_y; };
constexpr Consts kDefaultConsts = {0x1.0624dep-10f, 0x1.f3fffep+9f, 0x1.0624dep-11f, 0x1.0624dep-12f, 0x1.1642a8p+1f, -0x1.39eb86p+3f,
                                   0x1.0690a2p-11f, 0x1.6f6c0ap-11f, 0x1.a3da2ap-11f, 0x1.ff2ddap-1f, 0x1.6d4bbap-1f, 0x1.3fadb2p-1f,
                                   0x1.810624p-5f, 0x1.e353f8p-5f};
constexpr uint32_t kPlainConsts = 1u << 30;      // StepParams.flags: the handle's constants are kDefaultConsts
// StepParams.flags: rotor / aero parameters are the SAME for every env (none of them is randomised, the handle never imported a state
// blob): the step then takes them from the kernel arguments instead of loading four chunks (64 B of the 329 B an env-step reads)
constexpr uint32_t kUniformParams = 1u << 29;
constexpr uint32_t kDebugSlowServer = 1u << 31;  // StepParams.flags: test hook, compiled only under -DTACO_TEST_HOOKS (taco_capi.hip derive())
enum : uint32_t { STREAM_RESET = 1, STREAM_CMD = 2, STREAM_DEPLOY = 3, STREAM_ROTOR = 4, STREAM_OBS = 5 };


// ---- memory access.  State, history and ring rows are reached through 128-bit buffer descriptors BASED AT THE WAVEFRONT'S TILE
// (the base is wave-uniform scalar arithmetic): one VGPR holds the lane's byte offset inside a row (16 * (i % 64)) for EVERY row of
// every array, the row offset (row * 1024) rides in the scalar soffset operand.  With plain 64-bit pointers each of the row addresses
// would occupy a VGPR pair for the whole kernel (measured: > 130 VGPRs).  Accesses outside the tile's block are dropped / return 0
// by the hardware range check instead of faulting.
// The intrinsics are declared directly (the composable_kernel idiom): on this toolchain (clang 22 / ROCm 7.2)
// __builtin_amdgcn_raw_buffer_load_b128 lowers to a ONE-dword load whose value is splat over the vector.
typedef int rsrc_t __attribute__((ext_vector_type(4)));
typedef float f32x4_t __attribute__((ext_vector_type(4)));
__device__ f32x4_t llvm_amdgcn_raw_buffer_load_v4f32(rsrc_t rsrc, int voffset, int soffset, int aux) __asm("llvm.amdgcn.raw.buffer.load.v4f32");
__device__ void llvm_amdgcn_raw_buffer_store_v4f32(f32x4_t v, rsrc_t rsrc, int voffset, int soffset, int aux) __asm("llvm.amdgcn.raw.buffer.store.v4f32");
typedef float f32x2_t __attribute__((ext_vector_type(2)));
__device__ f32x2_t llvm_amdgcn_raw_buffer_load_v2f32(rsrc_t rsrc, int voffset, int soffset, int aux) __asm("llvm.amdgcn.raw.buffer.load.v2f32");
__device__ void llvm_amdgcn_raw_buffer_store_v2f32(f32x2_t v, rsrc_t rsrc, int voffset, int soffset, int aux) __asm("llvm.amdgcn.raw.buffer.store.v2f32");
__device__ void llvm_amdgcn_raw_buffer_store_f32(float v, rsrc_t rsrc, int voffset, int soffset, int aux) __asm("llvm.amdgcn.raw.buffer.store.f32");
TD rsrc_t make_rsrc(const void *base, uint32_t bytes) {
    const uint64_t a = (uint64_t)base;  // wave-uniform: comes from the kernel arguments
    return rsrc_t{(int)(uint32_t)a, (int)((uint32_t)(a >> 32) & 0xffffu), (int)bytes, 0x00020000};
}
TD float4 buf_ld4(rsrc_t r, uint32_t voff, uint32_t soff) {
    const f32x4_t u = llvm_amdgcn_raw_buffer_load_v4f32(r, (int)voff, (int)soff, 0);
    return make_float4(u.x, u.y, u.z, u.w);
}
// 16-byte STORES never use the scalar offset operand: on gfx950 a buffer_store_dwordx4 with an SGPR soffset followed by a
// VALU write to its data registers can ship the NEW value of the first data dword for the lanes read last (observed:
// lanes 12-15 of each 16-lane group, non-deterministically).  LLVM's hazard recogniser only guards the no-SGPR-soffset
// form (GCNHazardRecognizer::createsVALUHazard), so the row offset is folded into the per-lane offset instead.
TD void buf_st4(rsrc_t r, float4 v, uint32_t voff, uint32_t soff) {
    llvm_amdgcn_raw_buffer_store_v4f32(f32x4_t{v.x, v.y, v.z, v.w}, r, (int)(voff + soff), 0, 0);
}
// per-component select (a ternary on the aggregate makes clang pick between two stack slots)
TD float4 sel4(bool c, float4 a, float4 b) { return make_float4(c ? a.x : b.x, c ? a.y : b.y, c ? a.z : b.z, c ? a.w : b.w); }
TD float as_f(int x) { return __builtin_bit_cast(float, x); }
TD int as_i(float x) { return __builtin_bit_cast(int, x); }

struct V3 { float x, y, z; };
struct Q4 { float x, y, z, w; };

// TU:19-40 quat_mul
TD Q4 quat_mul(Q4 a, Q4 b) {
    float ww = (a.z + a.x) * (b.x + b.y);
    float yy = (a.w - a.y) * (b.w + b.z);
    float zz = (a.w + a.y) * (b.w - b.z);
    float xx = ww + yy + zz;
    float qq = 0.5f * (xx + (a.z - a.x) * (b.x - b.y));
    Q4 o;
    o.w = qq - ww + (a.z - a.y) * (b.y - b.z);
    o.x = qq - xx + (a.x + a.w) * (b.x + b.w);
    o.y = qq - yy + (a.w - a.x) * (b.y + b.z);
    o.z = qq - zz + (a.z + a.y) * (b.w - b.x);
    return o;
}
TD Q4 conj(Q4 q) { return Q4{-q.x, -q.y, -q.z, q.w}; }
// TU:58-68 quat_rotate(q, v)
TD V3 quat_rotate(Q4 q, V3 v) {
    float s = 2.0f * (q.w * q.w) - 1.0f;
    float cx = cross_term(q.y, v.z, q.z, v.y);
    float cy = cross_term(q.z, v.x, q.x, v.z);
    float cz = cross_term(q.x, v.y, q.y, v.x);
    float dot = q.x * v.x + q.y * v.y + q.z * v.z;
    V3 o;
    o.x = v.x * s + cx * q.w * 2.0f + q.x * dot * 2.0f;
    o.y = v.y * s + cy * q.w * 2.0f + q.y * dot * 2.0f;
    o.z = v.z * s + cz * q.w * 2.0f + q.z * dot * 2.0f;
    return o;
}
// TU:175-196 get_euler_xyz_v1
TD V3 euler_xyz_v1(Q4 q) {
    V3 e;
    float sinr_cosp = 2.0f * (q.w * q.x + q.y * q.z);
    float cosr_cosp = q.w * q.w - q.x * q.x - q.y * q.y + q.z * q.z;
    e.x = atan2(sinr_cosp, cosr_cosp);
    float sinp = 2.0f * (q.w * q.y - q.z * q.x);
    if (absf(sinp) >= 1.0f) e.y = kHalfPi * (sinp > 0.0f ? 1.0f : (sinp < 0.0f ? -1.0f : 0.0f));
    else e.y = asin(sinp);
    float siny_cosp = 2.0f * (q.w * q.z + q.x * q.y);
    float cosy_cosp = q.w * q.w + q.x * q.x - q.y * q.y - q.z * q.z;
    e.z = atan2(siny_cosp, cosy_cosp);
    return e;
}
// TU:199-213 quat_from_euler_xyz
TD Q4 quat_from_sincos(float sr, float cr, float sp, float cp, float sy, float cy) {
    Q4 q;
    q.w = cy * cr * cp + sy * sr * sp;
    q.x = cy * sr * cp - sy * cr * sp;
    q.y = cy * cr * sp + sy * sr * cp;
    q.z = sy * cr * cp - cy * sr * sp;
    return q;
}
TD Q4 quat_from_euler(float roll, float pitch, float yaw) {
    float cy, sy, cr, sr, cp, sp;
    sincos(yaw * 0.5f, sy, cy);
    sincos(roll * 0.5f, sr, cr);
    sincos(pitch * 0.5f, sp, cp);
    Q4 q;
    q.w = cy * cr * cp + sy * sr * sp;
    q.x = cy * sr * cp - sy * cr * sp;
    q.y = cy * cr * sp + sy * sr * cp;
    q.z = sy * cr * cp - cy * sr * sp;
    return q;
}
// TJ:389-416 quaternion_to_matrix, row-major
TD void quat_to_matrix(Q4 q, float m[9]) {
    float i = q.x, j = q.y, k = q.z, r = q.w;
    float two_s = 2.0f / (((i * i + j * j) + k * k) + r * r);
    m[0] = 1.0f - two_s * (j * j + k * k);
    m[1] = two_s * (i * j - k * r);
    m[2] = two_s * (i * k + j * r);
    m[3] = two_s * (i * j + k * r);
    m[4] = 1.0f - two_s * (i * i + k * k);
    m[5] = two_s * (j * k - i * r);
    m[6] = two_s * (i * k - j * r);
    m[7] = two_s * (j * k + i * r);
    m[8] = 1.0f - two_s * (i * i + j * j);
}
// unwrap step of refresh_state FA:339-347
TD void unwrap(float rpy, float &old, float &cont) {
    float d = rpy - old;
    d = (d > 1.0f) ? d - kTwoPi : d;
    d = (d < -1.0f) ? d + kTwoPi : d;
    cont = cont + d;
    old = rpy;
}
// CTRL/angvel_control.py:67-88, one axis
// FIN: des, cur, prev and integ are known to be finite (see `fin` in the kernel), so the clamps need no NaN pass-through
template <bool FIN = false> TD float pid_axis(float dt, float rdt, float kp, float des, float cur, float &prev, float &integ) {
    float e = clamp_const_t<FIN>(des - cur, -400.0f, 400.0f);
    float pv = (prev == 0.0f) ? e : prev;
    float P = kp * e;
    float I = clamp_const_t<FIN>(integ + e * dt, -500.0f, 500.0f);
    float I_term = 0.0f * I;
    float deriv = div_const(e - pv, dt, rdt);  // == (e - pv) / dt bit for bit: taco_create verified it for this dt
    float D = clamp_const_t<FIN>(0.5f * deriv, -150.0f, 150.0f);
    float FF = 0.0f * des;
    integ = I;
    prev = e;
    // FIN: I and des are finite, so I_term and FF are +-0 and ((P + I_term) + D) + FF == P + D up to the sign of a zero result -- which no
    // consumer can see: the allocator adds it to a sum that starts from u0 * 1 (never -0), and clamps an all-zero row to 100
    if constexpr (FIN) return 0.4f * (P + D);
    return 0.4f * (P + I_term + D + FF);
}
TD float two_level(float d) { return 1.0f / (1.0f + d * d) + 1.0f / (1.0f + 10.0f * d * d); }

// round(N(0,1)) clamped to +-lim by inverse CDF on one uniform (distribution of FA:324 / FA:576)
TD int rounded_normal(float u, int lim) {
    int k = -3;
    k += (u >= 0.0062096653f);
    k += (u >= 0.0668072013f);
    k += (u >= 0.3085375387f);
    k += (u >= 0.6914624613f);
    k += (u >= 0.9331927987f);
    k += (u >= 0.9937903347f);
    return k < -lim ? -lim : (k > lim ? lim : k);
}

// R(q) u for a unit quaternion by the sandwich product: u + w t + qv x t with t = 2 qv x u
TD V3 quat_sandwich(Q4 q, V3 u) {
    float tx = fma(q.y, u.z, -(q.z * u.y)), ty = fma(q.z, u.x, -(q.x * u.z)), tz = fma(q.x, u.y, -(q.y * u.x));
    tx = tx + tx; ty = ty + ty; tz = tz + tz;
    V3 o;
    o.x = fma(q.y, tz, fma(-q.z, ty, fma(q.w, tx, u.x)));
    o.y = fma(q.z, tx, fma(-q.x, tz, fma(q.w, ty, u.y)));
    o.z = fma(q.x, ty, fma(-q.y, tx, fma(q.w, tz, u.z)));
    return o;
}

// One gym.simulate(dt) for one free rigid body (replaces VT:313 + FA:633-635; spec in DESIGN.md "row I"): `substeps`
// sub-iterations of semi-implicit Euler in the body frame (start value of the body rates: row C's body-frame angular
// velocity wb), closed-form quaternion update q <- normalize(q (x) exp(h/2 b)), body-frame wrench held constant.  The body
// rates stay in the body frame from one simulate() to the next: taken from the root state once per step (row C's formula),
// fed to the rate PID of every substep, turned back into the root state's world-frame angular velocity after the 10th.  The rare forms (|b| h/2 > 0.5 rad per
// sub-iteration, |q|^2 off by > 1e-3, NaN) are evaluated only when some lane of the wavefront needs them (ballot), and
// selected per lane, so the common case has no divergent control flow at all.
// Returns (wave-uniformly) whether some lane took a rare form, i.e. whether the new body rates may be large, infinite or NaN.
// SUBS: the sub-iteration count when it is known at compile time (the PLAIN loop form: 2, unrolled), 0 = P.substeps.
template <int SUBS = 0> TD bool integrate(const Consts &P, int substeps, V3 &p, Q4 &q, V3 &v, V3 &wb, V3 F, V3 tq) {
    float b0 = wb.x, b1 = wb.y, b2 = wb.z;
    bool any_big = false;
    auto iteration = [&]() {
        float L0 = P.J0 * b0, L1 = P.J1 * b1, L2 = P.J2 * b2;
        float g0 = fma(b1, L2, -(b2 * L1));
        float g1 = fma(b2, L0, -(b0 * L2));
        float g2 = fma(b0, L1, -(b1 * L0));
        b0 = fma(P.hJi0, tq.x - g0, b0);
        b1 = fma(P.hJi1, tq.y - g1, b1);
        b2 = fma(P.hJi2, tq.z - g2, b2);
        const V3 RF = quat_sandwich(q, F);
        v.x = fma(P.h, RF.x * P.inv_m, v.x);
        v.y = fma(P.h, RF.y * P.inv_m, v.y);
        v.z = fma(P.h, fma(RF.z, P.inv_m, P.g), v.z);
        p.x = fma(P.h, v.x, p.x);
        p.y = fma(P.h, v.y, p.y);
        p.z = fma(P.h, v.z, p.z);
        const float w2 = fma(b2, b2, fma(b1, b1, b0 * b0));
        const float A2 = (P.half_h * P.half_h) * w2;
        const float sp = fma(fma(fma(fma(2.7557319224e-6f, A2, -1.9841269841e-4f), A2, 8.3333333333e-3f), A2, -1.6666666667e-1f), A2, 1.0f);
        float c = fma(fma(fma(fma(2.4801587302e-5f, A2, -1.3888888889e-3f), A2, 4.1666666667e-2f), A2, -0.5f), A2, 1.0f);
        float k = P.half_h * sp;
        const bool big = !(A2 <= 0.25f);
        if (__builtin_expect(__builtin_amdgcn_ballot_w64(big) != 0, 0)) {
            any_big = true;
            const float wn = __builtin_sqrtf(w2);
            float sn, cs;
            sincos(P.half_h * wn, sn, cs);
            float ks = sn / wn;
            if (A2 != A2) { ks = nanf32(); cs = nanf32(); }
            k = big ? ks : k;
            c = big ? cs : c;
        }
        const float dx = b0 * k, dy = b1 * k, dz = b2 * k;
        const float nx = fma(q.w, dx, fma(q.x, c, fma(q.y, dz, -(q.z * dy))));
        const float ny = fma(q.w, dy, fma(q.y, c, fma(q.z, dx, -(q.x * dz))));
        const float nz = fma(q.w, dz, fma(q.z, c, fma(q.x, dy, -(q.y * dx))));
        const float nw = fma(q.w, c, -fma(q.x, dx, fma(q.y, dy, q.z * dz)));
        const float n2 = fma(nw, nw, fma(nz, nz, fma(ny, ny, nx * nx)));
        float inv = fma(-0.5f, n2, 1.5f);
        const bool off = !(absf(n2 - 1.0f) <= 1e-3f);
        if (__builtin_expect(__builtin_amdgcn_ballot_w64(off) != 0, 0)) {
            const float ie = 1.0f / __builtin_sqrtf(n2);
            inv = off ? ie : inv;
        }
        q.x = nx * inv; q.y = ny * inv; q.z = nz * inv; q.w = nw * inv;
    };
    if constexpr (SUBS == 2) { iteration(); iteration(); }
    else for (int it = 0; it < substeps; ++it) iteration();
    wb = V3{b0, b1, b2};
    return any_big;
}


// ---- quad layout (LPE = 4 lanes per env, the latency regime): inside the substep loop lane j of an env's quad holds
// component j of every vector (x y z [w], rotor j, PID axis j); cross-lane operands come from DPP quad_perm reads, which
// ride on the consuming VALU instruction or cost one v_mov.  Every lane runs the scalar code's operation sequence for its
// component, and every reduction adds in the scalar code's order, so the results are bit-identical to LPE = 1.
constexpr int QP(int a, int b, int c, int d) { return a | (b << 2) | (c << 4) | (d << 6); }
template <int CTRL> TD float dppf(float x) { return as_f(__builtin_amdgcn_update_dpp(0, as_i(x), CTRL, 0xf, 0xf, true)); }
TD float bc0(float x) { return dppf<QP(0, 0, 0, 0)>(x); }
TD float bc1(float x) { return dppf<QP(1, 1, 1, 1)>(x); }
TD float bc2(float x) { return dppf<QP(2, 2, 2, 2)>(x); }
TD float bc3(float x) { return dppf<QP(3, 3, 3, 3)>(x); }
TD float rot1(float x) { return dppf<QP(1, 2, 0, 3)>(x); }  // lane j <- component (j + 1) mod 3
TD float rot2(float x) { return dppf<QP(2, 0, 1, 3)>(x); }  // lane j <- component (j + 2) mod 3
TD float swp(float x) { return dppf<QP(1, 0, 3, 2)>(x); }   // neighbour inside the pairs (0,1) (2,3)
TD float xorf(float x, uint32_t m) { return from_bits(bits(x) ^ m); }
// lane `sub` of the quad takes operand `sub`.  Bitwise on purpose (two lane masks, three v_bfi_b32): written as a chain of ternaries the compiler
// turned every call into CONTROL FLOW -- exec-mask if / else blocks, ~18 instructions and three branches each; the ~25 calls on the step wavefront's
// path ahead of its loop (slot table, scatter into the quad layout) were ~450 of its instructions.
TD float pick4(int sub, float a, float b, float c, float d) {
    const uint32_t m1 = 0u - (uint32_t)(sub & 1), m2 = 0u - (uint32_t)((sub >> 1) & 1);
    const uint32_t lo = (bits(b) & m1) | (bits(a) & ~m1), hi = (bits(d) & m1) | (bits(c) & ~m1);
    return from_bits((hi & m2) | (lo & ~m2));
}
// select by a precomputed lane mask (all ones: take a): one v_bfi_b32, no VCC round trip, never a branch
TD float selm(uint32_t m, float a, float b) { return from_bits((bits(a) & m) | (bits(b) & ~m)); }
// component j of TU:58-68 quat_rotate((qv, w), v); qv / v hold components 0..2 in lanes 0..2
TD float quad_rotate(float qv, float w, float v) {
    const float s = 2.0f * (w * w) - 1.0f;
    const float c = cross_term(rot1(qv), rot2(v), rot2(qv), rot1(v));
    const float pr = qv * v;
    // (pr0 + pr1) + pr2 in lane 0 (a + b == b + a exactly), then broadcast: two DPP reads ride on the adds
    const float dot = bc0((rot1(pr) + pr) + rot2(pr));
    return v * s + c * w * 2.0f + qv * dot * 2.0f;
}
// row I (integrate above) in the quad layout.  qq = (x y z w) over the four lanes; pq vq bq Fq tqq components 0..2 in lanes 0..2.
// Jq / hJiq: this lane's inertia terms; gzq = (-0, -0, g, -0) (fma(RF, 1/m, -0) == RF * (1/m) exactly); sm3 = sign bit in lane 3,
// k3 = all ones in lane 3.
// the linear lines of one sub-iteration of row I in the quad layout: v <- v + h (R(q) F / m + g), p <- p + h v, with q the attitude BEFORE the
// sub-iteration's update (lane j = component j; gzq = (-0, -0, g, -0)).  Used by integrate_quad and, in the four-role quad forms, by the
// wavefront that runs the linear chain behind the step wavefront (step_core: "linear chain served"): one definition, the same bits.
// The integrator's step constants as the quad forms' loops take them: PINNED in scalar registers (pin_s).  Read through the parameter block's
// reference they are loads the register allocator prefers to REPEAT inside the substep loop over keeping them (an invariant load is trivially
// rematerialisable) -- a scalar-cache round trip per use per substep on the path of a wavefront that has nothing else to issue.
struct QuadLoopP { float h, half_h, inv_m; int substeps; };
TD float pin_s(float x) { asm volatile("" : "+s"(x)); return x; }
template <class PT> TD void linear_iteration_quad(const PT &P, float qq, float Fq, float gzq, float &vq, float &pq) {
    const float ww = bc3(qq);
    const float q1 = rot1(qq), q2 = rot2(qq);
    float t = fma(q1, rot2(Fq), -(q2 * rot1(Fq)));
    t = t + t;
    const float RF = fma(q1, rot2(t), fma(-q2, rot1(t), fma(ww, t, Fq)));
    vq = fma(P.h, fma(RF, P.inv_m, gzq), vq);
    pq = fma(P.h, vq, pq);
}
// NOLIN: the angular lines only (pq, vq, Fq are not touched); q_mid <- the attitude after the FIRST sub-iteration (what the second one's
// linear lines rotate the force with) -- the linear chain runs elsewhere, see linear_iteration_quad.
template <int SUBS = 0, bool DEFER = false, bool NOLIN = false, class PT = StepParams>
TD bool integrate_quad(const PT &P, uint32_t k3, float &pq, float &qq, float &vq, float &bq, float Fq, float tqq, float Jq, float hJiq,
                       float gzq, uint32_t sm3, float &q_mid) {
    bool any_big = false;
    uint64_t bad = 0;   // DEFER: the lanes that met a rare form (nothing was branched on; the caller redoes the call with DEFER = false)
    int it_no = 0;
    auto iteration = [&]() {
        const float L = Jq * bq;
        const float g = fma(rot1(bq), rot2(L), -(rot2(bq) * rot1(L)));
        bq = fma(hJiq, tqq - g, bq);
        const float ww = bc3(qq);
        if constexpr (!NOLIN) linear_iteration_quad(P, qq, Fq, gzq, vq, pq);
        const float sqb = bq * bq, b1 = bc1(bq), b2 = bc2(bq);
        const float w2 = fma(b2, b2, fma(b1, b1, bc0(sqb)));
        const float A2 = (P.half_h * P.half_h) * w2;
        const float sp = fma(fma(fma(fma(2.7557319224e-6f, A2, -1.9841269841e-4f), A2, 8.3333333333e-3f), A2, -1.6666666667e-1f), A2, 1.0f);
        float c = fma(fma(fma(fma(2.4801587302e-5f, A2, -1.3888888889e-3f), A2, 4.1666666667e-2f), A2, -0.5f), A2, 1.0f);
        float k = P.half_h * sp;
        const bool big = !(A2 <= 0.25f);
        if constexpr (DEFER) bad |= __builtin_amdgcn_ballot_w64(big);
        else if (__builtin_expect(__builtin_amdgcn_ballot_w64(big) != 0, 0)) {
            any_big = true;
            const float wn = __builtin_sqrtf(w2);
            float sn, cs;
            sincos(P.half_h * wn, sn, cs);
            float ks = sn / wn;
            if (A2 != A2) { ks = nanf32(); cs = nanf32(); }
            k = big ? ks : k;
            c = big ? cs : c;
        }
        // n = q (x) (d, c): lanes 0..2  fma(w, d_j, fma(q_j, c, fma(q_j+1, d_j+2, -(q_j+2 * d_j+1))));
        // lane 3  fma(w, c, -fma(x, dx, fma(y, dy, z * dz))) == fma(w, c, fma(-x, dx, fma(-y, dy, -(z * dz))))  (same shape)
        const float d4 = selm(k3, c, bq * k);
        const float Qn = xorf(dppf<QP(0, 1, 2, 0)>(qq), sm3), Cc = dppf<QP(3, 3, 3, 0)>(d4);
        const float Qa = xorf(dppf<QP(1, 2, 0, 1)>(qq), sm3), Db = dppf<QP(2, 0, 1, 1)>(d4);
        const float Qb = dppf<QP(2, 0, 1, 2)>(qq), Da = dppf<QP(1, 2, 0, 2)>(d4);
        const float n = fma(ww, d4, fma(Qn, Cc, fma(Qa, Db, -(Qb * Da))));
        const float nsq = n * n, n1 = bc1(n), n2_ = bc2(n), n3 = bc3(n);
        const float n2 = fma(n3, n3, fma(n2_, n2_, fma(n1, n1, bc0(nsq))));
        float inv = fma(-0.5f, n2, 1.5f);
        const bool off = !(absf(n2 - 1.0f) <= 1e-3f);
        if constexpr (DEFER) bad |= __builtin_amdgcn_ballot_w64(off);
        else if (__builtin_expect(__builtin_amdgcn_ballot_w64(off) != 0, 0)) {
            const float ie = 1.0f / __builtin_sqrtf(n2);
            inv = off ? ie : inv;
        }
        qq = n * inv;
        if (it_no++ == 0) q_mid = qq;
    };
    if constexpr (SUBS == 2) { iteration(); iteration(); }
    else for (int it = 0; it < P.substeps; ++it) iteration();
    return DEFER ? bad != 0 : any_big;
}

// reset_idx for one env (FA:475-517), in the reference's call order: reset_copter_idx -> reset_controller_idx ->
// reset_env_idx -> reset_target_idx.  The chunks the step evolves come back IN REGISTERS (the caller's up-front loads are overwritten
// for the lanes that reset: no store -> reload round trip, and the three values a reset keeps -- angular velocity of a flip env, command,
// aero parameters -- are taken from those loads instead of being fetched again: round 1 paid three dependent memory round trips in every
// wavefront that held a reset lane, i.e. in half of them at the benchmark's 1.1 % resets per step); the chunks the step only reads
// (target pose, unwrapped angles, rotor / aero parameters) are stored here, by the lane that stores for its env.  The 37 uniforms come
// from 10 Philox blocks of STREAM_RESET; a block whose consumers are all switched off (launch-uniform flags) is not generated.
#define CST(c, val) buf_st4(rS, (val), voff, (uint32_t)(c) * row_bytes)
#define CLD(c) buf_ld4(rS, voff, (uint32_t)(c) * row_bytes)
// LPE == 4 (quad layout: the four lanes of an env run this redundantly): the Philox blocks are SHARED OUT -- lane `sub` evaluates blocks
// sub, 4 + sub, 8 + sub (three Philox evaluations per lane instead of up to ten) and the quad reads each block's four uniforms from the
// lane that made them (DPP quad_perm broadcasts).  Same numbers, ~200 instructions fewer on the path of every wavefront that holds a
// resetting env -- at 4 096 envs that is some wavefront of nearly every launch, i.e. the launch time.
// which STREAM_RESET blocks a configuration draws from (launch-uniform), as a bit mask
TD uint32_t reset_blocks_wanted(uint32_t fl) {
    const bool rc = (fl & TACO_F_RANDOM_ROTORDYNAMIC_COE) != 0, ra = (fl & TACO_F_RANDOM_AERODYNAMIC_COE) != 0;
    const bool rtau = (fl & TACO_F_ROTOR_RESPONSE) && (fl & TACO_F_RANDOM_ROTOR_RESPONSE), rspd = (fl & TACO_F_RANDOM_ROTOR_SPEED) != 0;
    const bool rquat = (fl & TACO_F_RANDOM_COPTER_QUAT) != 0, rvel = (fl & TACO_F_RANDOM_COPTER_VEL) != 0;
    return 1u | ((rquat || rvel) ? 2u : 0u) | (rvel ? 4u : 0u) | ((rvel || (fl & TACO_F_RANDOM_TARGET_POS)) ? 8u : 0u) |
           (((fl & (TACO_F_RANDOM_TARGET_YAW | TACO_F_RANDOM_VOLTAGE)) || rc) ? 16u : 0u) | ((rc || rtau) ? 32u : 0u) |
           ((rtau || rspd) ? 64u : 0u) | ((rspd || ra) ? 128u : 0u) | (ra ? 256u : 0u) | ((fl & TACO_F_RANDOM_DELAY_TIME) ? 512u : 0u);
}
// one Philox block as its four uniforms
TD float4 uniform_block(const StepParams &P, uint32_t step, int gid, uint32_t stream, uint32_t ctr) {
    const U4 r = philox(P.seed_lo, P.seed_hi, (uint32_t)gid, step, stream, ctr);
    return make_float4(uniform(r.x), uniform(r.y), uniform(r.z), uniform(r.w));
}
// this lane's block of round `rnd` of the quad layout's share-out (block 4 rnd + sub)
TD float4 reset_block_quad(const StepParams &P, uint32_t step, int gid, int rnd, int sub) { return uniform_block(P, step, gid, STREAM_RESET, (uint32_t)(4 * rnd + sub)); }
// `drawn` (LPE == 4, optional): the three rounds' blocks, [rnd * 64 + lane], drawn ahead by the role wavefronts (step_core: "reset draws")
// `drawn_shared` (LPE == 1, wave-uniform; a flag, not a null test: the wavefront's scratch may sit at LDS address 0): the env's blocks 0..15 are at drawn[drawn_row * 16 + b], drawn by sixteen lanes of the wavefront
// at once (step_core: "shared reset draws")
// FL: the randomisation flags this build of the function is for, or ~0u = read them from the handle.  A resetting env puts ~700 issue slots on
// the path of the wavefront that decides most launches' time, a good third of them the tests and selects of the dozen flags reset_idx consults
// (launch-uniform, but run-time: a scalar branch around every few instructions) -- the four-role quad forms call a build with the flags of the
// documented configurations compiled in when the handle's flags are those (kResetFlagsBaseline: BASELINE configs 2-4; kResetFlagsAll: config 5).
constexpr uint32_t kResetFlagMask = TACO_F_RANDOM_COPTER_POS | TACO_F_RANDOM_COPTER_QUAT | TACO_F_RANDOM_COPTER_VEL | TACO_F_RANDOM_TARGET_POS | TACO_F_RANDOM_TARGET_YAW |
                                    TACO_F_RANDOM_VOLTAGE | TACO_F_ROTOR_RESPONSE | TACO_F_RANDOM_ROTORDYNAMIC_COE | TACO_F_RANDOM_ROTOR_RESPONSE | TACO_F_RANDOM_ROTOR_SPEED |
                                    TACO_F_RANDOM_AERODYNAMIC_COE | TACO_F_RANDOM_DELAY_TIME;
constexpr uint32_t kResetFlagsBaseline = TACO_F_RANDOM_COPTER_POS | TACO_F_RANDOM_COPTER_QUAT | TACO_F_RANDOM_COPTER_VEL | TACO_F_RANDOM_TARGET_POS | TACO_F_RANDOM_TARGET_YAW |
                                         TACO_F_RANDOM_VOLTAGE | TACO_F_ROTOR_RESPONSE | TACO_F_RANDOM_ROTOR_SPEED;
constexpr uint32_t kResetFlagsAll = kResetFlagMask;
template <int LPE, uint32_t FL = ~0u>
TD void reset_env(const StepParams &P, uint32_t step, rsrc_t rS, uint32_t voff, uint32_t row_bytes, int gid, int grp, bool mix, bool store, int sub,
                  float4 &c_pos, float4 &c_quat, float4 &c_lin, float4 &c_ang, float4 &c_pp, float4 &c_pi, float4 &c_om, float4 &c_misc, float4 &c_tau,
                  float4 &c_op, float4 &c_a0, float4 &c_a1, const __attribute__((address_space(3))) f32x4_t *drawn = nullptr, bool drawn_shared = false, int drawn_row = 0) {
    const uint32_t fl = FL == ~0u ? P.flags : ((P.flags & ~kResetFlagMask) | (FL & kResetFlagMask));
    const bool rc = (fl & TACO_F_RANDOM_ROTORDYNAMIC_COE) != 0, ra = (fl & TACO_F_RANDOM_AERODYNAMIC_COE) != 0;
    const bool rtau = (fl & TACO_F_ROTOR_RESPONSE) && (fl & TACO_F_RANDOM_ROTOR_RESPONSE), rspd = (fl & TACO_F_RANDOM_ROTOR_SPEED) != 0;
    const bool rpos = (fl & TACO_F_RANDOM_COPTER_POS) != 0, rquat = (fl & TACO_F_RANDOM_COPTER_QUAT) != 0, rvel = (fl & TACO_F_RANDOM_COPTER_VEL) != 0;
    const uint32_t want = reset_blocks_wanted(fl);   // (FL: a constant)
    float qu[3][4] = {{0.0f, 0.0f, 0.0f, 0.0f}, {0.0f, 0.0f, 0.0f, 0.0f}, {0.0f, 0.0f, 0.0f, 0.0f}};  // LPE == 4: this lane's uniforms of blocks sub, 4 + sub, 8 + sub
    if constexpr (LPE == 4) {
#pragma unroll
        for (int rnd = 0; rnd < 3; ++rnd) {
            if ((want >> (4 * rnd)) & 15u) {  // (launch-uniform: some lane's block of this round is drawn from)
                float4 u;
                if (drawn) { const f32x4_t d = drawn[rnd * 64 + (int)(threadIdx.x & 63)]; u = make_float4(d.x, d.y, d.z, d.w); }
                else u = reset_block_quad(P, step, gid, rnd, sub);
                qu[rnd][0] = u.x; qu[rnd][1] = u.y; qu[rnd][2] = u.z; qu[rnd][3] = u.w;
            }
        }
    }
    auto from_lane = [](float v, int k) {  // the value lane k of this quad holds
        const int iv = __builtin_bit_cast(int, v);
        const int o = k == 0 ? __builtin_amdgcn_update_dpp(0, iv, 0x00, 0xf, 0xf, true) : k == 1 ? __builtin_amdgcn_update_dpp(0, iv, 0x55, 0xf, 0xf, true)
                    : k == 2 ? __builtin_amdgcn_update_dpp(0, iv, 0xaa, 0xf, 0xf, true) : __builtin_amdgcn_update_dpp(0, iv, 0xff, 0xf, 0xf, true);
        return __builtin_bit_cast(float, o);
    };
    auto block = [&](bool wanted, uint32_t b, float &u0, float &u1, float &u2, float &u3) {
        u0 = u1 = u2 = u3 = 0.0f;
        if (wanted) {  // (launch-uniform)
            if constexpr (LPE == 4) {
                const int rnd = (int)(b >> 2), k = (int)(b & 3u);
                u0 = from_lane(qu[rnd][0], k); u1 = from_lane(qu[rnd][1], k); u2 = from_lane(qu[rnd][2], k); u3 = from_lane(qu[rnd][3], k);
            } else if (drawn_shared) {  // (wave-uniform) LPE == 1, few lanes reset: the wavefront drew this env's blocks together (step_core: "shared reset draws")
                const f32x4_t d = drawn[drawn_row * 16 + (int)b];
                u0 = d.x; u1 = d.y; u2 = d.z; u3 = d.w;
            } else {
                U4 r = philox(P.seed_lo, P.seed_hi, (uint32_t)gid, step, STREAM_RESET, b);
                u0 = uniform(r.x); u1 = uniform(r.y); u2 = uniform(r.z); u3 = uniform(r.w);
            }
        }
    };
    float u0, u1, u2, u3, u4, u5, u6, u7;
    const float pi_sc = (float)(3.14159265358979323846 - (-3.14159265358979323846)), pi_lo = (float)(-3.14159265358979323846);
    // ---- reset_copter_idx: FA:725-756 pos, :783-812 rotate, :850-884 flip, :981-1056 mix (pos-style ranges for all groups)
    // (flip and rotate draw x, y even without random_copter_pos)
    block(true, 0, u0, u1, u2, u3);                 // uniforms 0..3: pos x y z, euler a
    block(rquat || rvel, 1, u4, u5, u6, u7);        // uniforms 4..7: euler b c, linvel x y
    V3 p;
    if (grp == TACO_TASK_FLIP && !mix) {
        if (rpos) {
            p.x = P.flip_xy_sc * u0 + P.flip_xy_lo;
            p.y = P.flip_xy_sc * u1 + P.flip_xy_lo;
            p.z = 3.0f + P.df * (4.0f * u2 + -2.0f);
        } else {
            p.x = 1.0f * u0 + -0.5f;
            p.y = 1.0f * u1 + -0.5f;
            p.z = 3.0f;
        }
    } else if (grp == TACO_TASK_ROTATE && !mix && !rpos) {
        p.x = 1.0f * u0 + -0.5f;
        p.y = 1.0f * u1 + -0.5f;
        p.z = 2.5f;
    } else if (rpos) {
        p.x = 4.0f * u0 + -2.0f;
        p.y = 4.0f * u1 + -2.0f;
        p.z = 2.5f + (4.0f * u2 + -2.0f);
    } else {
        p.x = 0.0f; p.y = 0.0f; p.z = 2.5f;
    }
    c_pos = make_float4(p.x, p.y, p.z, as_f((int)make_pw(0, 0)));  // progress <- 0 (FA:510-511), no runs queued
    float c0, c1, c2, c3;
    block((fl & (TACO_F_RANDOM_TARGET_YAW | TACO_F_RANDOM_VOLTAGE)) || rc, 4, c0, c1, c2, c3);  // uniforms 16..19: target yaw, battery E, omega_para 0 1
    const float tgt_yaw = (fl & TACO_F_RANDOM_TARGET_YAW) ? pi_sc * c0 + pi_lo : 0.0f;   // reset_target_idx FA:523-548
    Q4 q{0.0f, 0.0f, 0.0f, 1.0f}, qt;
    V3 rpy0;
    if constexpr (LPE == 4) {
        // the quad shares out the transcendental functions as well: lane j takes the sine / cosine of half of angle j (roll, pitch, yaw of
        // rand_quat, and the target's yaw in lane 3), and of get_euler_xyz_v1 lane 0 the roll atan2, lane 2 the yaw atan2, lane 1 the pitch
        // asin (every lane runs one atan2 and one asin on its own operands).  Same functions on the same arguments: same bits.
        const float l_sc = (grp == TACO_TASK_FLIP) ? 0.0f : pi_sc;
        const float l_lo = (grp == TACO_TASK_FLIP) ? -0.0f : pi_lo;
        const float a_r = pi_sc * u3 + pi_lo, a_p = l_sc * u4 + l_lo, a_y = l_sc * u5 + l_lo;
        const float ang = sub == 0 ? a_r : (sub == 1 ? a_p : (sub == 2 ? a_y : tgt_yaw));
        float sn, cs;
        sincos(ang * 0.5f, sn, cs);
        if (rquat)  // rand_quat FA:698-704 (flip: limits (pi, 0, 0))
            q = quat_from_sincos(from_lane(sn, 0), from_lane(cs, 0), from_lane(sn, 1), from_lane(cs, 1), from_lane(sn, 2), from_lane(cs, 2));
        float s0, c0_;
        sincos(0.0f * 0.5f, s0, c0_);  // (constant-folded) what quat_from_euler(0, 0, yaw) takes for its roll and pitch
        qt = quat_from_sincos(s0, c0_, s0, c0_, from_lane(sn, 3), from_lane(cs, 3));
        const float n0 = 2.0f * (q.w * q.x + q.y * q.z), d0 = q.w * q.w - q.x * q.x - q.y * q.y + q.z * q.z;
        const float sinp = 2.0f * (q.w * q.y - q.z * q.x);
        const float n2 = 2.0f * (q.w * q.z + q.x * q.y), d2 = q.w * q.w + q.x * q.x - q.y * q.y - q.z * q.z;
        const float at = atan2(sub == 2 ? n2 : n0, sub == 2 ? d2 : d0);
        float pit;
        if (absf(sinp) >= 1.0f) pit = kHalfPi * (sinp > 0.0f ? 1.0f : (sinp < 0.0f ? -1.0f : 0.0f));
        else pit = asin(sinp);
        rpy0 = V3{from_lane(at, 0), from_lane(pit, 1), from_lane(at, 2)};  // FA:752-754
    } else {
        if (rquat) {  // rand_quat FA:698-704 (flip: limits (pi, 0, 0))
            const float l_sc = (grp == TACO_TASK_FLIP) ? 0.0f : pi_sc;
            const float l_lo = (grp == TACO_TASK_FLIP) ? -0.0f : pi_lo;
            q = quat_from_euler(pi_sc * u3 + pi_lo, l_sc * u4 + l_lo, l_sc * u5 + l_lo);
        }
        qt = quat_from_euler(0.0f, 0.0f, tgt_yaw);
        rpy0 = euler_xyz_v1(q);  // FA:752-754
    }
    c_quat = make_float4(q.x, q.y, q.z, q.w);
    if (store) {
        CST(C_RPY_OLD, make_float4(rpy0.x, rpy0.y, rpy0.z, 0.0f));
        CST(C_RPY_CONT, make_float4(rpy0.x, rpy0.y, rpy0.z, 0.0f));
    }
    float a0, a1, a2, a3, b0, b1, b2, b3;
    block(rvel, 2, a0, a1, a2, a3);                                          // uniforms 8..11: linvel z, angvel x y z
    block(rvel || (fl & TACO_F_RANDOM_TARGET_POS), 3, b0, b1, b2, b3);       // uniforms 12..15: flip sign, target x y z
    float d0, d1, d2, d3;
    block(rc || rtau, 5, d0, d1, d2, d3);                                    // uniforms 20..23: omega_para 2 3 4, tau 0
    float h0, h1, h2, h3;
    block((fl & TACO_F_RANDOM_DELAY_TIME) != 0, 9, h0, h1, h2, h3);          // uniform 36: delay length
    // ---- reset_env_idx FA:560-581: delay length
    int L = P.delay_time;
    if (fl & TACO_F_RANDOM_DELAY_TIME) {
        L = P.delay_time - rounded_normal(h0, 3);
        L = L < 0 ? 0 : L;
    }
    {
        V3 v{0.0f, 0.0f, 0.0f}, w{0.0f, 0.0f, 0.0f};
        if (grp == TACO_TASK_FLIP) {
            w = V3{c_ang.x, c_ang.y, c_ang.z};  // standalone FpvFlip leaves angvel untouched unless told otherwise (FA:876-878)
            if (rvel) {
                v = V3{P.flip_v_sc * u6 + P.flip_v_lo, P.flip_v_sc * u7 + P.flip_v_lo, P.flip_v_sc * a0 + P.flip_v_lo};
                w.x = 10.0f * (b0 < 0.5f ? -1.0f : 1.0f);  // w.y, w.z keep their values (FA:876, :1047)
            } else if (mix) {
                w = V3{0.0f, 0.0f, 0.0f};  // FA:1050
            }
        } else if (rvel) {
            v = V3{3.0f * (2.0f * u6 + -1.0f), 3.0f * (2.0f * u7 + -1.0f), 3.0f * (2.0f * a0 + -1.0f)};
            w = V3{3.0f * (2.0f * a1 + -1.0f), 3.0f * (2.0f * a2 + -1.0f), 3.0f * (2.0f * a3 + -1.0f)};
        }
        c_lin = make_float4(v.x, v.y, v.z, as_f((int)make_dw(L, L, 0, 0, 0, false)));  // Z = L: the whole pending line is reset-zeros
                                                                                        // (FA:572-574), the run queue empty
        c_ang = make_float4(w.x, w.y, w.z, 0.0f);  // battery_voltage <- 0 (FA:566)
    }
    // ---- reset_target_idx FA:523-548
    if (store) {
        if (fl & TACO_F_RANDOM_TARGET_POS) CST(C_TGT_POS, make_float4(P.df * (4.0f * b1 + -2.0f), P.df * (4.0f * b2 + -2.0f), 3.0f + P.df * (4.0f * b3 + -2.0f), 0.0f));
        else CST(C_TGT_POS, make_float4(0.0f, 0.0f, 3.0f, 0.0f));
    }
    if (store) CST(C_TGT_QUAT, make_float4(qt.x, qt.y, qt.z, qt.w));
    // ---- reset_controller_idx FA:550-558
    c_pp = make_float4(0.0f, 0.0f, 0.0f, (fl & TACO_F_RANDOM_VOLTAGE) ? (float)2.2 * c1 + 0.0f : 0.0f);  // + battery E_c
    c_pi = make_float4(0.0f, 0.0f, 0.0f, 0.0f);                                                          // + battery u_1
    c_misc = make_float4(0.0f, c_misc.y, c_misc.z, c_misc.w);  // battery time <- 0; command / flip_radian survive a reset until
                                                               // reset_command_idx rewrites them
    c_op = make_float4(rc ? 0.0f * (P.dr_sc * c2 + P.dr_lo) : 0.0f, rc ? 12.9466f * (P.dr_sc * c3 + P.dr_lo) : 12.9466f,
                       rc ? 0.1872f * (P.dr_sc * d0 + P.dr_lo) : 0.1872f, rc ? -5.1220f * (P.dr_sc * d1 + P.dr_lo) : -5.1220f);
    float e0, e1, e2, e3, f0, f1, f2, f3, g0, g1, g2, g3;
    block(rtau || rspd, 6, e0, e1, e2, e3);  // uniforms 24..27: tau 1 2 3, omega0 0
    block(rspd || ra, 7, f0, f1, f2, f3);    // uniforms 28..31: omega0 1 2 3, cf
    block(ra, 8, g0, g1, g2, g3);            // uniforms 32..35: ct, dx, dy, kt
    {
        const float tu[4] = {d3, e0, e1, e2};
        const float ou[4] = {e3, f0, f1, f2};
        float t[4], o[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            if (fl & TACO_F_ROTOR_RESPONSE) t[k] = (fl & TACO_F_RANDOM_ROTOR_RESPONSE) ? P.tau_sc * tu[k] + P.tau_lo : P.tau_fixed * 1.0f;
            else t[k] = 0.001f * 1.0f;
            o[k] = rspd ? 400.0f * ou[k] + 0.0f : 0.0f;
        }
        c_tau = make_float4(t[0], t[1], t[2], t[3]);
        c_om = make_float4(o[0], o[1], o[2], o[3]);
    }
    {   // aero parameters keep their values unless re-randomised: CTRL/thrust_dynamics.py:201-210 (a no-op when off)
        const float4 ae0 = c_a0, ae1 = c_a1;
        c_a0 = make_float4(rc ? 0.5906f * (P.dr_sc * d2 + P.dr_lo) : 0.5906f, ra ? 1.13e-05f * (P.dr_sc * f3 + P.dr_lo) : ae0.y,
                           ra ? 0.05f * (P.dr_sc * g0 + P.dr_lo) : ae0.z, ra ? -0.386f * (P.dr_sc * g1 + P.dr_lo) : ae0.w);
        c_a1 = make_float4(ra ? -0.53f * (P.dr_sc * g2 + P.dr_lo) : ae1.x, ra ? 0.009f * (P.dr_sc * g3 + P.dr_lo) : ae1.y, 0.0f, 0.0f);
    }
    if (store && !(P.flags & kUniformParams)) {  // the parameter chunks are only ever written here -- and not even here while they hold the same nominal
        // values in every env (kUniformParams: nothing randomises them, init_state_kernel wrote what this would write, nobody loads them):
        // four 16-byte stores of a lone lane less per reset (partial lines: each a read-modify-write somewhere below L2)
        CST(C_TAU, c_tau); CST(C_OPARA, c_op); CST(C_AERO0, c_a0); CST(C_AERO1, c_a1);
    }
    // the ring is NOT zeroed (FA:574): the delay-line word marks every pending slot as an implied zero
}


// Instantiations (host side: choose_form in taco_capi.hip): one wavefront per workgroup everywhere except the four-role forms (a workgroup
// is then replaced on its CU as soon as ITS wavefront is done, instead of waiting for the slowest of four: 175 vs 182 us at 1 M envs);
// registers unconstrained in the latency regime, capped at 128 VGPRs (CAP) in the throughput regime so that 4 wavefronts per SIMD hide
// each other's dependent-issue latency and memory waits.
// LPE (lanes per env) = 4 for launches that cannot fill the chip with one lane per env (4 096 envs = 64 wavefronts on 1 024 SIMDs):
// a wavefront then carries 16 envs; everything outside the substep loop runs the scalar code redundantly in the four lanes of an env
// (sub-lane 0 does the stores), the substep loop runs in the quad layout above.  SPLIT: the four-role form, see the kernel.
// Battery_Dynamics.sim_process CTRL/battery_dynamics.py:47-75 for one substep (Pm = mechanical power FA:614)
TD float battery_step(float dt, float Pm, float &bat_E, float &bat_u1, float &bat_t) {
    bat_t = bat_t + dt;
    float p_c = TACO_DIVC(TACO_DIVC(Pm, 0.75f), 9000.0f);
    bat_E = bat_E + p_c * dt;
    float P_avg = bat_E / bat_t;
    float r0_ = 0.0015778f + -7.7608e-5f * P_avg + (float)(0.0069498 * 1500.0);
    float r0 = (r0_ > 4.5f) ? r0_ : 4.5f;
    float uo = 4.35f + -0.1102178f * bat_E + 0.0103368f * (bat_E * bat_E) + -4.3778e-4f * ((bat_E * bat_E) * bat_E);
    float u1_dot = TACO_DIVC(0.00104846f * p_c - bat_u1, 3.3f);
    bat_u1 = bat_u1 + u1_dot * dt;
    float dd = uo - bat_u1;
    float rad = dd * dd - 4.0f * r0 * p_c;
    return 0.5f * (dd + __builtin_sqrtf(rad)) * 6.0f;
}
// What the post-phase (frame stacks, reward / done) needs from the rest of the step.
struct Carry {
    V3 p, v, w;
    Q4 q;
    float bat_V, act[4], cmd0, cmd1, flip_radian;
    int progress;
    float4 c_tp, c_tq;
};
constexpr int CARRY_WORDS = 32;  // 30 used

// SPLIT (BLOCK = 256; LPE = 4 up to 8 192 envs -- 16 384 with frame stacks --, LPE = 1 for 16 385 ... 65 536 envs with frame stacks):
// wave 0 of the workgroup runs the step as the one-wavefront instantiation does; the other three wavefronts wait at a barrier until the
// substeps are done, take the Carry out of LDS and run one post-phase role each (1: reward + done, 2: obs frame stack, 3: states frame
// stack) on their own SIMDs while wave 0 stores the state, so the post-phase critical path is the longest part instead of the sum.  While
// they wait, waves 2 and 3 move the history of the frame stacks (which does not depend on this step) and wave 1 serves the battery model
// (launches of at most 4 096 envs, where every wavefront has a SIMD to itself).
// CAP (LPE = 1, BLOCK = 64, not SPLIT): the throughput instantiation -- at most 128 VGPRs, so that four wavefronts share a SIMD, and
// only the loop forms that fit that budget.
// OUT: the instantiation that also writes the clamped copies VecTask.step returns (obs_out / states_out).  A template parameter, not a
// run-time switch: the copy code sits in the role wavefronts' post-phase, the critical path of the 4 096-env launch, where its mere
// presence cost 0.75 us of 13.8 (SGPR spills, skipped branches) -- profiles/r02_d_ab_out_copy_paths.txt.
// RESET_ONLY (<64, 1> only; taco_reset_done = VecTask.reset_done, VT:363-375): the kernel stops after reset_idx / reset_command_idx, stores what
// they changed, clears the flags of the envs it reset and returns -- no action is pushed, no substep runs, the clock does not advance.
// WIDE (<256, 1, true>): likewise for the one-lane role form while its launch puts at most two wavefronts on a SIMD (<= 32 768 envs): no register
// cap, no spills (the capped build has 9 VGPR spills).
// WIDE (<256, 4, true>): the same four-role quad form compiled for two wavefronts per SIMD instead of four (up to 256 VGPRs): used
// while every wavefront of the launch has a SIMD to itself anyway (<= 4 096 envs, no frame stacks), where the 128-register cap of the
// general form buys nothing and costs 2 % (12.90 vs 13.19 us at 4 096 envs; with stacks or from 8 192 envs on the capped form is faster).
// FUSED (taco_rollout.hpp, taco_rollout_kernel): the same step as the body of a PERSISTENT kernel whose workgroup (16 envs, quad layout, the
// four roles) also holds the actor -- four more wavefronts that compute this step's action from the previous step's observation WHILE the
// step wavefront runs the substeps: with at least ten pending slots the ten slots a step consumes are older actions (delay_time >= 10 ms), so
// the step only needs its own action for the bookkeeping after the loop (history row, frame words 19..22).  FusedCtx carries what the
// enclosing kernel provides instead of the launch: the role of this wavefront, the step clock, the actor's action mailbox and the LDS
// buffer the next observation goes to.  Everything else -- every arithmetic operation -- is the code of the launch-per-step forms.
struct FusedCtx {
    int role;            // 0 step wavefront, 1 battery server + reward, 2 obs frame, 3 states frame (what `threadIdx.x >> 6` is in the SPLIT forms)
    uint32_t step;       // the clock of this step
    int head, hh;
    lds_f32 *act_lds;    // [16][4]: the action the actor sampled for this step, clipped to the policy's limits (ppo_asymmetry.py:310)
    lds_i32 *act_seq;    // >= act_want once act_lds holds this step's action
    int act_want;
    lds_f32 *xin;        // [16][xin_ld]: the actor's input rows; the obs role leaves the new (noised) frame there
    int xin_ld;
    lds_i32 *reset_lds;  // [16]: the done flags the reward role computed in the PREVIOUS step of this kernel (the step wavefront starts the next
    lds_i32 *reset_seq;  //   step while the roles are still in their post-phase: reset_buf in memory may not have been written yet);
    int reset_want;      //   valid once *reset_seq >= reset_want.  reset_want == 0 (first step of the kernel): the flags are read from reset_buf.
};
template <int BLOCK, int LPE, bool SPLIT, bool CAP, bool OUT, bool RESET_ONLY, bool WIDE, bool FUSED, bool LIN = false>
TD void step_core(const StepParams &Pin, const FusedCtx &FX) {
    static_assert(!LIN || ((BLOCK == 256 || BLOCK == 128) && LPE == 4 && SPLIT), "LIN is a variant of the four-role quad form and of the served pair");
    static_assert(!FUSED || (BLOCK == 256 && SPLIT && !OUT && !RESET_ONLY && !CAP), "FUSED is a four-role form (quad, or one lane per env) inside the rollout kernel");
    static_assert(!WIDE || (BLOCK == 256 && SPLIT && !RESET_ONLY), "WIDE is a variant of the four-role forms");
    static_assert(!RESET_ONLY || (BLOCK == 64 && LPE == 1 && !SPLIT && !CAP && !OUT), "RESET_ONLY exists in the plain one-lane form only");
    static_assert(!SPLIT || BLOCK == 256 || (BLOCK == 128 && LPE == 4 && !WIDE && !FUSED),
                  "SPLIT is the four-role form (one step wavefront + three role wavefronts per workgroup) or the two-wavefront SERVED quad form");
    // ROLES: the four-role forms.  PAIR (<128, 4, true>): the SERVED quad form -- wavefront 0 runs the WHOLE step as the one-wavefront quad form does
    // (post-phase included), wavefront 1 only serves it (rotor-noise table, euler angles) and leaves.  For launches of 8 193 ... 16 384 envs
    // (one step wavefront per SIMD either way) whose flags give the servers work: the role forms' barriers and hand-over cost more there
    // than their parallel post-phase saves.
    constexpr bool ROLES = SPLIT && BLOCK == 256, PAIR = SPLIT && BLOCK == 128;
    static_assert(!CAP || (BLOCK == 64 && LPE == 1 && !SPLIT), "CAP is the one-wavefront-per-workgroup throughput form");
    // Per-wavefront LDS scratch, used for two things one after the other:
    //   substeps : the 10 pending-action slots this step consumes, slots[s][lane] as float4 (10 KiB) -- keeps 40 values
    //              out of the register file and lets substep k fetch its action with one ds_read_b128;
    //   post-step: the 64 x 26 frame tile that transposes lane-major registers into env-major bytes (6.5 KiB).
#define TACO_STAMP(k) do { if (P.stamps_on && P.stamps && blockIdx.x == 0 && threadIdx.x == 0) P.stamps[k] = __builtin_readcyclecounter(); } while (0)
    unsigned long long t_entry = 0;   // (stamp 0, written once the parameter block is there)
    if (Pin.stamps_on) t_entry = __builtin_readcyclecounter();
    // the step clock: kernel arguments on the eager path (which also leaves the NEXT values in the control block, so that a capture can start
    // at any time); the device-resident copy when this launch was captured into a HIP graph (its arguments are frozen)
    // (read unconditionally by scalar loads through the constant address space and selected without a branch: while a captured step kernel
    // runs nothing writes the clock -- the advance kernel comes after it -- and on the eager path the loaded values are simply not used)
    // (this wavefront's pair of wclk: envs [wave_env0, wave_env0 + 64 / LPE) -- the expression of wave_env0 below)
    const uint32_t clk_env0 = SPLIT ? (uint32_t)(blockIdx.x * 64 / LPE) : (uint32_t)((blockIdx.x * BLOCK + (threadIdx.x >> 6) * 64) / LPE);
    const __attribute__((address_space(4))) uint32_t *wc =
        (const __attribute__((address_space(4))) uint32_t *)(Pin.wclk + (size_t)(clk_env0 >> 4) * kWclkWordsPerGroup);
    const bool from_ctl = Pin.use_ctl != 0;
    uint32_t c_step = 0u, c_aux = 0u;
    if (!FUSED && clk_env0 < (uint32_t)Pin.n) { c_step = wc[0]; c_aux = wc[1]; }   // (wave-uniform; a wavefront past the last env returns below)
    struct { uint32_t step; int head, hh, phase; } clk{0u, 0, 0, 0};   // (filled behind the up-front loads: it reads the parameter block)
    constexpr int EPW = 64 / LPE;  // envs per wavefront
    constexpr int WAVE_LDS_WORDS = 10 * EPW * 4;
    static_assert(WAVE_LDS_WORDS >= EPW * 26, "tile must fit in the per-wave scratch");
    // SPLIT: only wavefront 0 needs the slot table; the obs / states role wavefronts need a frame tile each, the reward wavefront nothing
    constexpr int TILE_WORDS = EPW * 26;
    static_assert(TILE_WORDS % 4 == 0, "tiles stay 16-byte aligned");
    constexpr int LDS_WORDS = SPLIT ? WAVE_LDS_WORDS + 2 * TILE_WORDS : (BLOCK / 64) * WAVE_LDS_WORDS;
    __shared__ __attribute__((aligned(16))) float lds_all[LDS_WORDS];
    const int lane = threadIdx.x & 63;
    const int wv = FUSED ? FX.role : (int)(threadIdx.x >> 6);
    float *tile = SPLIT ? lds_all + (wv == 0 ? 0 : WAVE_LDS_WORDS + (wv == 3 ? TILE_WORDS : 0)) : lds_all + wv * WAVE_LDS_WORDS;
    float4 *slots = reinterpret_cast<float4 *>(tile);  // (wavefront 0 / non-SPLIT only)
    const int sub = lane & (LPE - 1);  // lane inside the env's quad (0 when LPE == 1)
    const int el = lane / LPE;         // env slot inside the wavefront
    const int i_raw = SPLIT ? (blockIdx.x * 64 + lane) / LPE : (blockIdx.x * BLOCK + threadIdx.x) / LPE;
    const bool in_range = i_raw < Pin.n;
    const bool active = in_range && sub == 0;  // the lane that stores for its env
    // post-phase roles (SPLIT: wave 0 = the step itself; wave 1 = battery server, then reward + done; wave 2 = obs stack; wave 3 = states stack)
    const bool roleS = ROLES ? wv == 3 : (!SPLIT || wv == 0), roleR = ROLES ? wv == 1 : (!SPLIT || wv == 0);
    const bool roleO = ROLES ? wv == 2 : (!SPLIT || wv == 0);
    const uint32_t wave_env0 = SPLIT ? (uint32_t)(blockIdx.x * 64 / LPE) : (uint32_t)((blockIdx.x * BLOCK + wv * 64) / LPE);  // first env of this wavefront
    if (wave_env0 >= (uint32_t)Pin.n) return;  // a wavefront past the last env has nothing to do (SPLIT: the whole workgroup shares wave_env0, so
                                               // the two barriers below stay uniform)
    // ------------------------------------------------------------------ the up-front loads (pre_physics_step FA:317-332 starts further down)
    // Every load the step needs is issued HERE, before the reset flag is known and before anything reads the parameter block: the flag, the
    // action, the 13 state chunks and the history rows are independent, so they share ONE memory round trip -- and everything their addresses
    // need (state / flag / action / history pointers, env count, flags, the history row) arrives in registers with the wavefront
    // (StepKernelArgs: kernel-argument preload), so that round trip no longer queues behind the one to the argument segment.
    const int i = in_range ? i_raw : Pin.n - 1;  // tail lanes shadow the last env and store nothing
    const uint32_t voff = (uint32_t)(i & 63) * 16u;  // this lane's byte offset inside every row of its tile
    constexpr uint32_t row_bytes = kRowBytes;         // one float4 row (chunk, history row or ring slot) of a tile
    // all of a wavefront's envs lie in one tile (EPW divides 64; tail lanes shadow env n - 1 of the same wavefront)
    const uint32_t tile_id = (uint32_t)__builtin_amdgcn_readfirstlane((int)(wave_env0 >> 6));
    const rsrc_t rS = make_rsrc(reinterpret_cast<const char *>(Pin.S) + (size_t)tile_id * (NUM_CHUNKS * kRowBytes), NUM_CHUNKS * kRowBytes);
    const rsrc_t rH = make_rsrc(reinterpret_cast<const char *>(Pin.hist) + (size_t)tile_id * (HIST_ROWS * kRowBytes), HIST_ROWS * kRowBytes);
    const uint32_t fl = Pin.flags;
    // rotor / aero parameters: per env only if something randomises them (launch-uniform switch, see kUniformParams)
    const bool uniform_params = (fl & kUniformParams) != 0;
    bool is_reset = false;
    float4 a_in = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    float4 c_pos{}, c_quat{}, c_lin{}, c_ang{}, c_pp{}, c_pi{}, c_om{}, c_misc{}, c_tau{}, c_op{}, c_a0{}, c_a1{};
    // the four most recent history rows (wave-uniform addresses): they hold the two oldest queued runs whenever the
    // queue is at most 4 runs deep, i.e. for delays up to ~40 ms; deeper queues fetch per lane further down
    float4 hwin[4];
    auto upfront_loads = [&]() {
        if constexpr (FUSED) {
            if (FX.reset_want > 0) is_reset = false;   // (comes from the reward role's mailbox, below: the loads are issued first)
            else is_reset = Pin.reset[i] != 0;
        } else {
            is_reset = Pin.reset[i] != 0;
        }
        if constexpr (!RESET_ONLY && !FUSED) a_in = reinterpret_cast<const float4 *>(Pin.act_in)[i];
        c_pos = CLD(C_POS); c_quat = CLD(C_QUAT); c_lin = CLD(C_LINVEL); c_ang = CLD(C_ANGVEL);
        c_pp = CLD(C_PID_PREV); c_pi = CLD(C_PID_INT); c_om = CLD(C_OMEGA); c_misc = CLD(C_MISC);
        if (!uniform_params) { c_tau = CLD(C_TAU); c_op = CLD(C_OPARA); c_a0 = CLD(C_AERO0); c_a1 = CLD(C_AERO1); }
        // the history row of this step: a register on the eager path; a captured launch (frozen arguments) waits for its pair of the device clock
        int hh_e = FUSED ? FX.hh : Pin.hh;
        if (!FUSED && from_ctl) { uint32_t a = c_aux; asm volatile("" : "+v"(a)); hh_e = __builtin_amdgcn_readfirstlane(wclk_hh(a)); }   // (wave-uniform)
#pragma unroll
        for (int k = 0; k < 4; ++k) {  // (rows this configuration's queue never reaches are not fetched: hw_rows)
            hwin[k] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
            if (k < Pin.hw_rows) hwin[k] = buf_ld4(rH, voff, (uint32_t)((hh_e - 1 - k) & (HIST_ROWS - 1)) * row_bytes);
        }
    };
    if constexpr (!FUSED) { if (!SPLIT || wv == 0) upfront_loads(); }   // (the rollout kernel, whose arguments are in registers anyway, keeps them where they were)
    if constexpr (ROLES && LPE == 4 && !FUSED && !RESET_ONLY) {   // (= DRAWS_SERVED, defined below)
        // The role wavefronts' reset draws (see rs_tab below): one Philox block per lane, from preloaded arguments too -- they run while the
        // argument segment's round trip is under way instead of behind it (barrier 1 waits for them).
        // (round 2 holds blocks 8 and 9 only: lane 2 of every quad draws the env's command block there, STREAM_CMD block 0 -- consumed by
        // reset_command_idx of the rotate / flip envs, at a reset and at progress 500)
        if (wv != 0) {
            uint32_t step_e = Pin.step;
            if (from_ctl) { uint32_t a = c_step; asm volatile("" : "+s"(a)); step_e = a; }   // (a captured launch waits for its pair of the device clock)
            const bool cmd_lane = wv == 3 && sub == 2 && Pin.task_mode != TACO_TASK_POS;
            if (((reset_blocks_wanted(fl) >> (4 * (wv - 1))) & 15u) || (wv == 3 && Pin.task_mode != TACO_TASK_POS)) {
                const float4 u = uniform_block(Pin, step_e, Pin.env_offset + i, cmd_lane ? STREAM_CMD : STREAM_RESET, cmd_lane ? 0u : (uint32_t)(4 * (wv - 1) + sub));
                ((__attribute__((address_space(3))) f32x4_t *)(lds_all + WAVE_LDS_WORDS))[(wv - 1) * 64 + lane] = f32x4_t{u.x, u.y, u.z, u.w};
            }
        }
    }
    // ------------------------------------------------------------------ the parameter block
    // Read from HERE on, behind the loads above, through a copy of the argument-segment pointer the compiler cannot see through: as ordinary
    // kernel arguments its fields are loaded in the kernel's entry block, and the first use of any of them (or the first reuse of a register one
    // of those scalar loads is still due to write) waits for ALL of them -- scalar loads return out of order, the wait is lgkmcnt(0) -- i.e. for
    // the round trip to the argument segment, ahead of the state loads.  One dword of every 64-byte line of the block first and ONE wait: the
    // compiler reads the fields in several dependent groups, each a round trip to a cold scalar cache unless its lines are already there.
    const StepParams *Pp = &Pin;
    if constexpr (!FUSED) {
        static_assert(sizeof(StepKernelArgs) > 384 && sizeof(StepKernelArgs) <= 448, "one touch per 64-byte line of the argument block");
        const __attribute__((address_space(4))) char *ka = (const __attribute__((address_space(4))) char *)__builtin_amdgcn_kernarg_segment_ptr();
        uint32_t t0, t1, t2, t3, t4, t5, t6;
        asm volatile("s_load_dword %0, %7, 0x0\n\ts_load_dword %1, %7, 0x40\n\ts_load_dword %2, %7, 0x80\n\ts_load_dword %3, %7, 0xc0\n\t"
                     "s_load_dword %4, %7, 0x100\n\ts_load_dword %5, %7, 0x140\n\ts_load_dword %6, %7, 0x180\n\ts_waitcnt lgkmcnt(0)"
                     : "=&s"(t0), "=&s"(t1), "=&s"(t2), "=&s"(t3), "=&s"(t4), "=&s"(t5), "=&s"(t6), "+s"(ka) : : "memory");
        Pp = (const StepParams *)(ka + kParamsOffset);
    }
    const StepParams &P = *Pp;
    if (Pin.stamps_on && P.stamps && blockIdx.x == 0 && threadIdx.x == 0) P.stamps[0] = t_entry;
    // (the fields the code between here and barrier 1 consults, in ONE group of scalar loads: read where they are used, each was its own
    // load-and-wait on the path to the barrier -- a dozen scalar-cache round trips of a wavefront with nothing else to issue.  What arrived
    // preloaded -- step, history row, env offset, task, flags -- is taken from Pin)
    const int b_head = P.head, b_phase = P.st_phase, b_substeps = P.substeps, b_mix1 = P.mix_n1, b_mix2 = P.mix_n2;
    if constexpr (!FUSED) asm volatile("" : : "s"(b_head), "s"(b_phase), "s"(b_substeps), "s"(b_mix1), "s"(b_mix2));
    clk.step = FUSED ? FX.step : (from_ctl ? c_step : Pin.step); clk.head = FUSED ? FX.head : (from_ctl ? wclk_head(c_aux) : b_head);
    clk.hh = FUSED ? FX.hh : (from_ctl ? wclk_hh(c_aux) : Pin.hh); clk.phase = from_ctl ? wclk_phase(c_aux) : b_phase;
    // the next values, left in this wavefront's pairs (and, by workgroup 0, in the control block) when the step wavefront is done: by then every
    // wavefront of the workgroup has its copy (the role wavefronts wait for theirs ahead of barrier 2)
    auto publish_clock = [&]() {
        if constexpr (!FUSED && !RESET_ONLY) {
            if ((threadIdx.x & 63) == 0) {
                const int nhead = clk.head + 10 >= TACO_RING_SLOTS ? clk.head + 10 - TACO_RING_SLOTS : clk.head + 10, nhh = (clk.hh + 1) & (HIST_ROWS - 1);
                const int nphase = P.st_period > 0 ? (clk.phase + 1 == P.st_period ? 0 : clk.phase + 1) : clk.phase;   // (a launch that does not write the ring leaves its phase alone)
                const uint32_t nstep = clk.step + 1u, naux = wclk_aux(nhead, nhh, nphase);
                uint32_t *w = P.wclk + (size_t)(clk_env0 >> 4) * kWclkWordsPerGroup;
#pragma unroll
                for (int gq = 0; gq < (64 / LPE) / 16; ++gq) { w[2 * gq] = nstep; w[2 * gq + 1] = naux; }
                if (blockIdx.x == 0 && threadIdx.x == 0) {
                    P.ctl[kCtlStep] = nstep; P.ctl[kCtlHead] = (uint32_t)nhead; P.ctl[kCtlHh] = (uint32_t)nhh; P.ctl[kCtlPhase] = (uint32_t)nphase;
                    if (from_ctl) atomicAdd(&P.ctl[kCtlDevSteps], 1u);
                }
            }
        }
    };
    // Frame stacks with len > 1: the wavefront's EPW rows are one contiguous range of EPW * len * 13 eight-byte words.  Word w of the new
    // stack is word w + 13 of the previous one (one frame further) inside the first len - 1 frames of its row (the HISTORY words), and
    // comes from this step's frame in the last frame of the row: the whole wavefront copies 512 contiguous bytes per instruction instead of
    // 64 rows 104 * len bytes apart.  In place (prev == buf) the reads run 13 words ahead of the writes; loads are issued in batches of eight
    // ahead of their stores.  shift_history moves the history words only -- they do not depend on this step, so the role wavefronts of the
    // SPLIT form do it while wavefront 0 runs the substeps and only the newest frame is left for the post-phase.
    // `out` (optional): the clamped copy VecTask.step hands to the caller (VT:331-332) receives clamp(word, +-clip) next to every word moved
    auto clamp2 = [](f32x2_t v, float c) { return f32x2_t{clampf(v.x, -c, c), clampf(v.y, -c, c)}; };
    auto clamp4 = [](f32x4_t v, float c) { return f32x4_t{clampf(v.x, -c, c), clampf(v.y, -c, c), clampf(v.z, -c, c), clampf(v.w, -c, c)}; };
    // The OUTPUT pointers (frame stacks, reward / time-out / done images, the gather block, the clamped copies) are consumed only by the post-phase:
    // the launch-per-step forms read them from the kernel-argument segment WHERE THEY ARE USED, behind an opaque copy of the segment pointer.
    // As ordinary kernel arguments the compiler loaded them at the kernel's entry and kept them -- an SGPR pair each -- alive across the step
    // wavefront's whole path: the OUT instantiations' loop carried 16 spill reloads per substep for it (finite-clip step() at 4 096 envs
    // 15.5 -> 14.2 us, round 4).  (FUSED: P is a per-step copy in registers inside the rollout kernel, not the kernel's argument.)
    auto lazy_arg = [&](size_t off) -> float * {
        const __attribute__((address_space(4))) char *ka = (const __attribute__((address_space(4))) char *)__builtin_amdgcn_kernarg_segment_ptr();
        asm volatile("" : "+s"(ka));
        return *(float *const __attribute__((address_space(4))) *)(ka + kParamsOffset + off);
    };
#define TACO_ARG(field) (FUSED ? (float *)P.field : lazy_arg(offsetof(StepParams, field)))
    // ... all of them at ONE point of the consuming wavefront's path: the role wavefronts at their entry (they idle at barrier 1 anyway; read at
    // their use sites the scalar-load latency sat on the tail of the launch: + 0.2 us at 4 096 envs), the one-wavefront forms behind the substep loop
    struct { float *obs, *states, *obs_prev, *states_prev, *rew, *done_f32, *gather, *obs_out, *states_out; uint8_t *timeout; } A{};
    auto load_out_args = [&]() {
        A.obs = TACO_ARG(obs); A.states = TACO_ARG(states); A.obs_prev = TACO_ARG(obs_prev); A.states_prev = TACO_ARG(states_prev);
        A.rew = TACO_ARG(rew); A.done_f32 = TACO_ARG(done_f32); A.gather = TACO_ARG(gather); A.timeout = (uint8_t *)TACO_ARG(timeout);
        A.obs_out = OUT ? TACO_ARG(obs_out) : nullptr; A.states_out = OUT ? TACO_ARG(states_out) : nullptr;
    };
    auto shift_history = [&](float *buf, const float *prev, uint32_t buf_bytes, int len, float *out, float clip) {
        const rsrc_t rB = make_rsrc(buf, buf_bytes), rP = make_rsrc(prev, buf_bytes);
        // (the clamped copies go out through plain global stores with an explicit range check: a third buffer descriptor here cost the
        // 4 096-env launch 68 more SGPR spills and 1 us, profiles/r02_d_ab_out_copy_paths.txt)
        char *const ob = reinterpret_cast<char *>(out);
        const uint32_t L13 = (uint32_t)len * 13u, hist = L13 - 13u, tot = (uint32_t)EPW * L13;
        const uint32_t magic = 0xFFFFFFFFu / L13 + 1u;  // row = (w * magic) >> 32 for w < 2^16
        const uint32_t base = wave_env0 * L13;          // first word of the wavefront's range
        for (uint32_t w0 = 0; w0 < tot; w0 += 8u * 64u) {
            f32x2_t v[8];
            bool mine[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const uint32_t w = w0 + (uint32_t)k * 64u + (uint32_t)lane;
                const uint32_t row = __umulhi(w, magic), wl = w - row * L13;
                mine[k] = w < tot && wl < hist;
                v[k] = f32x2_t{0.0f, 0.0f};
                if (mine[k]) v[k] = llvm_amdgcn_raw_buffer_load_v2f32(rP, (int)((base + w + 13u) * 8u), 0, 0);
            }
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const uint32_t w = w0 + (uint32_t)k * 64u + (uint32_t)lane;
                if (mine[k]) llvm_amdgcn_raw_buffer_store_v2f32(v[k], rB, (int)((base + w) * 8u), 0, 0);
                if (OUT && out && mine[k] && (base + w) * 8u < buf_bytes) *reinterpret_cast<f32x2_t *>(ob + (size_t)(base + w) * 8u) = clamp2(v[k], clip);
            }
        }
    };
    // (quad forms: 40 words per env instead of CARRY_WORDS -- the block doubles as the rotor-noise table, see rn_tab below)
    __shared__ __attribute__((aligned(16))) float hand[SPLIT ? (64 / LPE) * 40 : 4];
    // SPLIT: while wavefront 0 runs the substeps, wavefront 1 serves the battery model one substep AHEAD: the voltage of substep k
    // depends only on the rotor speeds left by substep k - 1, which are known ~2 000 clocks before the voltage is needed.
    // mb_om: rotor speeds (wave 0 -> 1), mb_v: voltage (1 -> 0), mb_bs: battery state E u1 t (0 -> 1 before, 1 -> 0 after), mb_seq: counters
    __shared__ float mb_om[SPLIT ? 64 : 4], mb_v[SPLIT ? 16 : 4], mb_bs[SPLIT ? 64 : 4];
    __shared__ int mb_seq[8];   // mailbox counters: 0 rotor speeds, 1 voltage, 2 rotor-noise table, 3 attitudes for the euler server, 4 its result
    // Rotor noise (CTRL/thrust_dynamics.py:68-78: every substep multiplies each rotor speed by a fresh uniform factor) SERVED by the obs role
    // wavefront in the four-role quad forms: the 10 x EPW Philox blocks of a step depend on counters only, so that wavefront draws them all
    // into this table right after barrier 1 (~3 blocks per lane) while wavefront 0 is still in its pre-phase, and a substep costs wavefront 0
    // one LDS read and the multiply instead of a Philox block evaluated redundantly in the four lanes of every env (~1 000 instructions of a
    // config-5 step's critical path).  Same draws, same expression for the factor: same bits.
    // Euler angles + unwrap (FA:339-347: consumed only by the flip command, once per step) SERVED by the states role wavefront in the same
    // forms: wavefront 0 leaves the attitude of every substep in eq_tab (one LDS write + the counter) and the server runs the eleven
    // atan2 / asin / unwrap evaluations of the step behind it -- ~150 instructions per substep off the critical path of every wavefront
    // that holds flip envs (flip at 4 096 envs: 17.9 -> us; the tail of a mixed-task rollout).  Same quad-layout code, same bits.
    constexpr bool EULER_TAB = SPLIT && !(WIDE && LPE == 4);   // (WIDE quad = the headline form: launches without flip envs and rotor noise only, see wide_form)
    // (eq_tab lives in the two role tiles, 2 x 16 x 26 words, which are idle until barrier 2 -- by then the server has read every attitude)
    // (one lane per env: the attitude is a float4 per lane and substep, the unwrap memory six words per lane)
    static_assert(!EULER_TAB || 2 * TILE_WORDS >= 11 * 64 * (LPE == 4 ? 1 : 4), "eq_tab must fit in the role wavefronts' frame tiles");
    float *const eq_tab = lds_all + WAVE_LDS_WORDS;
    __shared__ float eu_io[EULER_TAB ? (LPE == 4 ? 2 : 6) * 64 : 4];
    // Reset draws SERVED in the four-role quad forms: a resetting env consumes up to ten Philox blocks of STREAM_RESET, shared out over its quad as
    // three per lane (reset_env) -- ~200 instructions on the path of the wavefront that decides the launch time (some wavefront of nearly every
    // 4 096-env launch holds a resetting env).  The blocks depend on counters only, and the three role wavefronts idle at barrier 1 while wavefront 0
    // waits for its loads: each draws ONE round (block 4 (wv - 1) + sub of its lane's env) into rs_tab ahead of the barrier, every step, whether or
    // not an env resets; wavefront 0 passes the barrier before its reset and reads its lane's three entries.  Same counters, same uniforms.
    // (rs_tab lives in the role wavefronts' frame tiles like eq_tab: read by the reset, before wavefront 0's first substep writes an attitude there)
    constexpr bool DRAWS_SERVED = ROLES && LPE == 4 && !FUSED && !RESET_ONLY;
    static_assert(!DRAWS_SERVED || 2 * TILE_WORDS >= 3 * 64 * 4, "rs_tab must fit in the role wavefronts' frame tiles");
    __attribute__((address_space(3))) f32x4_t *const rs_tab = (__attribute__((address_space(3))) f32x4_t *)(lds_all + WAVE_LDS_WORDS);
    // Linear chain SERVED in the four-role quad forms.  Inside a step nothing feeds back from the linear state into the angular chain: rate PID,
    // allocator, rotors, torque, body rates and attitude never read position or linear velocity; v and p only consume the attitude (before each of
    // the two sub-iterations) and the rotors' thrust.  So wavefront 0 keeps the angular chain -- the part that is serial from substep to substep --
    // and leaves, per substep, the two attitudes and its lanes' pair thrust sums in lin_ring (three LDS writes; the counter, mb_seq[5], is
    // published at the top of the NEXT substep, when the integrator's rare-form test of this one is settled); the obs role wavefront (idle until the
    // post-phase) runs body-frame velocity, drag, force rotation and the v / p updates behind it (linear_iteration_quad: the same expressions),
    // hands p and v to the Carry and stores their words of the state -- ~80 of the ~315 instructions of every substep leave the critical path
    // (wavefront 0 issues one instruction per ~5 clocks whatever its type), at the price of the server's last substep (~500 clocks) before barrier 2.
    // LIN (template parameter): the host picks the build per launch -- served up to 8 192 envs (at most two wavefronts per SIMD: 14.9 -> 14.5 us at
    // 8 192), not beyond (at 12 288 the server's instructions cost the step wavefronts next to it more than they save: 15.6 -> 16.2 us), always in
    // the rollout kernel's own CU.  (A run-time switch inside one kernel -- both builds of every PLAIN form compiled in -- cost the 4 096-env launch
    // 0.3 us: profiles/r04_n_ab_linear_chain.txt.)
    // PAIR + LIN (round 5): the served pair's second wavefront runs the linear chain -- the one server a launch WITHOUT flip envs and rotor noise has
    // work for (pos / rotate at 12 289 ... 16 384 envs: exactly two wavefronts per SIMD, the regime in which the server paid in the four-role form).
    // Wavefront 0 runs the post-phase itself there, so it takes p and v back from the server (mb_seq[6], behind its own state stores).
    constexpr bool LIN_SERVED = (ROLES || PAIR) && LPE == 4 && LIN;
    __shared__ float lin_ring[LIN_SERVED ? (10 * 3 + 1) * 64 : 4];   // [substep][0: q before the substep, 1: q after its first sub-iteration, 2: rf + swp(rf)][lane]
                                                                     // ([.][0] is written at the end of the substep BEFORE: the server starts on it one counter earlier)
    __shared__ float lin_init[LIN_SERVED ? 5 * 64 : 4];        // [p, v, drag coefficient, kt, keep mask][lane]: posted with counter value 1
    // (the server runs exactly two sub-iterations; in the pair form it is the ONLY serving wavefront: with flip envs it has the euler angles to serve
    // instead -- the host never picks the LIN build of the pair for such a launch, and the PLAIN forms, which assume it, are then not taken)
    const bool lin_served = LIN_SERVED && b_substeps == 2 &&
                            !(PAIR && (Pin.task_mode == TACO_TASK_FLIP || Pin.task_mode == TACO_TASK_MIX || (Pin.flags & TACO_F_TRACK_RPY) != 0));
    constexpr bool NOISE_TAB = SPLIT && !(WIDE && LPE == 4);
    constexpr bool NOISE_INLINE = !SPLIT;   // the one-wavefront instantiations have nobody to serve a table: their PLAIN forms draw the rotor noise themselves
    // (rn_tab lives in `hand`: the table is read by the substeps, the Carry is written after the last one and read after barrier 2; the
    // server writes the next table only after barrier 1 of the next step)
    static_assert(!NOISE_TAB || (64 / LPE) * 40 >= 10 * (64 / LPE) * 4, "rn_tab must fit in the Carry block");
    float *const rn_tab = hand;
    const bool noise_served = NOISE_TAB && (fl & TACO_F_ROTOR_NOISE) != 0;
    // The arrays are named directly at every use (macros, not lambdas or pointer parameters) so that the accesses stay LDS instructions;
    // through a generic pointer they become flat loads.  MB_WAIT is bounded, so a protocol bug can never hang the GPU; a wait that gives up
    // sets kStatusMailboxTimeout in the workspace's sticky status word (taco_check reports TACO_ERR_STATE) and poisons the voltage with NaN.
    // The LDS executes one wavefront's operations in order (data before counter on the writer's side, counter before data on the
    // reader's), so the fences only have to stop the compiler: wavefront scope, no s_waitcnt on the critical path.
#define MB_SEQ(idx) __hip_atomic_load(&mb_seq[idx], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)
#define MB_WAIT(idx, want) do { int spins_ = 0; while (MB_SEQ(idx) < (want) && ++spins_ < (1 << 20)) __builtin_amdgcn_s_sleep(1); \
                                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront"); if (spins_ >= (1 << 20)) mb_timeout = true; } while (0)
    bool mb_timeout = false;  // a bounded wait gave up: reported through the sticky status word and a NaN voltage, see below
#define MB_POST(idx, value) do { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); \
                                 __hip_atomic_store(&mb_seq[idx], (value), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); } while (0)
    // (only while every wavefront of the launch has a SIMD to itself -- 1 024 on the MI355X -- or the server would take issue slots
    // from another workgroup's wavefront 0: 19.0 vs 18.4 us at 8 192 envs)
    const bool bat_served = ROLES && LPE == 4 && (fl & TACO_F_BATTERY_CONSUMPTION) != 0 && (FUSED || TACO_AB_SERVE_ALWAYS || (uint32_t)((Pin.n + EPW - 1) / EPW) * 4u <= 1024u);   // (FUSED: one workgroup per CU; the grid of a SPLIT launch
                                                                                                                     // from the env count: gridDim is a load from the implicit arguments)
    const int euler_role = bat_served ? 3 : 1;   // which role wavefront serves the euler angles: the reward wavefront while it is idle (no battery to serve),
                                                 // otherwise the states wavefront (which may first have a stack history to move)
    Carry K;
    // the frame's operands in the quad layout (lane j of an env's quad: component j; lane 3: battery voltage / cmd0 / cmd1 / -), see the post-phase.
    // QUAD_POST: the quad forms in which ONE wavefront runs the whole post-phase (one-wavefront quad form, served pair).  In the four-role forms the
    // post-phase is three wavefronts wide and its end does not follow their instruction counts (same-box A/B: neutral to + 0.1 us there,
    // profiles/r04_s_ab_post_phase.txt): they keep the scalar post-phase.
    constexpr bool QUAD_POST = LPE == 4 && !FUSED && !ROLES;
    struct { float p, v, w, a, tp; } QF{};
    bool qf = false;
    const int gid = Pin.env_offset + i;
    const rsrc_t rR = make_rsrc(reinterpret_cast<const char *>(P.ring) + (size_t)tile_id * (TACO_RING_SLOTS * kRowBytes), TACO_RING_SLOTS * kRowBytes);
    const int grp = (Pin.task_mode != TACO_TASK_MIX) ? Pin.task_mode
                    : (gid < b_mix1 ? TACO_TASK_POS : (gid < b_mix2 ? TACO_TASK_ROTATE : TACO_TASK_FLIP));
    const bool mix = Pin.task_mode == TACO_TASK_MIX;
    // copter_rpy_continuous is consumed only by the flip command (FA:831, :930); other envs keep their reset-time value
    const bool track_rpy = grp == TACO_TASK_FLIP || (fl & TACO_F_TRACK_RPY) != 0;  // (record_flag: every env, as the reference does)
    // wave-uniform form of the same predicate: the euler/unwrap block runs for a whole wavefront or not at all (lanes of
    // other tasks then compute values nobody reads), which keeps the substep loop free of divergent control flow
    const bool wave_tracks_rpy = __builtin_amdgcn_ballot_w64(track_rpy) != 0;
    const bool euler_served = EULER_TAB && wave_tracks_rpy;   // (the same for every wavefront of the workgroup: they hold the same envs)

    // Observation noise FA:402-410: 12 normals (Box-Muller on uniforms 4..15 of STREAM_OBS) and the small random rotation.  None of it
    // depends on the state, so the obs role wavefront of the SPLIT form draws it while the substeps run.
    float nrm[12];
    Q4 nq{0.0f, 0.0f, 0.0f, 1.0f};
    bool noise_ready = false;
    auto gen_obs_noise = [&]() {
#pragma unroll
        for (int pr = 0; pr < 3; ++pr) {  // uniforms 4..15 of STREAM_OBS = blocks 1..3, two Box-Muller pairs per block
            U4 r = philox(P.seed_lo, P.seed_hi, (uint32_t)gid, clk.step, STREAM_OBS, (uint32_t)(1 + pr));
            float ua = 1.0f - uniform(r.x), ub = uniform(r.y);
            float rad = __builtin_sqrtf(-2.0f * log(ua));
            float sn, cs;
            sincos(kTwoPi * ub, sn, cs);
            nrm[4 * pr] = rad * cs; nrm[4 * pr + 1] = rad * sn;
            ua = 1.0f - uniform(r.z); ub = uniform(r.w);
            rad = __builtin_sqrtf(-2.0f * log(ua));
            sincos(kTwoPi * ub, sn, cs);
            nrm[4 * pr + 2] = rad * cs; nrm[4 * pr + 3] = rad * sn;
        }
        U4 r0 = philox(P.seed_lo, P.seed_hi, (uint32_t)gid, clk.step, STREAM_OBS, 0u);
        nq = quat_from_euler(P.nq_sc * uniform(r0.x) + P.nq_lo, P.nq_sc * uniform(r0.y) + P.nq_lo, P.nq_sc * uniform(r0.z) + P.nq_lo);
    };
    if (!SPLIT || wv == 0) {
    // ------------------------------------------------------------------ pre_physics_step FA:317-332
    // (the up-front loads were issued at the top of the kernel, see "up-front loads"; what is left here are the parameter VALUES of a launch
    // without per-env randomisation)
    if constexpr (FUSED) upfront_loads();
    if (uniform_params) {
        const float t0 = P.tau_fixed * 1.0f;  // what reset_env and init_state_kernel hold for every env
        c_tau = make_float4(t0, t0, t0, t0); c_op = make_float4(0.0f, 12.9466f, 0.1872f, -5.1220f);
        c_a0 = make_float4(0.5906f, 1.13e-05f, 0.05f, -0.386f); c_a1 = make_float4(-0.53f, 0.009f, 0.0f, 0.0f);
    }
    if constexpr (FUSED) {
        if (FX.reset_want > 0) {   // the previous step's done flags: posted by the reward role, ~a load round trip after this wavefront got here
            int spins_ = 0;
            while (__hip_atomic_load(FX.reset_seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) < FX.reset_want && ++spins_ < (1 << 22)) __builtin_amdgcn_s_sleep(1);
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            if (spins_ >= (1 << 22)) mb_timeout = true;
            is_reset = FX.reset_lds[el] != 0;
        }
    }
    // reset_idx (FA:475-517): the lanes that reset get their fresh state in the registers the up-front loads filled (see reset_env)
    const bool wave_has_reset = __builtin_amdgcn_ballot_w64(is_reset) != 0;
    if constexpr (DRAWS_SERVED) __syncthreads();   // barrier 1 of 2 (here in these forms: the role wavefronts' reset draws are in rs_tab, see below)
    if (P.stamps_on) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); TACO_STAMP(1); }  // all up-front loads have landed
    if (wave_has_reset) {
        // Shared reset draws (LPE == 1).  A resetting env draws from up to ten Philox blocks, and in these forms the lane that holds it evaluated
        // them one after the other with the wavefront's other 63 lanes switched off: ~100 VALU instructions per block in every wavefront that
        // holds a resetting env -- half of them at the benchmark's 1.1 % resets per step.  With at most four resetting lanes in the wavefront
        // (else every lane draws for itself, as before) the sixteen lanes [16 j, 16 j + 16) evaluate blocks 0..15 of the j-th resetting lane's
        // env in ONE Philox pass and leave the uniforms in the wavefront's scratch (the slot table is not built yet); the resetting lane reads
        // the blocks it wants.  Same counters, same bits.  Steady state at 262 144 envs: 28.63 M -> 28.01 M VALU instructions per launch,
        // 42.6 -> 42.4 us; one wavefront per SIMD (32 768 envs) 20.5 -> 20.2 us, flip 65 536 29.4 -> 28.9 (profiles/r05_x_*).  The quad forms
        // keep their own share-out (three blocks per lane of the quad): three independent Philox chains interleave at the issue rate, one chain
        // runs at its dependent latency, and with the hop through LDS the shared pass came out 0.15-0.3 us SLOWER there (same profile).
        const __attribute__((address_space(3))) f32x4_t *coop_tab = nullptr;
        int coop_row = 0;
        bool coop = false;
        if constexpr (LPE == 1 && !FUSED) {
            const uint64_t rm = __builtin_amdgcn_ballot_w64(is_reset);
            if (__builtin_popcountll(rm) <= 4) {   // (wave-uniform)
                uint64_t m = rm;
                int src[4];
#pragma unroll
                for (int k = 0; k < 4; ++k) {   // the k-th resetting lane (the last one again when there are fewer)
                    src[k] = (int)__builtin_ctzll(m);
                    const uint64_t m2 = m & (m - 1);
                    if (m2) m = m2;
                }
                const int i0 = __builtin_amdgcn_readlane(i, src[0]), i1 = __builtin_amdgcn_readlane(i, src[1]);
                const int i2 = __builtin_amdgcn_readlane(i, src[2]), i3 = __builtin_amdgcn_readlane(i, src[3]);
                const int jg = lane >> 4;
                const int i_src = jg == 0 ? i0 : (jg == 1 ? i1 : (jg == 2 ? i2 : i3));
                const float4 u = uniform_block(P, clk.step, Pin.env_offset + i_src, STREAM_RESET, (uint32_t)(lane & 15));
                __attribute__((address_space(3))) f32x4_t *tab = (__attribute__((address_space(3))) f32x4_t *)tile;
                tab[lane] = f32x4_t{u.x, u.y, u.z, u.w};
                __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
                coop_tab = tab;
                coop = true;
                coop_row = (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(rm >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)rm, 0u));   // resetting lanes below this one
            }
        }
        if (is_reset) {
            auto do_reset = [&](auto fl_c) {
                reset_env<LPE, decltype(fl_c)::value>(P, clk.step, rS, voff, row_bytes, gid, grp, mix, active, sub, c_pos, c_quat, c_lin, c_ang, c_pp, c_pi, c_om, c_misc, c_tau, c_op,
                                                      c_a0, c_a1, DRAWS_SERVED ? rs_tab : coop_tab, coop, coop_row);
            };
            if constexpr (!RESET_ONLY) {   // (see reset_env's FL)
                if ((fl & kResetFlagMask) == kResetFlagsBaseline) do_reset(std::integral_constant<uint32_t, kResetFlagsBaseline>{});
                else if ((fl & kResetFlagMask) == kResetFlagsAll) do_reset(std::integral_constant<uint32_t, kResetFlagsAll>{});
                else do_reset(std::integral_constant<uint32_t, ~0u>{});
            } else {
                do_reset(std::integral_constant<uint32_t, ~0u>{});
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");  // (the target pose and the unwrapped angles it stored are loaded further down)
    }
    TACO_STAMP(9);  // reset_idx done (no wait for its stores)
    V3 p{c_pos.x, c_pos.y, c_pos.z};
    Q4 q{c_quat.x, c_quat.y, c_quat.z, c_quat.w};
    V3 v{c_lin.x, c_lin.y, c_lin.z};
    V3 w{c_ang.x, c_ang.y, c_ang.z};
    float rpy_old[3] = {0.0f, 0.0f, 0.0f}, rpy_cont[3] = {0.0f, 0.0f, 0.0f};
    if (wave_tracks_rpy) {
        const float4 c_ro = CLD(C_RPY_OLD), c_rc = CLD(C_RPY_CONT);
        rpy_old[0] = c_ro.x; rpy_old[1] = c_ro.y; rpy_old[2] = c_ro.z;
        rpy_cont[0] = c_rc.x; rpy_cont[1] = c_rc.y; rpy_cont[2] = c_rc.z;
    }
    float pid_prev[3] = {c_pp.x, c_pp.y, c_pp.z}, pid_int[3] = {c_pi.x, c_pi.y, c_pi.z};
    float omega[4] = {c_om.x, c_om.y, c_om.z, c_om.w};
    float act[4];
    const float tau[4] = {c_tau.x, c_tau.y, c_tau.z, c_tau.w};
    const float opara[5] = {c_op.x, c_op.y, c_op.z, c_op.w, c_a0.x};
    float bat_E = c_pp.w, bat_u1 = c_pi.w, bat_t = c_misc.x, bat_V = c_ang.w;
    if constexpr (SPLIT) {
        if constexpr (!DRAWS_SERVED) __syncthreads();  // barrier 1 of 2: wavefront 1 has zeroed the mailbox counters (the helpers have been waiting here since
                                                       // their launch and wavefront 0 arrives a load round trip later, so it does not wait)
        if (bat_served) {  // the battery server can start on the first substep's voltage while wavefront 0 finishes its pre-phase
            mb_om[lane] = pick4(sub, omega[0], omega[1], omega[2], omega[3]);
            mb_bs[lane] = pick4(sub, bat_E, bat_u1, bat_t, 0.0f);
            MB_POST(0, 1);
        }
    }
    float cmd0 = c_misc.y, cmd1 = c_misc.z, flip_radian = c_misc.w;
    const float cf = c_a0.y, ct = c_a0.z, dx = c_a0.w, dy = c_a1.x, kt = c_a1.y;
    int progress = pw_progress((uint32_t)as_i(c_pos.w));
    // ---- pending actions (actions_remained_buffer [4,100] + actions_remained_length, FA:189-193, :323-332, :366, :378-380).
    // Every step appends its action T times at logical slots [L, L+T) and consumes slots 0..9, so the dense buffer is a
    // queue of RUNS.  While no write of an env has reached the tail slots [90,100) since its reset (L + T <= 90 always; with
    // delay_time = 20 that is the whole episode) the reference's buffer is exactly
    //       [ Z zeros of the reset | run_0 (rem0 slots left) | run_1 | ... | run_{m-1} | zeros ],
    // and it is kept as such: 2-bit length codes + (m, rem0) in C_QUEUE, Z and L in the delay word, the run VALUES in the
    // action history hist[16][npad] (the action of the step j steps ago sits in row (hh - j) & 15: one coalesced 16-byte
    // store per step instead of T slot stores, and the two oldest runs -- all that ten slots can span -- are read instead
    // of ten slots).  The first write that touches the tail switches the env to DENSE: the queue is expanded once into the
    // literal slot ring (logical slot s at physical (head + s) % 100, head advancing by 10 per step on the host), and from
    // then on the reference's buffer semantics are followed to the letter (truncated mask write, the shift that leaves
    // [90,100) stale, stale reads: SURVEY section 7 "overflow regime").  Exact in every regime.
    const uint32_t dl_word = (uint32_t)as_i(c_lin.w);  // "queue words" at the top of this file
    int dlen = dw_L(dl_word);    // L
    int zlead = dw_Z(dl_word);   // Z
    bool dense = dw_dense(dl_word);
    int q_lens = qw_lens(dl_word, (uint32_t)as_i(c_pos.w));
    int q_m = dw_m(dl_word);
    int q_rem0 = dw_rem0(dl_word);
    const bool at_time = !is_reset && progress == 500;  // reset_command_condition FA:595-598 (a reset env has progress 0 here;
                                                        // its own progress==500 case is overwritten by the reset branch below)
    if (is_reset || at_time) {  // reset_command_idx: FA:758-759, :814-821, :886-917, :1058-1112
        if (grp == TACO_TASK_POS) {
            cmd0 = 0.0f; cmd1 = 0.0f;
        } else {
            float cu_x, cu_y;   // uniforms 0, 1 of STREAM_CMD block 0 (DRAWS_SERVED: drawn by the states role wavefront, lane 2 of the env's quad)
            if constexpr (DRAWS_SERVED) { const f32x4_t d = rs_tab[2 * 64 + el * 4 + 2]; cu_x = d.x; cu_y = d.y; }
            else { const float4 d = uniform_block(P, clk.step, gid, STREAM_CMD, 0u); cu_x = d.x; cu_y = d.y; }
            if (grp == TACO_TASK_ROTATE) {
                cmd0 = 1.0f;
                cmd1 = (fl & TACO_F_RANDOM_COMMAND) ? 12.0f * cu_x + -6.0f : 1.0f;
            } else {
                if (at_time) {
                    float uu = cu_y, t = 0.0f;
                    if (uu < 1.0f / 8) t = -3.0f;
                    if (uu >= 1.0f / 8 && uu < 2.0f / 8) t = -2.0f;
                    if (uu >= 2.0f / 8 && uu < 3.0f / 8) t = -1.0f;
                    if (uu >= 5.0f / 8 && uu < 6.0f / 8) t = 1.0f;
                    if (uu >= 6.0f / 8 && uu < 7.0f / 8) t = 2.0f;
                    if (uu >= 7.0f / 8) t = 3.0f;
                    flip_radian = flip_radian + kTwoPi * t;
                }
                if (is_reset) flip_radian = (w.x > 5.0f) ? kTwoPi : -kTwoPi;
                cmd0 = -1.0f;
            }
        }
    }
    if constexpr (RESET_ONLY) {
        // reset_done (VT:363-375) = reset_idx on the flagged envs, outside a step: the fresh state (reset_env above) and the command go to
        // the env's chunks, reset_buf / progress_buf are cleared (FA:510-511); reset_command_condition's progress == 500 envs get their
        // command re-drawn as well (FA:500-503) -- and again, with the same draw, in the next step, whose pre_physics_step calls reset_idx too
        if (active && (is_reset || at_time)) {
            if (is_reset) {
                CST(C_POS, c_pos); CST(C_QUAT, c_quat); CST(C_LINVEL, c_lin); CST(C_ANGVEL, c_ang);
                CST(C_PID_PREV, c_pp); CST(C_PID_INT, c_pi); CST(C_OMEGA, c_om);
                P.reset[i] = 0;
            }
            CST(C_MISC, make_float4(bat_t, cmd0, cmd1, flip_radian));
        }
        return;
    }
    int T = 10;
    auto draw_T = [&]() {
        if (fl & TACO_F_RANDOM_DEPLOY_TIME) {
            U4 r = philox(P.seed_lo, P.seed_hi, (uint32_t)gid, clk.step, STREAM_DEPLOY, 0u);
            T = 10 - rounded_normal(uniform(r.x), 1);
        }
    };
    // FUSED: this step's action is being computed by the actor wavefronts right now.  The ten slots this step consumes hold it only if an
    // env has fewer than ten pending slots, or keeps the literal slot ring (DENSE: the ring is written ahead of the substeps); then -- wave-
    // uniformly, rarely: delay_time < 10 ms or the overflow regime -- the step waits for the actor here.  Otherwise the action is picked up
    // after the substeps (`have_act` false: the zeros standing in for it below are never selected into a slot).
    bool have_act = !FUSED;
    auto fetch_action = [&]() {   // (FUSED) wait for the actor's mailbox, take this env's clipped action
        int spins_ = 0;
        while (__hip_atomic_load(FX.act_seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) < FX.act_want && ++spins_ < (1 << 22)) __builtin_amdgcn_s_sleep(1);
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        if (spins_ >= (1 << 22)) mb_timeout = true;
        a_in = make_float4(FX.act_lds[el * 4 + 0], FX.act_lds[el * 4 + 1], FX.act_lds[el * 4 + 2], FX.act_lds[el * 4 + 3]);
        have_act = true;
    };
    // ... or, with 0..9 pending slots, only from slot `dlen` on: the substeps before it run first and the action is patched into the slot table
    // right before its first use (`patch_at`; the random walk of the deploy time takes envs there in the documented config 5, and a
    // workgroup that waited for the actor up front ran actor and step one after the other: the tail of the whole rollout kernel).
    int patch_at = 10;   // the substep before which the action must be in the slot table; 10 = the substeps never read it
    const int dlen0 = dlen;
    if constexpr (FUSED) {
        draw_T();
        const bool now = dense || dlen + T > TACO_RING_SLOTS - 10;   // (the literal slot ring is written ahead of the substeps)
        if (__builtin_amdgcn_ballot_w64(now) != 0) {
            fetch_action();
        } else {
#pragma unroll
            for (int k = 9; k >= 0; --k)
                if (__builtin_amdgcn_ballot_w64(dlen <= k) != 0) patch_at = k;   // (ends at min(dlen) over the wavefront: the first substep that reads it;
                                                                                   // 0 = no pending slot at all: only the pre-phase runs ahead of the actor)
        }
    }
    act[0] = clampf(a_in.x, -P.clip_act, P.clip_act);  // VT:304; actions_old <- actions <- a (FA:321-322) is implicit in hist
    act[1] = clampf(a_in.y, -P.clip_act, P.clip_act);
    act[2] = clampf(a_in.z, -P.clip_act, P.clip_act);
    act[3] = clampf(a_in.w, -P.clip_act, P.clip_act);
    float4 act4 = make_float4(act[0], act[1], act[2], act[3]);
    const float4 zero4 = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    if (have_act && active) buf_st4(rH, act4, voff, (uint32_t)(clk.hh & (HIST_ROWS - 1)) * row_bytes);  // this step's action, row hh
    if constexpr (!FUSED) draw_T();
    // value of queued run j (0 = oldest) BEFORE this step's push: the action of (m - j) steps ago
    auto run_value = [&](int j) -> float4 {
        const uint32_t row = (uint32_t)((clk.hh - (q_m - j)) & (HIST_ROWS - 1));
        return buf_ld4(rH, row * row_bytes + voff, 0u);
    };
    if (!dense && dlen + T > TACO_RING_SLOTS - 10) {
        // first write that reaches the tail [90,100): expand the queue into the literal slot ring, once
        if (active) {
            int pos = 0;
            for (; pos < zlead; ++pos) {
                int ph = clk.head + pos; ph = ph >= TACO_RING_SLOTS ? ph - TACO_RING_SLOTS : ph;
                buf_st4(rR, zero4, (uint32_t)ph * row_bytes + voff, 0u);
            }
            for (int j = 0; j < q_m; ++j) {
                const float4 val = run_value(j);
                const int len = (j == 0) ? q_rem0 : run_len(q_lens, j);
                for (int t = 0; t < len; ++t, ++pos) {
                    int ph = clk.head + pos; ph = ph >= TACO_RING_SLOTS ? ph - TACO_RING_SLOTS : ph;
                    buf_st4(rR, val, (uint32_t)ph * row_bytes + voff, 0u);
                }
            }
            for (; pos < TACO_RING_SLOTS; ++pos) {
                int ph = clk.head + pos; ph = ph >= TACO_RING_SLOTS ? ph - TACO_RING_SLOTS : ph;
                buf_st4(rR, zero4, (uint32_t)ph * row_bytes + voff, 0u);
            }
        }
        dense = true;
        zlead = 0; q_m = 0; q_rem0 = 0; q_lens = 0;
    }
    // ---- the ten slots this step consumes -> per-wave LDS table (substep k reads entry min(L-1, k), FA:366)
    const bool wave_dense = __builtin_amdgcn_ballot_w64(dense) != 0;  // wave-uniform: normally no lane is DENSE
    float in_mag;  // sum of the magnitudes of everything the PID / allocator of this step consume (see `fin` below)
    if constexpr (LPE == 1) {
        // queue lanes: slots 0..9 of [Z zeros | run_0 | run_1 | ...] after the push span at most two runs
        float4 v0 = act4, v1 = act4;  // the pushed run itself is run_0 (empty queue) or run_1 (one run queued)
        v0 = sel4(q_m == 1, hwin[0], v0); v0 = sel4(q_m == 2, hwin[1], v0); v0 = sel4(q_m == 3, hwin[2], v0); v0 = sel4(q_m >= 4, hwin[3], v0);
        v1 = sel4(q_m == 2, hwin[0], v1); v1 = sel4(q_m == 3, hwin[1], v1); v1 = sel4(q_m >= 4, hwin[2], v1);
        if (__builtin_amdgcn_ballot_w64(!dense && q_m > P.hw_rows)) {  // queue deeper than the rows fetched up front: the two oldest runs per lane
            const bool deep = !dense && q_m > P.hw_rows;
            const float4 g0 = run_value(0), g1 = run_value(1);
            v0 = sel4(deep, g0, v0); v1 = sel4(deep, g1, v1);
        }
        const int n0 = (q_m >= 1) ? q_rem0 : T;  // slots of run_0 ahead of run_1
        in_mag = ((absf(v0.x) + absf(v0.y)) + (absf(v0.z) + absf(v0.w))) + ((absf(v1.x) + absf(v1.y)) + (absf(v1.z) + absf(v1.w)));
#pragma unroll
        for (int s = 0; s < 10; ++s) slots[s * EPW + el] = sel4(s < zlead, zero4, sel4((s - zlead) < n0, v0, v1));
        if (__builtin_expect(wave_dense, 0)) {  // DENSE lanes overwrite their entries from the ring (kept out of the common path)
#pragma unroll 1
            for (int s = 0; s < 10; ++s) {
                int ph = clk.head + s; ph = ph >= TACO_RING_SLOTS ? ph - TACO_RING_SLOTS : ph;
                const float4 r = buf_ld4(rR, voff, (uint32_t)ph * row_bytes);  // physical row is wave-uniform: coalesced
                const bool fresh = (s >= dlen) && (s < dlen + T);             // covered by this step's write: from registers
                if (dense) slots[s * EPW + el] = sel4(fresh, act4, r);
            }
        }
    } else {
        // the same table in the quad layout: lane j of an env only ever reads component j of a slot, so it selects and stores that one
        float *slotsw = reinterpret_cast<float *>(slots);
        const float ac = pick4(sub, act[0], act[1], act[2], act[3]);
        const float h0 = pick4(sub, hwin[0].x, hwin[0].y, hwin[0].z, hwin[0].w), h1 = pick4(sub, hwin[1].x, hwin[1].y, hwin[1].z, hwin[1].w);
        const float h2 = pick4(sub, hwin[2].x, hwin[2].y, hwin[2].z, hwin[2].w), h3 = pick4(sub, hwin[3].x, hwin[3].y, hwin[3].z, hwin[3].w);
        float v0 = ac, v1 = ac;
        v0 = (q_m == 1) ? h0 : v0; v0 = (q_m == 2) ? h1 : v0; v0 = (q_m == 3) ? h2 : v0; v0 = (q_m >= 4) ? h3 : v0;
        v1 = (q_m == 2) ? h0 : v1; v1 = (q_m == 3) ? h1 : v1; v1 = (q_m >= 4) ? h2 : v1;
        if (__builtin_amdgcn_ballot_w64(!dense && q_m > P.hw_rows)) {  // queue deeper than the rows fetched up front: the two oldest runs per lane
            const bool deep = !dense && q_m > P.hw_rows;
            const float4 g0 = run_value(0), g1 = run_value(1);
            v0 = deep ? pick4(sub, g0.x, g0.y, g0.z, g0.w) : v0;
            v1 = deep ? pick4(sub, g1.x, g1.y, g1.z, g1.w) : v1;
        }
        const int n0 = (q_m >= 1) ? q_rem0 : T;  // slots of run_0 ahead of run_1
        in_mag = absf(v0) + absf(v1);  // this lane's component; the ballot below covers the quad
#pragma unroll
        for (int s = 0; s < 10; ++s) slotsw[(s * EPW + el) * 4 + sub] = (s < zlead) ? 0.0f : (((s - zlead) < n0) ? v0 : v1);
        if (__builtin_expect(wave_dense, 0)) {  // DENSE lanes overwrite their entries from the ring (kept out of the common path)
#pragma unroll 1
            for (int s = 0; s < 10; ++s) {
                int ph = clk.head + s; ph = ph >= TACO_RING_SLOTS ? ph - TACO_RING_SLOTS : ph;
                const float4 r = buf_ld4(rR, voff, (uint32_t)ph * row_bytes);  // physical row is wave-uniform: coalesced
                const bool fresh = (s >= dlen) && (s < dlen + T);             // covered by this step's write: from registers
                if (dense) slotsw[(s * EPW + el) * 4 + sub] = fresh ? ac : pick4(sub, r.x, r.y, r.z, r.w);
            }
        }
    }
    if (!dense) {  // push the run (T, this action)
        q_lens |= (T - 9) << (2 * q_m);
        if (q_m == 0) q_rem0 = T;
        q_m += 1;
    }
    if (wave_dense) {
        if (dense && active) {
            // FA:327-330: write the action into logical slots [L, L+T) & [0,100)
            for (int j = 0; j < 11; ++j) {
                int sl = dlen + j;
                if (j < T && sl < TACO_RING_SLOTS) {
                    int ph = clk.head + sl; ph = ph >= TACO_RING_SLOTS ? ph - TACO_RING_SLOTS : ph;
                    buf_st4(rR, act4, (uint32_t)ph * row_bytes + voff, 0u);
                }
            }
            const int dlen_after = dlen + T;
            // FA:378 leaves logical slots [90,100) untouched by the shift.  With the ring advancing by 10 (host side), the
            // new logical [90,100) are the physical slots just consumed, so they must be given the values of the old
            // logical [90,100) (= new [80,90)); a slot this step's action write covered is taken from registers.
            // Done ahead of the substeps so that its traffic drains while the loop computes.
#pragma unroll
            for (int j = 0; j < 10; ++j) {
                int src = clk.head + 90 + j; src = src >= TACO_RING_SLOTS ? src - TACO_RING_SLOTS : src;
                int dst = clk.head + j;  // old logical j == new logical 90+j
                dst = dst >= TACO_RING_SLOTS ? dst - TACO_RING_SLOTS : dst;
                const bool fresh = (90 + j >= dlen) && (90 + j < dlen_after);
                float4 old = buf_ld4(rR, voff, (uint32_t)src * row_bytes);
                old = sel4(fresh, act4, old);
                buf_st4(rR, old, voff, (uint32_t)dst * row_bytes);
            }
        }
    }
    dlen += T;

    // ------------------------------------------------------------------ 10 x (mid_physics_step + simulate) VT:309-313
    TACO_STAMP(2);  // pre-phase done
    const bool bat_on = (fl & TACO_F_BATTERY_CONSUMPTION) != 0;
    // the target pose is consumed only after the loop; issuing the loads here hides their latency behind the substeps
    const float4 c_tp = CLD(C_TGT_POS), c_tq = CLD(C_TGT_QUAT);
    V3 wb = quat_rotate(conj(q), w);  // body-frame angular velocity of the root state (FA:350); row I carries it from here on
    // Round-trip mode: the world-frame rate simulate() left in the root state, w = R(q) b -- what the NEXT refresh_state re-derives b from, and,
    // after the tenth substep, what the root state keeps (rounds 3-4 rebuilt it once more from the round-tripped b: one rotation pair too many,
    // ~80 % of the stored rates 1-4 ulp off the reference's own, tests/golden/glue_*_ieee.npz).  Every path through the loops below assigns it.
    V3 w_rt{0.0f, 0.0f, 0.0f};
    // `fin` (wave-uniform): every value the rate PID and the allocator of the next substep consume -- the pending actions of this step,
    // the body rates, the PID memory -- is finite and below 1e30 in every lane.  Then no intermediate of that block can be a NaN or
    // overflow, and the selects that carry torch's NaN semantics (clamp pass-through, the NaN-propagating max) are dead: the block runs
    // its FIN form (v_med3 / v_max alone), which is the same function on such inputs.  The PID memory stays clamped, the body rates stay
    // bounded as long as the integrator takes its common form (|b| h/2 <= 0.5), so `fin` survives a substep unless integrate() reports
    // a rare form; DENSE wavefronts (slot values straight from the ring) never take the FIN form.
    in_mag = in_mag + ((absf(wb.x) + absf(wb.y)) + absf(wb.z)) + ((absf(pid_prev[0]) + absf(pid_prev[1])) + absf(pid_prev[2])) +
             ((absf(pid_int[0]) + absf(pid_int[1])) + absf(pid_int[2]));
    const bool fin = !wave_dense && __builtin_amdgcn_ballot_w64(!(in_mag < 1e30f)) == 0;
    // `plain` (launch-uniform): no rotor noise, the battery model is on and PhysX runs its default two sub-iterations; together with the
    // wave-uniform euler/unwrap switch (flip envs) the PLAIN forms of the loop have these decided at compile time (no branches over the unused blocks, integrator
    // unrolled; a lone wavefront pays ~10 issue slots for every taken branch)
    const Consts Crt{P.dt, P.rdt, P.h, P.half_h, P.inv_m, P.g, P.J0, P.J1, P.J2, P.hJi0, P.hJi1, P.hJi2, P.arm_x, P.arm_y};
    // (rotor noise: PLAIN forms exist for it where the noise table is served -- bit 4 of the form; + bit 3: euler/unwrap on or off)
    // The PLAIN forms are the DEFAULT arithmetic: the body rates make the reference's round trip through the root state after every simulate()
    // (TACO_F_WORLD_RATE_ROUNDTRIP, FA:350); the "carried" mode (flag cleared) runs the general forms.
    // (the instantiations without role wavefronts -- no one to serve a noise table -- draw the rotor noise inside their PLAIN forms, behind a launch-uniform branch)
    const bool plain1 = (!(fl & TACO_F_ROTOR_NOISE) || noise_served || NOISE_INLINE) && (fl & TACO_F_WORLD_RATE_ROUNDTRIP) != 0 && P.substeps == 2 && (fl & TACO_F_BATTERY_CONSUMPTION) != 0 &&
                        (!(PAIR && LIN_SERVED) || lin_served);
    // one lane per env: the PLAIN forms also carry the default airframe's constants as literals (kPlainConsts: the handle has exactly those)
    const bool plain = plain1 && (LPE != 1 || (fl & kPlainConsts) != 0);
    if constexpr (LPE == 1) {
        int noise_have = 0;   // highest rotor-noise counter value seen (NOISE_TAB)
        if (euler_served) {   // the unwrap memory goes to the euler server (published by the first MB_POST(3, ...))
#pragma unroll
            for (int k = 0; k < 3; ++k) { eu_io[k * 64 + lane] = rpy_old[k]; eu_io[(3 + k) * 64 + lane] = rpy_cont[k]; }
        }
        // one substep; FIN: see `fin` above.  Returns whether the integrator took a rare form in some lane (wave-uniform).
        auto substep = [&](auto fin_c, const int ks) -> bool {
            // loop forms: 0 exact, 1 FIN, 2 FIN + PLAIN, 10 FIN + PLAIN with euler/unwrap (wavefronts that hold flip envs)
            constexpr int MODE = decltype(fin_c)::value;
            constexpr bool FIN = MODE >= 1, PLAIN = (MODE & 2) != 0;
            Consts C;  // PLAIN: the default airframe as literals; otherwise the handle's values (SGPR operands)
            if constexpr (PLAIN) C = kDefaultConsts; else C = Crt;
            // refresh_state, the part the inner loop consumes FA:339-350
            // (CAP: the throughput instantiation has ONE PLAIN form -- a second one for wavefronts that hold flip envs costs it 28 spilled VGPRs --
            // with the euler / unwrap block behind a wave-uniform branch: four wavefronts share the SIMD there, a branch is cheap)
            const bool eul = (PLAIN && !CAP) ? (MODE & 8) != 0 : wave_tracks_rpy;
            if (EULER_TAB && eul) {
                reinterpret_cast<float4 *>(eq_tab)[ks * 64 + lane] = make_float4(q.x, q.y, q.z, q.w);   // the euler server takes it from here
                MB_POST(3, ks + 1);
            } else if (eul) {
                V3 e = euler_xyz_v1(q);
                unwrap(e.x, rpy_old[0], rpy_cont[0]);
                unwrap(e.y, rpy_old[1], rpy_cont[1]);
                unwrap(e.z, rpy_old[2], rpy_cont[2]);
            }
            Q4 cq = conj(q);
            V3 vb = quat_rotate(cq, v);
            // delayed action FA:366: slot min(L-1, ks); L >= 9 here, so only ks == 9 can be clipped (to slot 8)
            const int idx = (dlen - 1 < ks) ? dlen - 1 : ks;
            const float4 dact = slots[idx * EPW + el];
            const float d0 = dact.x, d1 = dact.y, d2 = dact.z, d3 = dact.w;
            float thr[4];
            {
                // angular_vel_control FA:637-650
                float u0 = (d0 + 1.0f) / 2.0f * 1000.0f;
                float u1 = pid_axis<FIN>(C.dt, C.rdt, 27.5f, d1 * 20.0f, wb.x, pid_prev[0], pid_int[0]);
                float u2 = pid_axis<FIN>(C.dt, C.rdt, 50.0f, d2 * 20.0f, wb.y, pid_prev[1], pid_int[1]);
                float u3 = pid_axis<FIN>(C.dt, C.rdt, 200.0f, d3 * 20.0f, wb.z, pid_prev[2], pid_int[2]);
                // control_allocator CTRL/fpv_dynamics.py:35-46
                u3 = clampf(u3, -u0 / 2.0f, u0 / 2.0f);
                float f0 = ((u0 * 1.0f + u1 * -1.0f) + u2 * 1.0f) + u3 * -1.0f;
                float f1 = ((u0 * 1.0f + u1 * -1.0f) + u2 * -1.0f) + u3 * 1.0f;
                float f2 = ((u0 * 1.0f + u1 * 1.0f) + u2 * -1.0f) + u3 * -1.0f;
                float f3 = ((u0 * 1.0f + u1 * 1.0f) + u2 * 1.0f) + u3 * 1.0f;
                float mx = f0 - 1000.0f, t1 = f1 - 1000.0f, t2 = f2 - 1000.0f, t3 = f3 - 1000.0f;
                float ex;
                if constexpr (FIN) {  // no NaN, and x - 1000 is never -0: plain maxima are the same function
                    mx = __builtin_fmaxf(__builtin_fmaxf(__builtin_fmaxf(mx, t1), t2), t3);
                    ex = __builtin_fmaxf(mx, 0.0f);
                } else {
                    mx = (t1 > mx || t1 != t1) ? t1 : mx;
                    mx = (t2 > mx || t2 != t2) ? t2 : mx;
                    mx = (t3 > mx || t3 != t3) ? t3 : mx;
                    ex = (mx < 0.0f) ? 0.0f : mx;
                }
                thr[0] = clamp_const_t<FIN>(f0 - ex, 100.0f, 1000.0f); thr[1] = clamp_const_t<FIN>(f1 - ex, 100.0f, 1000.0f);
                thr[2] = clamp_const_t<FIN>(f2 - ex, 100.0f, 1000.0f); thr[3] = clamp_const_t<FIN>(f3 - ex, 100.0f, 1000.0f);
            }
            // mechanical power FA:614
            float Pm;
            {
                float b = TACO_DIVC(omega[0] * 2.0f * kPi, 4500.0f); Pm = 400.0f * ((b * b) * b);
                b = TACO_DIVC(omega[1] * 2.0f * kPi, 4500.0f); Pm = Pm + 400.0f * ((b * b) * b);
                b = TACO_DIVC(omega[2] * 2.0f * kPi, 4500.0f); Pm = Pm + 400.0f * ((b * b) * b);
                b = TACO_DIVC(omega[3] * 2.0f * kPi, 4500.0f); Pm = Pm + 400.0f * ((b * b) * b);
            }
            // Battery_Dynamics.sim_process CTRL/battery_dynamics.py:47-75
            if (PLAIN || bat_on) {
                bat_t = bat_t + C.dt;
                float p_c = TACO_DIVC(TACO_DIVC(Pm, 0.75f), 9000.0f);
                bat_E = bat_E + p_c * C.dt;
                float P_avg = bat_E / bat_t;
                float r0_ = 0.0015778f + -7.7608e-5f * P_avg + (float)(0.0069498 * 1500.0);
                float r0 = (r0_ > 4.5f) ? r0_ : 4.5f;
                float uo = 4.35f + -0.1102178f * bat_E + 0.0103368f * (bat_E * bat_E) + -4.3778e-4f * ((bat_E * bat_E) * bat_E);
                float u1_dot = TACO_DIVC(0.00104846f * p_c - bat_u1, 3.3f);
                bat_u1 = bat_u1 + u1_dot * C.dt;
                float dd = uo - bat_u1;
                float rad = dd * dd - 4.0f * r0 * p_c;
                bat_V = 0.5f * (dd + __builtin_sqrtf(rad)) * 6.0f;
            } else {
                bat_V = 4.35f * 6.0f;
            }
            // RotorDynamics.sim_process CTRL/thrust_dynamics.py:98-104
            {
                float y = TACO_DIVC(bat_V - 23.0f, 3.0f);
    #pragma unroll
                for (int k = 0; k < 4; ++k) {
                    float x = TACO_DIVC(thr[k], 1000.0f);
                    float target = (opara[0] * 1.0f + opara[1] * x + opara[2] * y + opara[3] * (x * x) + opara[4] * x * y) * 100.0f;
                    omega[k] = omega[k] + (1.0f / tau[k]) * 0.001f * (target - omega[k]);
                }
            }
            if (NOISE_TAB && noise_served) {  // CTRL/thrust_dynamics.py:68-78, the factors drawn by the obs role wavefront (wave-uniform switch)
                if (noise_have < ks + 1) {   // (the server runs several substeps ahead: normally one read of the counter per step)
                    MB_WAIT(2, ks + 1);
                    noise_have = MB_SEQ(2);
                }
                const float4 f4 = reinterpret_cast<const float4 *>(rn_tab)[ks * EPW + el];
                omega[0] = omega[0] * f4.x; omega[1] = omega[1] * f4.y; omega[2] = omega[2] * f4.z; omega[3] = omega[3] * f4.w;
            } else if ((!PLAIN || NOISE_INLINE) && (fl & TACO_F_ROTOR_NOISE)) {  // CTRL/thrust_dynamics.py:68-78
                U4 r = philox(P.seed_lo, P.seed_hi, (uint32_t)gid, clk.step, STREAM_ROTOR, (uint32_t)ks);
                const float n_sc = (float)((1 + 10.0 / 700) - (1 - 10.0 / 700)), n_lo = (float)(1 - 10.0 / 700);
                omega[0] = omega[0] * (n_sc * uniform(r.x) + n_lo);
                omega[1] = omega[1] * (n_sc * uniform(r.y) + n_lo);
                omega[2] = omega[2] * (n_sc * uniform(r.z) + n_lo);
                omega[3] = omega[3] * (n_sc * uniform(r.w) + n_lo);
            }
            // AeroDynamics.sim_process CTRL/thrust_dynamics.py:173-199 + real->sim re-index CTRL/fpv_dynamics.py:48-56
            float rf[4], rt[4];
    #pragma unroll
            for (int k = 0; k < 4; ++k) { rf[k] = cf * omega[k] * omega[k]; rt[k] = ct * rf[k]; }
            float vxy = norm2(vb.x, vb.y);
            V3 F, tq;
            {
                const float fs0 = rf[2], fs1 = rf[3], fs2 = rf[0], fs3 = rf[1];
                const float ts0 = -rt[2], ts1 = rt[3], ts2 = -rt[0], ts3 = rt[1];
                F.x = dx * vb.x;
                F.y = dy * vb.y;
                F.z = kt * vxy * vxy + ((fs0 + fs1) + (fs2 + fs3));
                tq.x = C.arm_y * ((fs0 + fs1) - (fs2 + fs3));
                tq.y = -C.arm_x * ((fs0 - fs1) - (fs2 - fs3));
                tq.z = (ts0 + ts1) + (ts2 + ts3);
            }
            if (wave_has_reset) {  // (wave-uniform: normally skipped)
                if (is_reset) { F = V3{0.0f, 0.0f, 0.0f}; tq = V3{0.0f, 0.0f, 0.0f}; }  // FA:629-630: no force during the reset step
            }
            const bool rare = integrate<PLAIN ? 2 : 0>(C, P.substeps, p, q, v, wb, F, tq);
            if (PLAIN || (fl & TACO_F_WORLD_RATE_ROUNDTRIP)) {  // the reference's data flow (FA:350): the rates pass through the root state
                const V3 ww = quat_sandwich(q, wb);
                wb = quat_rotate(conj(q), ww);
                w_rt = ww;
            }
            return rare;
        };
        // the FIN form of the loop runs while `fin` holds (normally all ten substeps), the exact form takes over at the first rare form
        int ks = 0;
        // FUSED, 0..9 pending slots somewhere in the wavefront: the actor's action arrives at the top of the substep that reads slot min(dlen) (see the
        // quad loop's late_patch below: the same protocol, a float4 per env and slot here).  A non-finite action hands over to the exact form.
        bool force_exact = false;
        auto late_patch = [&]() {
            fetch_action();
            act[0] = clampf(a_in.x, -P.clip_act, P.clip_act); act[1] = clampf(a_in.y, -P.clip_act, P.clip_act);
            act[2] = clampf(a_in.z, -P.clip_act, P.clip_act); act[3] = clampf(a_in.w, -P.clip_act, P.clip_act);
            act4 = make_float4(act[0], act[1], act[2], act[3]);
            if (active) buf_st4(rH, act4, voff, (uint32_t)(clk.hh & (HIST_ROWS - 1)) * row_bytes);  // this step's action, row hh
#pragma unroll
            for (int sl = 0; sl < 10; ++sl)
                if (sl >= dlen0) slots[sl * EPW + el] = act4;
            __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");   // (same wavefront writes and reads the table: in order)
            const float mag = (absf(act[0]) + absf(act[1])) + (absf(act[2]) + absf(act[3]));
            force_exact = __builtin_amdgcn_ballot_w64(!(mag < 1e30f)) != 0;
        };
        if (fin && plain && (CAP || !wave_tracks_rpy)) {
    #pragma unroll 1
            while (ks < 10) {
                if constexpr (FUSED) { if (ks == patch_at) { late_patch(); if (force_exact) break; } }
                const bool rare = substep(std::integral_constant<int, 2>{}, ks);
                ++ks;
                if (rare) break;
            }
        } else if (!CAP && fin && plain) {  // (the throughput instantiation keeps two forms: a third costs it 28 spilled VGPRs)
    #pragma unroll 1
            while (ks < 10) {
                if constexpr (FUSED) { if (ks == patch_at) { late_patch(); if (force_exact) break; } }
                const bool rare = substep(std::integral_constant<int, 10>{}, ks);
                ++ks;
                if (rare) break;
            }
        } else if (BLOCK != 256 && !CAP && fin) {  // (the 128-VGPR instantiations keep two or three forms: one more costs them spills)
    #pragma unroll 1
            while (ks < 10) {
                const bool rare = substep(std::integral_constant<int, 1>{}, ks);
                ++ks;
                if (rare) break;
            }
        }
    #pragma unroll 1
        for (; ks < 10; ++ks) {
            if constexpr (FUSED) { if (ks == patch_at && !have_act) late_patch(); }
            substep(std::integral_constant<int, 0>{}, ks);
        }
        if (euler_served) {   // the attitude after the tenth substep (refresh_state of post_physics_step, FA:382)
            reinterpret_cast<float4 *>(eq_tab)[10 * 64 + lane] = make_float4(q.x, q.y, q.z, q.w);
            MB_POST(3, 11);
        }

    } else {
        // ---- quad layout: scatter (lane j keeps component j), run the ten substeps, gather back
        const QuadLoopP LP{pin_s(P.h), pin_s(P.half_h), pin_s(P.inv_m), b_substeps};   // (see QuadLoopP)
        const float dt_q = pin_s(P.dt), rdt_q = pin_s(P.rdt);
        float pq = pick4(sub, p.x, p.y, p.z, 0.0f), qq = pick4(sub, q.x, q.y, q.z, q.w), vq = pick4(sub, v.x, v.y, v.z, 0.0f);
        float bq = pick4(sub, wb.x, wb.y, wb.z, 0.0f);
        float ppq = pick4(sub, pid_prev[0], pid_prev[1], pid_prev[2], 0.0f), piq = pick4(sub, pid_int[0], pid_int[1], pid_int[2], 0.0f);
        float omq = pick4(sub, omega[0], omega[1], omega[2], omega[3]);
        float roq = pick4(sub, rpy_old[0], rpy_old[1], rpy_old[2], 0.0f), rcq = pick4(sub, rpy_cont[0], rpy_cont[1], rpy_cont[2], 0.0f);
        const float itq = (1.0f / pick4(sub, tau[0], tau[1], tau[2], tau[3])) * 0.001f;
        const float kpq = pick4(sub, 27.5f, 50.0f, 200.0f, 0.0f);
        const float Jq = pick4(sub, P.J0, P.J1, P.J2, 0.0f), hJiq = pick4(sub, P.hJi0, P.hJi1, P.hJi2, 0.0f);
        const float gzq = pick4(sub, -0.0f, -0.0f, P.g, -0.0f);
        const float dragq = pick4(sub, dx, dy, 0.0f, 0.0f);
        const uint32_t sgn = 0x80000000u;
        const uint32_t m1 = (sub < 2) ? sgn : 0u;               // allocator column of u1: - - + +
        const uint32_t m2 = (sub == 1 || sub == 2) ? sgn : 0u;  //                     u2: + - - +
        const uint32_t m3 = (sub == 0 || sub == 2) ? sgn : 0u;  //                     u3: - + - +   (also the sign of ts: -rt0 rt1 -rt2 rt3)
        const uint32_t sm3 = (sub == 3) ? sgn : 0u;
        const uint32_t m2_lane1 = (sub == 1) ? sgn : 0u, sgn_lane0 = (sub == 0) ? sgn : 0u, sgn_lane2 = (sub == 2) ? sgn : 0u;  // euler, see the loop
        const uint32_t k2 = (sub == 2) ? ~0u : 0u, k3 = (sub == 3) ? ~0u : 0u, keven = (sub & 1) ? 0u : ~0u;
        const uint32_t keep = is_reset ? 0u : ~0u;  // FA:629-630: no force during the reset step
        const float Sq = pick4(sub, P.arm_y, -P.arm_x, 1.0f, 1.0f);
        const float *slotsf = reinterpret_cast<const float *>(slots);
        // the delayed action of substep k + 1 is fetched during substep k: its LDS latency would otherwise sit at the head of every substep
        auto slot_of = [&](int k) { int r = (dlen - 1 < k) ? dlen - 1 : k; return r > 9 ? 9 : r; };
        float dq_next = slotsf[(slot_of(0) * EPW + el) * 4 + sub];
        float fix_p = 0.0f, fix_q = 0.0f, fix_v = 0.0f, fix_b = 0.0f, fix_F = 0.0f, fix_t = 0.0f;   // the integrator's inputs of the running substep (PLAIN forms)
        float wq_rt = 0.0f;   // (w_rt in the quad layout)
        auto roundtrip = [&]() {  // FA:350 literally: w = R(q) b (quat_sandwich), b = quat_rotate(conj(q), w), lane j = component j
            const float ww = bc3(qq), q1 = rot1(qq), q2 = rot2(qq);
            float t = fma(q1, rot2(bq), -(q2 * rot1(bq)));
            t = t + t;
            const float wq = fma(q1, rot2(t), fma(-q2, rot1(t), fma(ww, t, bq)));
            wq_rt = wq;
            bq = from_bits(bits(quad_rotate(-qq, ww, wq)) & ~k3);  // (lane 3 holds no rate component: kept at +0)
        };
        auto redo_integrate = [&](int ks_of) {   // the PLAIN forms' rare case: see the end of `substep`; ks_of: that substep
            pq = fix_p; qq = fix_q; vq = fix_v; bq = fix_b;
            float q_mid = 0.0f;
            integrate_quad<2, false, LIN_SERVED>(LP, k3, pq, qq, vq, bq, fix_F, fix_t, Jq, hJiq, gzq, sm3, q_mid);
            if constexpr (LIN_SERVED) { lin_ring[(ks_of * 3 + 1) * 64 + lane] = q_mid; lin_ring[(ks_of * 3 + 3) * 64 + lane] = qq; }   // (not published yet: the counter follows at the top of the next substep)
            roundtrip();
        };
        if (lin_served) {   // the linear chain's start values (published by the first substep's MB_POST(5, 1))
            lin_init[lane] = pq; lin_init[64 + lane] = vq; lin_init[128 + lane] = dragq; lin_init[192 + lane] = kt; lin_init[256 + lane] = from_bits(keep);
            lin_ring[lane] = qq;
        }
        auto substep = [&](auto fin_c, const int ks) -> bool {
            // loop forms: 0 exact, 1 FIN, 2 FIN + PLAIN (battery computed here), 6 FIN + PLAIN + battery served by wavefront 1; + 8: with
            // euler/unwrap (wavefronts that hold flip envs)
            constexpr int MODE = decltype(fin_c)::value;
            constexpr bool FIN = (MODE & 1) != 0 || MODE >= 2, PLAIN = (MODE & 2) != 0, NOISY = (MODE & 16) != 0;   // NOISY: PLAIN + served rotor noise
            const bool served = PLAIN ? (MODE & 4) != 0 : bat_served;
            if (EULER_TAB && (PLAIN ? (MODE & 8) != 0 : wave_tracks_rpy)) {
                eq_tab[ks * 64 + lane] = qq;   // the euler server (states role wavefront) takes it from here
                MB_POST(3, ks + 1);
            } else if (PLAIN ? (MODE & 8) != 0 : wave_tracks_rpy) {
                // refresh_state FA:339-347, get_euler_xyz_v1 TU:175-196 with lane j evaluating angle j: roll and yaw are the same atan2 on
                // different operands (lanes 0 and 2), pitch the asin form (lane 1) -- one atan2 and one asin per substep instead of two and
                // one.  With c = this lane's quaternion component: numerator 2 (w c +- c' c'') (c', c'' the next two components; minus in
                // lane 1: a - b == a + (-b)), denominator ((ww -+ xx) - yy) +- zz; operation order as in the scalar euler_xyz_v1.
                const float wq = bc3(qq), sq = qq * qq;
                const float num = 2.0f * (wq * qq + xorf(rot1(qq) * rot2(qq), m2_lane1));
                const float den = ((bc3(sq) + xorf(bc0(sq), sgn_lane0)) - bc1(sq)) + xorf(bc2(sq), sgn_lane2);
                const float at = atan2(num, den);
                float pit;
                if (absf(num) >= 1.0f) pit = kHalfPi * (num > 0.0f ? 1.0f : (num < 0.0f ? -1.0f : 0.0f));
                else pit = asin(num);
                unwrap(sub == 1 ? pit : at, roq, rcq);
            }
            const bool lin_off = PLAIN ? LIN_SERVED : lin_served;   // the linear chain runs in the obs role wavefront (PLAIN: two sub-iterations by definition)
            // the voltage of this substep has been in the making in wavefront 1 since the previous substep's rotor update (~850 clocks ago; it needs
            // ~650): counter and value are read HERE, at the top, ~45 instructions ahead of their use (LDS executes a wavefront's reads in order, so
            // a value read after a sufficient counter is the right one)
            int seq_early = 0;
            float v_early = 0.0f;
            if (served) {
                asm volatile("" ::: "memory");
                seq_early = MB_SEQ(1);
                v_early = __hip_atomic_load(&mb_v[el], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            }
            float vbq = 0.0f;
            if (!lin_off) vbq = quad_rotate(-qq, bc3(qq), vq);  // FA:350 body-frame linear velocity (here: its instructions fill the LDS latency of the reads around it)
            const float dq = dq_next;  // slot min(L - 1, ks), FA:366
            dq_next = slotsf[(slot_of(ks + 1) * EPW + el) * 4 + sub];
            const float d0 = bc0(dq);
            // angular_vel_control FA:637-650: lane j runs PID axis j on action component j + 1
            const float u0 = (d0 + 1.0f) / 2.0f * 1000.0f;
            const float upid = pid_axis<FIN>(dt_q, rdt_q, kpq, dppf<QP(1, 2, 3, 3)>(dq) * 20.0f, bq, ppq, piq);
            // control_allocator CTRL/fpv_dynamics.py:35-46: lane k computes motor k
            const float u1 = bc0(upid), u2 = bc1(upid);
            float u3 = bc2(upid);
            u3 = clampf(u3, -u0 / 2.0f, u0 / 2.0f);
            const float fq = ((u0 * 1.0f + xorf(u1, m1)) + xorf(u2, m2)) + xorf(u3, m3);
            const float tm = fq - 1000.0f;
            float thrq;
            if constexpr (FIN) {  // no NaN, and x - 1000 is never -0: plain maxima are the same function
                const float mx = __builtin_fmaxf(__builtin_fmaxf(__builtin_fmaxf(bc0(tm), bc1(tm)), bc2(tm)), bc3(tm));
                thrq = __builtin_amdgcn_fmed3f(fq - __builtin_fmaxf(mx, 0.0f), 100.0f, 1000.0f);
            } else {
                float mx = bc0(tm);
                { const float t1 = bc1(tm); mx = (t1 > mx || t1 != t1) ? t1 : mx; }
                { const float t2 = bc2(tm); mx = (t2 > mx || t2 != t2) ? t2 : mx; }
                { const float t3 = bc3(tm); mx = (t3 > mx || t3 != t3) ? t3 : mx; }
                const float ex = (mx < 0.0f) ? 0.0f : mx;
                thrq = clamp_const(fq - ex, 100.0f, 1000.0f);
            }
            if (served) {
                asm volatile("" : "+v"(seq_early), "+v"(v_early));  // keeps the compiler from waiting for the early reads before this point
                if (__builtin_expect(__builtin_amdgcn_ballot_w64(seq_early < ks + 1) != 0, 0)) {  // (not posted yet at the early read: rare)
                    MB_WAIT(1, ks + 1);
                    v_early = mb_v[el];
                }
                bat_V = v_early;
            } else {
                // mechanical power FA:614
                float Pm;
                {
                    const float b = TACO_DIVC(omq * 2.0f * kPi, 4500.0f);
                    const float c3 = 400.0f * ((b * b) * b);
                    Pm = bc0(c3);
                    Pm = Pm + bc1(c3);
                    Pm = Pm + bc2(c3);
                    Pm = Pm + bc3(c3);
                }
                // Battery_Dynamics.sim_process CTRL/battery_dynamics.py:47-75 (replicated in the quad)
                if (PLAIN || bat_on) bat_V = battery_step(dt_q, Pm, bat_E, bat_u1, bat_t);
                else bat_V = 4.35f * 6.0f;
            }
            // RotorDynamics.sim_process CTRL/thrust_dynamics.py:98-104: lane k = rotor k
            {
                const float y = TACO_DIVC(bat_V - 23.0f, 3.0f);
                const float x = TACO_DIVC(thrq, 1000.0f);
                const float target = (opara[0] * 1.0f + opara[1] * x + opara[2] * y + opara[3] * (x * x) + opara[4] * x * y) * 100.0f;
                omq = omq + itq * (target - omq);
            }
            if constexpr (NOISY) {
                omq = omq * rn_tab[(ks * EPW + el) * 4 + sub];   // CTRL/thrust_dynamics.py:68-78, the factor drawn by the obs role wavefront
            } else if ((!PLAIN || NOISE_INLINE) && (fl & TACO_F_ROTOR_NOISE)) {  // CTRL/thrust_dynamics.py:68-78
                if (noise_served) {
                    omq = omq * rn_tab[(ks * EPW + el) * 4 + sub];
                } else {
                    U4 r = philox(P.seed_lo, P.seed_hi, (uint32_t)gid, clk.step, STREAM_ROTOR, (uint32_t)ks);
                    const float n_sc = (float)((1 + 10.0 / 700) - (1 - 10.0 / 700)), n_lo = (float)(1 - 10.0 / 700);
                    const uint32_t rk = sub == 0 ? r.x : (sub == 1 ? r.y : (sub == 2 ? r.z : r.w));
                    omq = omq * (n_sc * uniform(rk) + n_lo);
                }
            }
            if (served) {  // the rotor speeds the NEXT substep's battery step starts from
                mb_om[lane] = omq;
                MB_POST(0, ks + 2);
            }
            // AeroDynamics.sim_process CTRL/thrust_dynamics.py:173-199 + real->sim re-index CTRL/fpv_dynamics.py:48-56
            // fs = (rf2 rf3 rf0 rf1), ts = (-rt2 rt3 -rt0 rt1): the pair sums live in lanes (2,3) and (0,1)
            float Fq = 0.0f, tqq;
            {
                const float rf = cf * omq * omq;
                const float rt = ct * rf;
                const float psum = rf + swp(rf);          // lanes 0,1: rf0 + rf1   lanes 2,3: rf2 + rf3
                const float pdif = rf - swp(rf);          // lane 0: rf0 - rf1      lane 2: rf2 - rf3
                const float srt = xorf(rt, m3);           // -rt0 rt1 -rt2 rt3
                const float pz = srt + swp(srt);          // lanes 0,1: -rt0 + rt1  lanes 2,3: -rt2 + rt3
                // tq.x = arm_y ((fs0+fs1) - (fs2+fs3)), tq.y = -arm_x ((fs0-fs1) - (fs2-fs3)): lane 0 / 1 read the pair terms of
                // M = (pdif psum pdif psum) from lanes (3,1) / (2,0); tq.z = (ts0+ts1) + (ts2+ts3)
                const float M = selm(keven, pdif, psum);
                const float txy = Sq * (dppf<QP(3, 2, 2, 2)>(M) - dppf<QP(1, 0, 0, 0)>(M));
                const float tz = bc2(pz) + bc0(pz);
                tqq = from_bits(bits(selm(k2, tz, txy)) & keep);
                if (lin_off) {
                    lin_ring[(ks * 3 + 2) * 64 + lane] = psum;
                } else {
                    const float vxy = norm2(bc0(vbq), bc1(vbq));
                    const float Fz = kt * vxy * vxy + (bc2(psum) + bc0(psum));
                    Fq = from_bits(bits(selm(k2, Fz, dragq * vbq)) & keep);
                }
            }
            bool rare;
            if constexpr (PLAIN) {
                // The integrator's four rare-form tests (two per sub-iteration) cost a lone wavefront ~30 clocks each as branches (VALU compare ->
                // SGPR -> s_cbranch, nothing else to issue meanwhile): the PLAIN forms run the common forms unconditionally, OR the four tests
                // together and leave the ONE branch to the loop (run_form: "rare" ends it); a wavefront that did meet a rare form redoes the
                // integrator from the substep's start values with the branching form (redo_integrate: same values as before, lane by lane).
                fix_p = pq; fix_q = qq; fix_v = vq; fix_b = bq; fix_F = Fq; fix_t = tqq;
                float q_mid = 0.0f;
                rare = integrate_quad<2, true, LIN_SERVED>(LP, k3, pq, qq, vq, bq, Fq, tqq, Jq, hJiq, gzq, sm3, q_mid);
                if constexpr (LIN_SERVED) { lin_ring[(ks * 3 + 1) * 64 + lane] = q_mid; lin_ring[(ks * 3 + 3) * 64 + lane] = qq; }   // (+ 3: the next substep's start attitude)
                roundtrip();
                asm volatile("" ::"v"(pq), "v"(qq), "v"(vq), "v"(bq));   // (or the compiler branches around the rest of the substep as soon as `rare` is known)
            } else {
                float q_mid = 0.0f;
                if (lin_off) {
                    rare = integrate_quad<0, false, true>(LP, k3, pq, qq, vq, bq, Fq, tqq, Jq, hJiq, gzq, sm3, q_mid);
                    lin_ring[(ks * 3 + 1) * 64 + lane] = q_mid; lin_ring[(ks * 3 + 3) * 64 + lane] = qq;
                } else {
                    rare = integrate_quad<0>(LP, k3, pq, qq, vq, bq, Fq, tqq, Jq, hJiq, gzq, sm3, q_mid);
                }
                if (fl & TACO_F_WORLD_RATE_ROUNDTRIP) roundtrip();
            }
            return rare;
        };
        if (euler_served) { eu_io[lane] = roq; eu_io[64 + lane] = rcq; }   // (published by the first MB_POST(3, ...))
        int ks = 0;
        // FUSED, 0..9 pending slots somewhere in the wavefront: the actor's action arrives here, at the top of the substep that reads slot
        // min(dlen) (whose delayed action, fetched during the previous substep, is fetched again).  The new action joins what the PID consumes: if it is not
        // finite the FIN / PLAIN forms hand over to the exact one, as they do for every other input.
        bool force_exact = false;
        auto late_patch = [&]() {
            fetch_action();
            act[0] = clampf(a_in.x, -P.clip_act, P.clip_act); act[1] = clampf(a_in.y, -P.clip_act, P.clip_act);
            act[2] = clampf(a_in.z, -P.clip_act, P.clip_act); act[3] = clampf(a_in.w, -P.clip_act, P.clip_act);
            act4 = make_float4(act[0], act[1], act[2], act[3]);
            if (active) buf_st4(rH, act4, voff, (uint32_t)(clk.hh & (HIST_ROWS - 1)) * row_bytes);  // this step's action, row hh
            const float ac = pick4(sub, act[0], act[1], act[2], act[3]);
            float *slotsw = reinterpret_cast<float *>(slots);
#pragma unroll
            for (int sl = 0; sl < 10; ++sl)
                if (sl >= dlen0) slotsw[(sl * EPW + el) * 4 + sub] = ac;
            __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");   // (same wavefront writes and reads the table: in order)
            dq_next = slotsf[(slot_of(ks) * EPW + el) * 4 + sub];
            const float mag = (absf(act[0]) + absf(act[1])) + (absf(act[2]) + absf(act[3]));
            force_exact = __builtin_amdgcn_ballot_w64(!(mag < 1e30f)) != 0;
        };
        auto run_form = [&](auto form) {
            bool rare = false;
            if (ks < 10) {
                bool more;   // (one wave-uniform flag, bitwise: `rare || ks == 10` as control flow is a second branch per substep)
#pragma unroll 1
                do {
                    if constexpr (FUSED) { if (ks == patch_at) late_patch(); }
                    if (((decltype(form)::value & 2) != 0 && LIN_SERVED) || lin_served) MB_POST(5, ks + 1);   // the linear chain may take substep ks - 1 (ks == 0: its start values)
                    rare = substep(form, ks);
                    ++ks;
                    more = !(bool)((int)rare | (int)(ks >= 10) | (int)(FUSED && force_exact));
                } while (more);
            }
            if constexpr ((decltype(form)::value & 2) != 0) { if (rare) redo_integrate(ks - 1); }
        };
        if (noise_served) {   // the noise table is complete before the first substep reads it (posted ~a pre-phase ago: normally no wait)
            MB_WAIT(2, 1);
        }
        if (NOISE_TAB && fin && plain && noise_served) {   // (the PLAIN forms with the served rotor noise compiled in)
            if constexpr (NOISE_TAB) {
                if (bat_served) {
                    if (!wave_tracks_rpy) run_form(std::integral_constant<int, 22>{});
                    else run_form(std::integral_constant<int, 30>{});
                } else {
                    if (!wave_tracks_rpy) run_form(std::integral_constant<int, 18>{});
                    else run_form(std::integral_constant<int, 26>{});
                }
            }
        } else if (SPLIT && fin && plain && bat_served) {
            if (!wave_tracks_rpy) run_form(std::integral_constant<int, 6>{});
            else run_form(std::integral_constant<int, 14>{});
        } else if (fin && plain) {
            if (!wave_tracks_rpy) run_form(std::integral_constant<int, 2>{});
            else run_form(std::integral_constant<int, 10>{});
        } else if (fin) {
            run_form(std::integral_constant<int, 1>{});
        }
#pragma unroll 1
        for (; ks < 10; ++ks) {
            if constexpr (FUSED) { if (ks == patch_at) late_patch(); }
            if (lin_served) MB_POST(5, ks + 1);
            substep(std::integral_constant<int, 0>{}, ks);
        }
        if (lin_served) MB_POST(5, 11);   // ... and the tenth
        if (bat_served) {
            MB_WAIT(1, 11);
            bat_E = mb_bs[el * 4]; bat_u1 = mb_bs[el * 4 + 1]; bat_t = mb_bs[el * 4 + 2];
            if (__builtin_expect(mb_timeout, 0)) {  // never silently: sticky status + a voltage no consumer can mistake for a result
                if (lane == 0) atomicOr(&P.ctl[kCtlStatus], kStatusMailboxTimeout);
                bat_V = nanf32();
            }
        }
        p = V3{bc0(pq), bc1(pq), bc2(pq)};
        q = Q4{bc0(qq), bc1(qq), bc2(qq), bc3(qq)};
        v = V3{bc0(vq), bc1(vq), bc2(vq)};
        wb = V3{bc0(bq), bc1(bq), bc2(bq)};
        w_rt = V3{bc0(wq_rt), bc1(wq_rt), bc2(wq_rt)};
        pid_prev[0] = bc0(ppq); pid_prev[1] = bc1(ppq); pid_prev[2] = bc2(ppq);
        pid_int[0] = bc0(piq); pid_int[1] = bc1(piq); pid_int[2] = bc2(piq);
        omega[0] = bc0(omq); omega[1] = bc1(omq); omega[2] = bc2(omq); omega[3] = bc3(omq);
        if (euler_served) {   // the attitude after the tenth substep (refresh_state of post_physics_step, FA:382), then the server's result
            eq_tab[10 * 64 + lane] = qq;
            MB_POST(3, 11);
        }
        rpy_old[0] = bc0(roq); rpy_old[1] = bc1(roq); rpy_old[2] = bc2(roq);
        rpy_cont[0] = bc0(rcq); rpy_cont[1] = bc1(rcq); rpy_cont[2] = bc2(rcq);
    }

    TACO_STAMP(3);  // substeps done
    if (!ROLES) load_out_args();   // (this wavefront runs the post-phase itself)
    if (FUSED && !bat_served && __builtin_expect(mb_timeout, 0)) {  // (the wait for the actor's action gave up; with the battery served this is reported above)
        if (lane == 0) atomicOr(&P.ctl[kCtlStatus], kStatusMailboxTimeout);
        bat_V = nanf32();
    }
    // root state: world-frame angular velocity -- what the tenth simulate() left there (round trip), or rebuilt once from the carried body rates
    if (fl & TACO_F_WORLD_RATE_ROUNDTRIP) w = w_rt; else w = quat_sandwich(q, wb);

    // ------------------------------------------------------------------ post_physics_step FA:374-388
    progress += 1;
    if (euler_served) {   // all eleven refreshes of the step were the server's
        if constexpr (EULER_TAB) {
            MB_WAIT(4, 1);
            if constexpr (LPE == 4) {
                const float ro = eu_io[lane], rc = eu_io[64 + lane];
                rpy_old[0] = bc0(ro); rpy_old[1] = bc1(ro); rpy_old[2] = bc2(ro);
                rpy_cont[0] = bc0(rc); rpy_cont[1] = bc1(rc); rpy_cont[2] = bc2(rc);
            } else {
#pragma unroll
                for (int k = 0; k < 3; ++k) { rpy_old[k] = eu_io[k * 64 + lane]; rpy_cont[k] = eu_io[(3 + k) * 64 + lane]; }
            }
            if (__builtin_expect(mb_timeout, 0)) {
                if (lane == 0) atomicOr(&P.ctl[kCtlStatus], kStatusMailboxTimeout);
                rpy_cont[0] = nanf32();
            }
        }
    } else if (wave_tracks_rpy) {  // refresh_state FA:382 (euler + unwrap part)
        V3 e = euler_xyz_v1(q);
        unwrap(e.x, rpy_old[0], rpy_cont[0]);
        unwrap(e.y, rpy_old[1], rpy_cont[1]);
        unwrap(e.z, rpy_old[2], rpy_cont[2]);
    }
    const float roll_cont = rpy_cont[0];
    if (grp == TACO_TASK_FLIP) cmd1 = clampf(flip_radian - roll_cont, -kTwoPi, kTwoPi);  // FA:831-832 / :930-931
    K.p = p; K.v = v; K.w = w; K.q = q; K.bat_V = bat_V; K.cmd0 = cmd0; K.cmd1 = cmd1; K.flip_radian = flip_radian; K.progress = progress;
    K.act[0] = act[0]; K.act[1] = act[1]; K.act[2] = act[2]; K.act[3] = act[3]; K.c_tp = c_tp; K.c_tq = c_tq;
    if (ROLES && sub == 0) {  // hand the Carry to the three role wavefronts
        float4 *h4 = reinterpret_cast<float4 *>(hand + el * CARRY_WORDS);
        if (lin_served) { hand[el * CARRY_WORDS + 3] = bat_V; hand[el * CARRY_WORDS + 7] = cmd0; }   // (p and v come from the wavefront that ran the linear chain)
        else { h4[0] = make_float4(p.x, p.y, p.z, bat_V); h4[1] = make_float4(v.x, v.y, v.z, cmd0); }
        h4[2] = make_float4(w.x, w.y, w.z, cmd1);
        h4[3] = make_float4(q.x, q.y, q.z, q.w); h4[4] = make_float4(act[0], act[1], act[2], act[3]);
        h4[5] = make_float4(flip_radian, as_f(progress), 0.0f, 0.0f); h4[6] = c_tp; h4[7] = c_tq;
    }
    if (ROLES) __syncthreads();  // barrier 2 of 2 (every wavefront executes exactly the same two; the served pair form has nothing to hand over: one barrier)
    if constexpr (FUSED) {
        if (!have_act) {  // (wave-uniform) the common case: the actor finished under the substeps -- barrier 2 is its completion too
            fetch_action();
            act4 = make_float4(clampf(a_in.x, -P.clip_act, P.clip_act), clampf(a_in.y, -P.clip_act, P.clip_act), clampf(a_in.z, -P.clip_act, P.clip_act),
                               clampf(a_in.w, -P.clip_act, P.clip_act));
            if (active) buf_st4(rH, act4, voff, (uint32_t)(clk.hh & (HIST_ROWS - 1)) * row_bytes);  // this step's action, row hh
        }
    }
    // The target pose was loaded up front and is first READ by the post-phase, behind the state stores below.  gfx9 has one vmcnt counter for
    // loads and stores, retired in order, and the stores sit in an exec-masked block the wavefront may skip -- so the wait the compiler put ahead
    // of that first read was vmcnt(0), and a wavefront that runs the post-phase itself sat out the round trip of its own state stores before it
    // started on the frame.  Pinned here, the wait comes ahead of the stores, where those loads have long landed (quad forms: 13.59 -> 13.47 us at
    // 4 096 envs, flip 16 384 16.94 -> 16.74; the one-lane forms, whose SIMDs hold other wavefronts to run meanwhile, lose 1 % to the earlier wait
    // and keep the compiler's placement: profiles/r04_s_ab_post_phase.txt).
    if constexpr (!ROLES && LPE == 4)
        asm volatile("" : "+v"(K.c_tp.x), "+v"(K.c_tp.y), "+v"(K.c_tp.z), "+v"(K.c_tp.w), "+v"(K.c_tq.x), "+v"(K.c_tq.y), "+v"(K.c_tq.z), "+v"(K.c_tq.w));
    if (active) {
        // Everything the substep loop evolved goes back to its SoA row NOW, so the registers are free for the
        // observation / reward code below.
        CST(C_QUAT, make_float4(q.x, q.y, q.z, q.w));
        CST(C_ANGVEL, make_float4(w.x, w.y, w.z, bat_V));
        CST(C_PID_PREV, make_float4(pid_prev[0], pid_prev[1], pid_prev[2], bat_E));
        CST(C_PID_INT, make_float4(pid_int[0], pid_int[1], pid_int[2], bat_u1));
        CST(C_OMEGA, make_float4(omega[0], omega[1], omega[2], omega[3]));
        if (track_rpy) {
            CST(C_RPY_OLD, make_float4(rpy_old[0], rpy_old[1], rpy_old[2], 0.0f));
            CST(C_RPY_CONT, make_float4(rpy_cont[0], rpy_cont[1], rpy_cont[2], 0.0f));
        }
        {   // consume ten slots: implied zeros first, then at most two runs (rem0 >= 1 and every full run >= 9 slots)
            int c = 10;
            const int zu = zlead < c ? zlead : c;
            zlead -= zu; c -= zu;
#pragma unroll
            for (int rep = 0; rep < 2; ++rep) {
                if (c > 0 && q_m > 0) {
                    const int take = q_rem0 < c ? q_rem0 : c;
                    q_rem0 -= take; c -= take;
                    if (q_rem0 == 0) {
                        q_lens = (int)((uint32_t)q_lens >> 2);
                        q_m -= 1;
                        q_rem0 = (q_m > 0) ? run_len(q_lens, 0) : 0;
                    }
                }
            }
        }
        dlen = dlen - 10 < 0 ? 0 : dlen - 10;
        if (lin_served) {   // (the x y z words are stored by the wavefront that ran the linear chain)
            llvm_amdgcn_raw_buffer_store_f32(as_f((int)make_pw(progress, q_lens)), rS, (int)(voff + 12u + (uint32_t)C_POS * row_bytes), 0, 0);
            llvm_amdgcn_raw_buffer_store_f32(as_f((int)make_dw(dlen, zlead, q_m, q_rem0, q_lens, dense)), rS, (int)(voff + 12u + (uint32_t)C_LINVEL * row_bytes), 0, 0);
        } else {
            CST(C_POS, make_float4(p.x, p.y, p.z, as_f((int)make_pw(progress, q_lens))));
            CST(C_LINVEL, make_float4(v.x, v.y, v.z, as_f((int)make_dw(dlen, zlead, q_m, q_rem0, q_lens, dense))));
        }
        CST(C_MISC, make_float4(bat_t, cmd0, cmd1, flip_radian));
    }
    publish_clock();
    if (ROLES) { TACO_STAMP(4); TACO_STAMP(5); return; }  // wave 0 is done; the roles below belong to the other three wavefronts
    if constexpr (PAIR && LIN_SERVED) {
        if (lin_served) {   // p and v of the step: from the wavefront that ran the linear chain (its last substep ends ~500 clocks after this wavefront's: under the stores above)
            MB_WAIT(6, 1);
            K.p = V3{hand[el * CARRY_WORDS], hand[el * CARRY_WORDS + 1], hand[el * CARRY_WORDS + 2]};
            K.v = V3{hand[el * CARRY_WORDS + 4], hand[el * CARRY_WORDS + 5], hand[el * CARRY_WORDS + 6]};
            if (__builtin_expect(mb_timeout, 0)) {
                if (lane == 0) atomicOr(&P.ctl[kCtlStatus], kStatusMailboxTimeout);
                K.p.x = nanf32();
            }
        }
    }
    } else {
        load_out_args();
        if (wv == 1 && lane < 8) mb_seq[lane] = 0;
        // (DRAWS_SERVED: this wavefront's reset draws were made at the top of the kernel, "up-front loads")
        __syncthreads();  // barrier 1 of 2
        if constexpr (NOISE_TAB) {
            if (wv == (ROLES ? 2 : 1) && noise_served) {   // the step's 10 x EPW rotor-noise blocks -> rn_tab (first: wavefront 0 needs them at its first substep)
                const float n_sc = (float)((1 + 10.0 / 700) - (1 - 10.0 / 700)), n_lo = (float)(1 - 10.0 / 700);
#pragma unroll 1
                for (int b = lane; b < 10 * EPW; b += 64) {
                    const int ks = b / EPW, e = b - ks * EPW;
                    const int ie = (int)wave_env0 + e < P.n ? (int)wave_env0 + e : P.n - 1;   // (tail slots shadow the last env, as wavefront 0's lanes do)
                    const U4 r = philox(P.seed_lo, P.seed_hi, (uint32_t)(P.env_offset + ie), clk.step, STREAM_ROTOR, (uint32_t)ks);
                    rn_tab[b * 4 + 0] = n_sc * uniform(r.x) + n_lo; rn_tab[b * 4 + 1] = n_sc * uniform(r.y) + n_lo;
                    rn_tab[b * 4 + 2] = n_sc * uniform(r.z) + n_lo; rn_tab[b * 4 + 3] = n_sc * uniform(r.w) + n_lo;
                    if constexpr (LPE == 1) MB_POST(2, ks + 1);   // (one pass of the loop = one substep's blocks: published as they are drawn)
                }
                if constexpr (LPE == 4) MB_POST(2, 1);
            }
        }
        if (wv == 3 && P.len_states > 1) shift_history(A.states, A.states_prev, P.states_bytes, P.len_states, A.states_out, P.clip_states);
        if (wv == 2 && P.len_obs > 1) shift_history(A.obs, A.obs_prev, P.obs_bytes, P.len_obs, A.obs_out, P.clip_obs);
        if constexpr (EULER_TAB) {
            if (LPE == 1 && wv == euler_role && euler_served) {   // one lane per env: the scalar get_euler_xyz_v1 + three unwraps per attitude
                MB_WAIT(3, 1);
                float ro[3], rc[3];
#pragma unroll
                for (int k = 0; k < 3; ++k) { ro[k] = eu_io[k * 64 + lane]; rc[k] = eu_io[(3 + k) * 64 + lane]; }
#pragma unroll 1
                for (int k = 0; k < 11; ++k) {
                    MB_WAIT(3, k + 1);
                    const float4 q4 = reinterpret_cast<const float4 *>(eq_tab)[k * 64 + lane];
                    const V3 e = euler_xyz_v1(Q4{q4.x, q4.y, q4.z, q4.w});
                    unwrap(e.x, ro[0], rc[0]);
                    unwrap(e.y, ro[1], rc[1]);
                    unwrap(e.z, ro[2], rc[2]);
                }
                __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
#pragma unroll
                for (int k = 0; k < 3; ++k) { eu_io[k * 64 + lane] = ro[k]; eu_io[(3 + k) * 64 + lane] = rc[k]; }
                MB_POST(4, 1);
            }
            if (LPE == 4 && wv == euler_role && euler_served) {   // the euler server: eleven attitudes -> get_euler_xyz_v1 + unwrap, lane j of an env's quad = angle j
                const uint32_t sgn = 0x80000000u;
                const uint32_t m2_lane1 = (sub == 1) ? sgn : 0u, sgn_lane0 = (sub == 0) ? sgn : 0u, sgn_lane2 = (sub == 2) ? sgn : 0u;
                MB_WAIT(3, 1);
                float roq = eu_io[lane], rcq = eu_io[64 + lane];
#pragma unroll 1
                for (int k = 0; k < 11; ++k) {
                    MB_WAIT(3, k + 1);
                    const float qq = eq_tab[k * 64 + lane];
                    const float wq = bc3(qq), sq = qq * qq;
                    const float num = 2.0f * (wq * qq + xorf(rot1(qq) * rot2(qq), m2_lane1));
                    const float den = ((bc3(sq) + xorf(bc0(sq), sgn_lane0)) - bc1(sq)) + xorf(bc2(sq), sgn_lane2);
                    const float at = atan2(num, den);
                    float pit;
                    if (absf(num) >= 1.0f) pit = kHalfPi * (num > 0.0f ? 1.0f : (num < 0.0f ? -1.0f : 0.0f));
                    else pit = asin(num);
                    unwrap(sub == 1 ? pit : at, roq, rcq);
                }
                __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");   // (wavefront 0 is past its reads of eu_io: it posted counter 3 after them)
                eu_io[lane] = roq; eu_io[64 + lane] = rcq;
                MB_POST(4, 1);
            }
        }
        if (wv == 2 && (fl & TACO_F_OBSERVATION_NOISE)) { gen_obs_noise(); noise_ready = true; }
        if constexpr (LIN_SERVED) {
            if (wv == (ROLES ? 2 : 1) && lin_served) {   // the linear chain, behind wavefront 0 (see lin_ring)
                MB_WAIT(5, 1);
                float pq = lin_init[lane], vq = lin_init[64 + lane];
                const float dragq = lin_init[128 + lane], kt_ = lin_init[192 + lane];
                const uint32_t keep = bits(lin_init[256 + lane]);
                const uint32_t k2 = (sub == 2) ? ~0u : 0u;
                const float gzq = pick4(sub, -0.0f, -0.0f, P.g, -0.0f);
                const QuadLoopP LP{pin_s(P.h), pin_s(P.half_h), pin_s(P.inv_m), 2};   // (see QuadLoopP; the server runs exactly two sub-iterations)
#pragma unroll 1
                for (int ks = 0; ks < 10; ++ks) {
                    // (the substep's start attitude came with the previous counter: body-frame velocity and drag are done when thrust and q1 arrive)
                    const float q0 = lin_ring[(ks * 3) * 64 + lane];
                    const float vbq = quad_rotate(-q0, bc3(q0), vq);  // FA:350 body-frame linear velocity
                    const float vxy = norm2(bc0(vbq), bc1(vbq));
                    const float kvv = kt_ * vxy * vxy, dv = dragq * vbq;
                    MB_WAIT(5, ks + 2);
                    const float q1 = lin_ring[(ks * 3 + 1) * 64 + lane], psum = lin_ring[(ks * 3 + 2) * 64 + lane];
                    const float Fz = kvv + (bc2(psum) + bc0(psum));
                    const float Fq = from_bits(bits(selm(k2, Fz, dv)) & keep);
                    linear_iteration_quad(LP, q0, Fq, gzq, vq, pq);
                    linear_iteration_quad(LP, q1, Fq, gzq, vq, pq);
                }
                if (sub < 3) {   // lane j holds component j: the Carry's p and v, and their words of the state
                    hand[el * CARRY_WORDS + sub] = pq; hand[el * CARRY_WORDS + 4 + sub] = vq;
                    if (in_range) {
                        llvm_amdgcn_raw_buffer_store_f32(pq, rS, (int)(voff + 4u * (uint32_t)sub + (uint32_t)C_POS * row_bytes), 0, 0);
                        llvm_amdgcn_raw_buffer_store_f32(vq, rS, (int)(voff + 4u * (uint32_t)sub + (uint32_t)C_LINVEL * row_bytes), 0, 0);
                    }
                }
                if (__builtin_expect(mb_timeout, 0)) {
                    if (lane == 0) atomicOr(&P.ctl[kCtlStatus], kStatusMailboxTimeout);
                    if (sub < 3) hand[el * CARRY_WORDS + sub] = nanf32();
                }
                if constexpr (PAIR) MB_POST(6, 1);   // wavefront 0 runs the post-phase itself: p and v are in `hand`
            }
        }
        if (bat_served && wv == 1) {  // battery server (the reward wavefront, idle until the post-phase): ten voltages, each one substep ahead of wavefront 0
            MB_WAIT(0, 1);
            float bE = mb_bs[el * 4], bu1 = mb_bs[el * 4 + 1], bt = mb_bs[el * 4 + 2];
            const float dt_s = pin_s(P.dt);
            for (int ks = 0; ks < 10; ++ks) {
                MB_WAIT(0, ks + 1);
                float Pm;
                {   // mechanical power FA:614, summed in rotor order as the quad form does
                    float b = TACO_DIVC(mb_om[el * 4] * 2.0f * kPi, 4500.0f); Pm = 400.0f * ((b * b) * b);
                    b = TACO_DIVC(mb_om[el * 4 + 1] * 2.0f * kPi, 4500.0f); Pm = Pm + 400.0f * ((b * b) * b);
                    b = TACO_DIVC(mb_om[el * 4 + 2] * 2.0f * kPi, 4500.0f); Pm = Pm + 400.0f * ((b * b) * b);
                    b = TACO_DIVC(mb_om[el * 4 + 3] * 2.0f * kPi, 4500.0f); Pm = Pm + 400.0f * ((b * b) * b);
                }
                const float V = battery_step(dt_s, Pm, bE, bu1, bt);
#ifdef TACO_TEST_HOOKS
                if (P.flags & kDebugSlowServer) { for (int z = 0; z < 40; ++z) __builtin_amdgcn_s_sleep(100); }  // test hook: arrive late
#endif
                if (sub == 0) mb_v[el] = V;
                if (ks == 9 && sub == 0) { mb_bs[el * 4] = bE; mb_bs[el * 4 + 1] = bu1; mb_bs[el * 4 + 2] = bt; }
                MB_POST(1, ks == 9 ? 11 : ks + 1);
            }
        }
        // (wavefront 0 rewrites this workgroup's clock pair when it is done: this wavefront's scalar reads of it have landed long ago -- made explicit)
        if (from_ctl) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if constexpr (PAIR) return;   // (the serving wavefront of the two-wavefront form has no post-phase role and nothing to wait for)
        __syncthreads();  // barrier 2 of 2
        const float4 *h4 = reinterpret_cast<const float4 *>(hand + el * CARRY_WORDS);
        const float4 a0 = h4[0], a1 = h4[1], a2 = h4[2], a3 = h4[3], a4 = h4[4], a5 = h4[5];
        K.p = V3{a0.x, a0.y, a0.z}; K.bat_V = a0.w; K.v = V3{a1.x, a1.y, a1.z}; K.cmd0 = a1.w; K.w = V3{a2.x, a2.y, a2.z}; K.cmd1 = a2.w;
        K.q = Q4{a3.x, a3.y, a3.z, a3.w}; K.act[0] = a4.x; K.act[1] = a4.y; K.act[2] = a4.z; K.act[3] = a4.w;
        if constexpr (FUSED) {  // the step wavefront may not have had the action when it wrote the Carry: the roles take it from the actor's mailbox
#pragma unroll
            for (int k = 0; k < 4; ++k) K.act[k] = clampf(FX.act_lds[el * 4 + k], -P.clip_act, P.clip_act);
        }
        K.flip_radian = a5.x; K.progress = as_i(a5.y); K.c_tp = h4[6]; K.c_tq = h4[7];
    }
    // ------------------------------------------------------------------ post-phase: works on the Carry only
    const V3 &p = K.p, &v = K.v, &w = K.w;
    const Q4 &q = K.q;
    const float bat_V = K.bat_V, cmd0 = K.cmd0, cmd1 = K.cmd1;
    const float (&act)[4] = K.act;
    const int progress = K.progress;
    const float4 c_tp = K.c_tp, c_tq = K.c_tq;
    // relative quantities FA:354-360 (target velocities are identically zero); the target pose is only needed from here on
    const V3 pt{c_tp.x, c_tp.y, c_tp.z};
    const Q4 qt{c_tq.x, c_tq.y, c_tq.z, c_tq.w};
    const Q4 cq = conj(q);
    const bool frames = roleS || roleO;  // (SPLIT: the reward wavefront needs only R00, and only for flip envs)
    float fr[26];
    float m[9];
    float tilt00 = 1.0f;  // R00 of the noise-free relative attitude: the flip reward's x_tiltage
    V3 rel_pos{0.0f, 0.0f, 0.0f}, rel_v{0.0f, 0.0f, 0.0f}, rel_pos_b{0.0f, 0.0f, 0.0f};
    Q4 rel_q_b{0.0f, 0.0f, 0.0f, 1.0f};
    // ---- the frame built in the QUAD LAYOUT (QUAD_POST).  The four lanes of an env used to build the same 26 words each
    // (~300 instructions of a wavefront that runs alone on its SIMD) and lane 0 wrote them to the tile.  Lane j now holds component j of every
    // vector (as the substep loop does): ONE quat_rotate stream rotates the three relative vectors' component j (quad_rotate: the scalar code's
    // operation sequence per component, DPP operands), lane j evaluates ROW j of quaternion_to_matrix (TJ:389-416 is cyclic in (i j k): the
    // diagonal, the element right of it and the one left of it are one expression each under j -> j + 1; products and sums commute exactly),
    // the constant divisions run once with the scalar words (battery, commands) riding in lane 3, and every lane writes its own words to the
    // tile (seven ds_write_b32).  Same operations on the same operands word for word: the tile holds the same bits.
    bool tile_filled = false;
    if constexpr (QUAD_POST) {
        qf = frames && !(roleO && ((fl & TACO_F_OBSERVATION_NOISE) || A.gather));   // (noise and the gather block work on the scalar frame)
        if (qf) {
            QF.p = pick4(sub, p.x, p.y, p.z, bat_V); QF.v = pick4(sub, v.x, v.y, v.z, cmd0); QF.w = pick4(sub, w.x, w.y, w.z, cmd1);
            QF.a = pick4(sub, act[0], act[1], act[2], act[3]); QF.tp = pick4(sub, pt.x, pt.y, pt.z, 0.0f);
        }
        if (qf) {
            const uint32_t k3 = (sub == 3) ? ~0u : 0u;
            const float cqv = -pick4(sub, q.x, q.y, q.z, q.w), qw = q.w;
            const float relp = QF.tp - QF.p, relv = 0.0f - QF.v, relw = 0.0f - QF.w;   // component j in lanes 0..2
            const float rpb = quad_rotate(cqv, qw, relp), rvb = quad_rotate(cqv, qw, relv), rwb = quad_rotate(cqv, qw, relw);
            rel_q_b = quat_mul(cq, qt);
            const float rq = pick4(sub, rel_q_b.x, rel_q_b.y, rel_q_b.z, rel_q_b.w), rr = rel_q_b.w;
            const float two_s = 2.0f / (((rel_q_b.x * rel_q_b.x + rel_q_b.y * rel_q_b.y) + rel_q_b.z * rel_q_b.z) + rr * rr);
            const float q1 = rot1(rq), q2 = rot2(rq);
            const float mD = 1.0f - two_s * (q1 * q1 + q2 * q2);   // m[4 j]
            const float mU = two_s * (rq * q1 - q2 * rr);           // m[3 j + (j + 1) % 3]
            const float mL = two_s * (rq * q2 + q1 * rr);           // m[3 j + (j + 2) % 3]
            const float f23 = 4.0f * clampf(bc2(QF.p), 0.0f, 0.5f) - 1.0f;
            const float y1 = TACO_DIVC(selm(k3, QF.p - 23.0f, rpb), 3.0f);   // fr[0..2] | fr[18]
            const float y2 = selm(k3, QF.v, rvb / 2.0f);                     // fr[12..14] | fr[24]
            const bool l3_rot = sub == 3 && grp == TACO_TASK_ROTATE;
            const float x3 = selm(k3, grp == TACO_TASK_FLIP ? QF.w / 2.0f : QF.w, rwb);
            float y3 = div_const(x3, l3_rot ? 6.0f : kPi, l3_rot ? 1.0f / 6.0f : 1.0f / kPi);   // fr[15..17] | fr[25]
            y3 = (sub == 3 && grp == TACO_TASK_POS) ? QF.w : y3;
            // word of the frame each lane writes, one byte per lane of the quad
            const uint32_t sh = 8u * (uint32_t)sub;
            auto wd = [&](uint32_t packed) { return (int)((packed >> sh) & 0xffu); };
            float *const trow = tile + el * 26;
            __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");   // (the tile is private to the wavefront: only the compiler needs the fences)
            trow[wd(0x12020100u)] = y1;                              // 0 1 2 | 18
            trow[wd(0x170b0703u)] = selm(k3, f23, mD);               // 3 7 11 | 23
            trow[wd(0x17090804u)] = selm(k3, f23, mU);               // 4 8 9 | 23 (again: the same value)
            trow[wd(0x170a0605u)] = selm(k3, f23, mL);               // 5 6 10 | 23
            trow[wd(0x180e0d0cu)] = y2;                              // 12 13 14 | 24
            trow[wd(0x1911100fu)] = y3;                              // 15 16 17 | 25
            trow[19 + sub] = QF.a;                                   // 19 .. 22
            __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
            tile_filled = true;
            if (roleR) {   // (one-wavefront forms: the reward below takes what it shares with the frame from here)
                rel_pos_b = V3{bc0(rpb), bc1(rpb), bc2(rpb)};
                tilt00 = bc0(mD);
            }
        }
    }
    if (roleR || !tile_filled) {
        rel_pos = V3{pt.x - p.x, pt.y - p.y, pt.z - p.z};
        rel_v = V3{0.0f - v.x, 0.0f - v.y, 0.0f - v.z};
    }
    if (!tile_filled) {
    const V3 rel_w{0.0f - w.x, 0.0f - w.y, 0.0f - w.z};
    rel_pos_b = quat_rotate(cq, rel_pos);
    if (frames || wave_tracks_rpy) {
        rel_q_b = quat_mul(cq, qt);
        quat_to_matrix(rel_q_b, m);
        tilt00 = m[0];
    }
    if (frames) {
        const V3 rel_v_b = quat_rotate(cq, rel_v);
        const V3 rel_w_b = quat_rotate(cq, rel_w);
        // 26-D frame FA:415-421 + task tails
        fr[0] = TACO_DIVC(rel_pos_b.x, 3.0f); fr[1] = TACO_DIVC(rel_pos_b.y, 3.0f); fr[2] = TACO_DIVC(rel_pos_b.z, 3.0f);
#pragma unroll
        for (int k = 0; k < 9; ++k) fr[3 + k] = m[k];
        fr[12] = rel_v_b.x / 2.0f; fr[13] = rel_v_b.y / 2.0f; fr[14] = rel_v_b.z / 2.0f;
        fr[15] = TACO_DIVC(rel_w_b.x, kPi); fr[16] = TACO_DIVC(rel_w_b.y, kPi); fr[17] = TACO_DIVC(rel_w_b.z, kPi);
        fr[18] = TACO_DIVC(bat_V - 23.0f, 3.0f);
#pragma unroll
        for (int k = 0; k < 4; ++k) fr[19 + k] = act[k];
        fr[23] = 4.0f * clampf(p.z, 0.0f, 0.5f) - 1.0f;
        fr[24] = cmd0;
        fr[25] = (grp == TACO_TASK_POS) ? cmd1 : (grp == TACO_TASK_ROTATE ? TACO_DIVC(cmd1, 6.0f) : TACO_DIVC(cmd1 / 2.0f, kPi));
    }
    }

    // ---- frame stacks [env][len][26] (newest frame last).  Two store paths:
    //  len == 1 : the wavefront's 64 rows are one contiguous 6 656-byte range.  The frames go through a per-wave LDS tile
    //             (lane-major registers -> env-major bytes) and leave as seven fully coalesced 16-byte-per-lane stores.
    //  len  > 1 : each lane shifts its own row by one frame (8-byte accesses, in place, ascending) and appends the frame.
    // Stores go through range-checked buffer descriptors, so ragged tails need no special casing.
    // fill_tile = false: the tile already holds this frame (the obs frame equals the states frame when there is no observation noise)
    auto put_frame = [&](float *buf, const float *prev, uint32_t buf_bytes, int len, const float (&f)[26], bool fill_tile, float *out, float clip) {
        const rsrc_t rB = make_rsrc(buf, buf_bytes);
        char *const ob = reinterpret_cast<char *>(out);
        if (fill_tile) {
            // the scratch is private to the wavefront and its LDS operations execute in order: only the compiler needs the fence
            __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
            f32x2_t *t2 = reinterpret_cast<f32x2_t *>(tile + el * 26);  // 104-byte rows: 8-byte aligned
            if (sub == 0) {
#pragma unroll
                for (int k = 0; k < 13; ++k) t2[k] = f32x2_t{f[2 * k], f[2 * k + 1]};
            }
            __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
        }
        if (len == 1) {
            const f32x4_t *t4 = reinterpret_cast<const f32x4_t *>(tile);
            constexpr uint32_t TILE_W4 = (uint32_t)EPW * 26u / 4u;  // 16-byte words of the wave's contiguous range (416 or 104)
#pragma unroll
            for (int it = 0; it < (int)((TILE_W4 + 63u) / 64u); ++it) {
                const uint32_t w4 = (uint32_t)(it * 64 + lane);
                if (w4 < TILE_W4) llvm_amdgcn_raw_buffer_store_v4f32(t4[w4], rB, (int)(wave_env0 * 104u + w4 * 16u), 0, 0);
            }
            if (OUT && out) {
#pragma unroll
                for (int it = 0; it < (int)((TILE_W4 + 63u) / 64u); ++it) {
                    const uint32_t w4 = (uint32_t)(it * 64 + lane), off = wave_env0 * 104u + w4 * 16u;
                    if (w4 < TILE_W4 && off < buf_bytes) *reinterpret_cast<f32x4_t *>(ob + off) = clamp4(t4[w4], clip);
                }
            }
        } else if (ROLES) {
            // the history words were moved by shift_history while the substeps ran: only the newest frame of every row is left
            const f32x2_t *tp = reinterpret_cast<const f32x2_t *>(tile);
            const uint32_t L13 = (uint32_t)len * 13u, hist = L13 - 13u;
            const uint32_t base = wave_env0 * L13;
#pragma unroll
            for (int it = 0; it < (EPW * 13 + 63) / 64; ++it) {
                const uint32_t t = (uint32_t)(it * 64 + lane);  // word t of the tile = word (t % 13) of row t / 13
                const uint32_t row = t / 13u;
                if (t < (uint32_t)EPW * 13u) llvm_amdgcn_raw_buffer_store_v2f32(tp[t], rB, (int)((base + row * L13 + hist + (t - row * 13u)) * 8u), 0, 0);
            }
            if (OUT && out) {
#pragma unroll
                for (int it = 0; it < (EPW * 13 + 63) / 64; ++it) {
                    const uint32_t t = (uint32_t)(it * 64 + lane), row = t / 13u, off = (base + row * L13 + hist + (t - row * 13u)) * 8u;
                    if (t < (uint32_t)EPW * 13u && off < buf_bytes) *reinterpret_cast<f32x2_t *>(ob + off) = clamp2(tp[t], clip);
                }
            }
        } else {
            // one wavefront does everything: history words from the previous stack, the newest frame from the LDS tile, in one pass
            const rsrc_t rP = make_rsrc(prev, buf_bytes);
            const f32x2_t *tp = reinterpret_cast<const f32x2_t *>(tile);
            const uint32_t L13 = (uint32_t)len * 13u, hist = L13 - 13u, tot = (uint32_t)EPW * L13;
            const uint32_t magic = 0xFFFFFFFFu / L13 + 1u;  // row = (w * magic) >> 32 for w < 2^16
            const uint32_t base = wave_env0 * L13;          // first word of the wavefront's range
            for (uint32_t w0 = 0; w0 < tot; w0 += 8u * 64u) {
                f32x2_t v[8];
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    const uint32_t w = w0 + (uint32_t)k * 64u + (uint32_t)lane;
                    const uint32_t row = __umulhi(w, magic), wl = w - row * L13;
                    v[k] = f32x2_t{0.0f, 0.0f};
                    if (w < tot) {
                        if (wl < hist) v[k] = llvm_amdgcn_raw_buffer_load_v2f32(rP, (int)((base + w + 13u) * 8u), 0, 0);
                        else v[k] = tp[row * 13u + (wl - hist)];
                    }
                }
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    const uint32_t w = w0 + (uint32_t)k * 64u + (uint32_t)lane;
                    if (w < tot) llvm_amdgcn_raw_buffer_store_v2f32(v[k], rB, (int)((base + w) * 8u), 0, 0);
                    if (OUT && out && w < tot && (base + w) * 8u < buf_bytes) *reinterpret_cast<f32x2_t *>(ob + (size_t)(base + w) * 8u) = clamp2(v[k], clip);
                }
            }
        }
    };
    // states first (noise-free frame), then obs (possibly noised)
    if (roleS) {
        if (!FUSED && P.st_period > 0) {   // the frame ring behind VecTask.step()'s state stacks (StepParams.st_period): one frame, two while the window wraps
            // (OUT: a second ring of the same geometry receives clamp(frame, +-clip_states) -- the clamped copy of a stack is the stack of the
            // clamped frames, so VecTask.step()'s return value (VT:332) is the same strided view of THAT ring)
            const size_t roff = (size_t)(P.st_front + clk.phase) * (P.states_bytes / 4u), toff = (size_t)P.st_period * (P.states_bytes / 4u);
            float *const row = A.states + roff;
            float *const crow = (OUT && A.states_out) ? A.states_out + roff : nullptr;
            put_frame(row, row, P.states_bytes, 1, fr, !tile_filled, crow, P.clip_states);
            if (clk.phase >= P.st_period - P.st_front)
                put_frame(row - toff, row - toff, P.states_bytes, 1, fr, false, crow ? crow - toff : nullptr, P.clip_states);
        } else {
            put_frame(A.states, A.states_prev, P.states_bytes, P.len_states, fr, !tile_filled, A.states_out, P.clip_states);
        }
    }
    if (roleO && (fl & TACO_F_OBSERVATION_NOISE)) {  // FA:402-410
        if (!noise_ready) gen_obs_noise();
#pragma unroll
        for (int k = 0; k < 3; ++k) fr[k] = fr[k] + P.df * (nrm[k] * (float)(0.06 / 3 / 3) + 0.0f);
        quat_to_matrix(quat_mul(rel_q_b, nq), m);
#pragma unroll
        for (int k = 0; k < 9; ++k) fr[3 + k] = m[k];
#pragma unroll
        for (int k = 0; k < 3; ++k) fr[12 + k] = fr[12 + k] + P.df * (nrm[3 + k] * (float)(0.1 / 3 / 2) + 0.0f);
#pragma unroll
        for (int k = 0; k < 3; ++k) fr[15 + k] = fr[15 + k] + P.df * (nrm[6 + k] * (float)(60.0 / 3 / 180) + 0.0f);
        fr[18] = fr[18] + P.df * (nrm[9] * (float)(0.06 / 3) + 0.0f);
        fr[23] = fr[23] + P.df * (nrm[10] * (float)(0.06 / 3 / 3) + 0.0f);
    }
    if (roleO) put_frame(A.obs, A.obs_prev, P.obs_bytes, P.len_obs, fr, !tile_filled && (ROLES || (fl & TACO_F_OBSERVATION_NOISE) != 0), A.obs_out, P.clip_obs);
    if constexpr (FUSED) {
        if (roleO && sub == 0) {  // ... and into the actor's input rows: the observation the next step's action is computed from never leaves the CU
#pragma unroll
            for (int k = 0; k < 26; ++k) FX.xin[el * FX.xin_ld + k] = fr[k];
        }
    }
    // Optional all-gather block, one 128-byte-aligned row per env: [obs stack | reward | done | time-out | pad].  The obs part is written by
    // the wavefront that holds the obs frame (the newest frame from registers, older frames re-read from this env's just-written obs row),
    // the three tail words by the reward wavefront below.
    float *const gather_p = A.gather;
    if (roleO && gather_p && active) {
        const rsrc_t rG = make_rsrc(gather_p, P.gather_bytes);
        const rsrc_t rO = make_rsrc(A.obs, P.obs_bytes);
        if (P.len_obs > 1) __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");  // the obs rows were stored by other lanes of this wavefront
        const uint32_t g0 = (uint32_t)i * P.gather_row * 4u;
        const uint32_t hist_pairs = (uint32_t)(P.len_obs - 1) * 13u;
        for (uint32_t j = 0; j < hist_pairs; ++j) {
            const f32x2_t o = llvm_amdgcn_raw_buffer_load_v2f32(rO, (int)((uint32_t)i * (uint32_t)P.len_obs * 104u + j * 8u), 0, 0);
            llvm_amdgcn_raw_buffer_store_v2f32(o, rG, (int)(g0 + j * 8u), 0, 0);
        }
#pragma unroll
        for (int k = 0; k < 13; ++k)
            llvm_amdgcn_raw_buffer_store_v2f32(f32x2_t{fr[2 * k], fr[2 * k + 1]}, rG, (int)(g0 + (hist_pairs + (uint32_t)k) * 8u), 0, 0);
    }

    TACO_STAMP(4);  // state stores + frames done
    if (roleR) {
    // ------------------------------------------------------------------ compute_reward CTRL/task_reward.py
    // FUSED: the done flag FIRST (all it needs is the task's position error, the height and the progress counter): in the persistent rollout
    // kernel the step wavefront is already at the top of the NEXT step, waiting for exactly this flag.  (The launch-per-step forms keep
    // the reward's own order: the reordering costs their one-wavefront instantiations SGPR spills.)
    float rew, pos_dist = 0.0f;
    auto done_flags = [&](long long &rs_, bool &tmo_) {
        long long die = 0;
        if (p.z < 0.1f) die = 1;
        if (pos_dist > 10.0f) die = 1;
        const float max_len_f = (float)P.max_len;
        rs_ = ((float)progress >= max_len_f - 1.0f) ? 1 : die;
        tmo_ = (progress >= P.max_len - 1) && (rs_ != 0);  // VT:323
    };
    long long rs = 0;
    bool tmo = false;
    // QUAD_POST: the four lanes of an env ran the same scalar reward, a dozen IEEE divisions (~12 instructions each) four times over.  The
    // reciprocals 1 / (1 + x) of the reward terms come in pairs of pairs: lane j of the quad divides for argument j -- ONE division instead of
    // four -- and the quotients come back through DPP; the two normalisations of the rotate reward divide component j in lane j.  The same
    // IEEE operations on the same operands, whichever lane performs them.
    auto recip4 = [&](float a, float b, float c, float d, float &ea, float &eb, float &ec, float &ed) {
        if constexpr (QUAD_POST) {
            const float e = 1.0f / (1.0f + pick4(sub, a, b, c, d));
            ea = bc0(e); eb = bc1(e); ec = bc2(e); ed = bc3(e);
        } else {
            ea = 1.0f / (1.0f + a); eb = 1.0f / (1.0f + b); ec = 1.0f / (1.0f + c); ed = 1.0f / (1.0f + d);
        }
    };
    auto div3 = [&](float &x, float &y, float &z, float den) {
        if constexpr (QUAD_POST) {
            const float e = pick4(sub, x, y, z, 0.0f) / den;
            x = bc0(e); y = bc1(e); z = bc2(e);
        } else {
            x = x / den; y = y / den; z = z / den;
        }
    };
    if constexpr (FUSED) {
        if (grp == TACO_TASK_ROTATE) {
            const float hori = norm2(rel_pos.x, rel_pos.y) - 1.2f, vert = absf(rel_pos.z);
            pos_dist = __builtin_sqrtf(hori * hori + vert * vert);
        } else {
            pos_dist = norm3(rel_pos_b.x, rel_pos_b.y, rel_pos_b.z);
        }
        done_flags(rs, tmo);
        // the next step of this kernel takes its reset flags from here (tail lanes shadow env n - 1, as the step wavefront's do)
        if (sub == 0) FX.reset_lds[el] = (int)rs;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        if (lane == 0) __hip_atomic_store(FX.reset_seq, FX.reset_want + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
    if (grp == TACO_TASK_POS) {  // :20-47
        if constexpr (!FUSED) pos_dist = norm3(rel_pos_b.x, rel_pos_b.y, rel_pos_b.z);
        Q4 mq = quat_mul(q, conj(qt));  // quat_diff_rad TJ:145-164
        float nn = norm3(mq.x, mq.y, mq.z);
        nn = (nn > 1.0f) ? 1.0f : nn;
        float qd = 2.0f * asin(nn);
        float e0, e1, e2, e3;
        recip4(pos_dist * pos_dist, 10.0f * pos_dist * pos_dist, qd * qd, 10.0f * qd * qd, e0, e1, e2, e3);
        float pr = e0 + e1;
        float rr = e2 + e3;
        rew = TACO_DIVC(pr * rr, 100.0f);
    } else if (grp == TACO_TASK_ROTATE) {  // :50-104
        float nx0 = -rel_pos.x, nx1 = -rel_pos.y, nx2 = 0.0f;
        float nn = norm3(nx0, nx1, nx2) + 1e-8f;
        div3(nx0, nx1, nx2, nn);
        float ny0 = cross_term(0.0f, nx2, 1.0f, nx1), ny1 = cross_term(1.0f, nx0, 0.0f, nx2), ny2 = cross_term(0.0f, nx1, 0.0f, nx0);
        float yn = norm3(ny0, ny1, ny2) + 1e-8f;
        div3(ny0, ny1, ny2, yn);
        if constexpr (!FUSED) {
            float hori = norm2(rel_pos.x, rel_pos.y) - 1.2f;
            float vert = absf(rel_pos.z);
            pos_dist = __builtin_sqrtf(hori * hori + vert * vert);
        }
        float normal = (rel_v.x * nx0 + rel_v.y * nx1) + rel_v.z * nx2;
        float tang = (rel_v.x * ny0 + rel_v.y * ny1) + rel_v.z * ny2;
        float ld = norm3(normal - 0.0f, tang - cmd1, rel_v.z - 0.0f);
        float e0, e1, e2, e3;
        recip4(pos_dist * pos_dist, 10.0f * pos_dist * pos_dist, ld * ld, 10.0f * ld * ld, e0, e1, e2, e3);
        float pr = e0 + e1;
        float lr = e2 + e3;
        float hm[9];
        quat_to_matrix(q, hm);
        float ddir = 1.0f + (nx0 * hm[0] + nx1 * hm[3]) / norm2(hm[0], hm[3]);
        recip4(ddir * ddir, 10.0f * ddir * ddir, ddir * ddir, 10.0f * ddir * ddir, e0, e1, e2, e3);
        float dr = e0 + e1;
        rew = TACO_DIVC(pr * lr * dr, 100.0f);
    } else {  // :107-143
        if constexpr (!FUSED) pos_dist = norm3(rel_pos_b.x, rel_pos_b.y, rel_pos_b.z);
        float xr = 1.0f / (1.0f + 10.0f * (1.0f - tilt00));
        float cd = TACO_DIVC(cmd1 / 2.0f, kPi);
        float e0, e1, e2, e3;
        recip4(1.0f * pos_dist, 10.0f * pos_dist, cd * cd, 10.0f * cd * cd, e0, e1, e2, e3);
        float pr = e0 + e1;
        float cr = e2 + e3;
        rew = TACO_DIVC(pr * xr * cr, 100.0f);
    }
    if constexpr (!FUSED) done_flags(rs, tmo);

    // ------------------------------------------------------------------ store
    if (active) {
        A.rew[i] = rew;
        P.reset[i] = rs;
        A.timeout[i] = tmo ? 1 : 0;
        if (A.done_f32) A.done_f32[i] = (float)rs;
        if (gather_p) {  // tail of this env's all-gather row (the obs part is written above)
            const rsrc_t rG = make_rsrc(gather_p, P.gather_bytes);
            const uint32_t tail = (uint32_t)i * P.gather_row * 4u + (uint32_t)P.len_obs * 104u;
            llvm_amdgcn_raw_buffer_store_f32(rew, rG, (int)tail, 0, 0);
            llvm_amdgcn_raw_buffer_store_f32((float)rs, rG, (int)(tail + 4u), 0, 0);
            llvm_amdgcn_raw_buffer_store_f32(tmo ? 1.0f : 0.0f, rG, (int)(tail + 8u), 0, 0);
        }
    }
    }
    TACO_STAMP(5);
    if (SPLIT && P.stamps_on && P.stamps && blockIdx.x == 0 && lane == 0) P.stamps[5 + wv] = __builtin_readcyclecounter();  // [6..8]: the role wavefronts' ends
#undef TACO_STAMP
#undef TACO_ARG
#undef MB_WAIT
#undef MB_POST
#undef MB_SEQ
}

// the launch-per-step forms: one step_core per launch (argument list: StepKernelArgs)
template <int BLOCK, int LPE, bool SPLIT = false, bool CAP = false, bool OUT = false, bool RESET_ONLY = false, bool WIDE = false, bool LIN = false>
__global__ __launch_bounds__(BLOCK) __attribute__((amdgpu_waves_per_eu((SPLIT || CAP) ? ((WIDE || BLOCK == 128) ? 2 : 4) : 1, CAP ? 4 : 8))) void taco_step_kernel(
    float *S, long long *reset, const float *act_in, uint32_t seed_lo, uint32_t seed_hi, uint32_t step, int task_mode, int n, uint32_t flags, uint32_t early,
    int env_offset, const StepParams P0) {
    StepParams P{};   // (the preloaded fields only; step_core reads the block P0 itself, behind its up-front loads)
    const size_t npad4 = (size_t)((n + 63) & ~63) * 4;   // floats per row of the workspace arrays (taco_capi.hip taco_create: S | hist | ring | wclk)
    P.S = S; P.reset = reset; P.act_in = act_in; P.hist = S + (size_t)NUM_CHUNKS * npad4;
    P.wclk = (uint32_t *)(S + (size_t)(NUM_CHUNKS + HIST_ROWS + TACO_RING_SLOTS) * npad4);
    P.seed_lo = seed_lo; P.seed_hi = seed_hi; P.step = step; P.task_mode = task_mode; P.n = n; P.flags = flags; P.env_offset = env_offset;
    P.hh = (int)(early & 15u); P.hw_rows = (int)((early >> 8) & 7u); P.stamps_on = (int)((early >> 30) & 1u); P.use_ctl = (int)(early >> 31);
    step_core<BLOCK, LPE, SPLIT, CAP, OUT, RESET_ONLY, WIDE, false, LIN>(P, FusedCtx{});
}

}  // namespace taco
