// dbh_forward.hip — the Deepbinner forward pass as ONE persistent gfx950 kernel: at most one
// 512-thread workgroup (8 wave64s, 2 per SIMD) per CU; workgroup b starts with window b and takes
// every further window off a counter in global memory, carrying each 1024-sample window through
// all 20 convolutions with the activations resident in registers (conv1 .. conv6) or in LDS the
// whole way.  Per
// window HBM gives 2 KiB of int16 and takes n_classes floats + a call; the weights stream from L2
// by LDS-DMA; the one round trip through global memory is conv17's 16 x 48 output, parked in a
// per-workgroup slot until the batched tail runs (3 KB per window, written and read back by the
// same CU - dirty in L2, so part of it does reach HBM: ~1.2 KB per window of write traffic
// against 56 B of results, profiles/r04_v1).
//
// What it computes: reference deepbinner/network_architecture.py:18-95 as evaluated by
// model.predict (deepbinner/classify.py:361) — see oracle/network_ref.py for the operator
// semantics (TensorFlow SAME padding, valid-count average pooling, BN after ReLU/pool) — and, in
// seam-b2 mode, the slicing and z-normalisation in front of it (classify.py:337-357,
// trim_signal.py:61-69) and the renormalise + call behind it (classify.py:285-295, 387-393).
//
// How (DESIGN.md section 4 has the full account and the measurements behind each choice):
//   - every convolution is a sum of [positions x C_in] . [C_in x C_out] products on the fp32
//     matrix pipe (v_mfma_f32_16x16x4_f32: M = 16 positions, N = 16 output channels, K = 4 input
//     channels).  The pipe is shared with the vector ALU: every other vector instruction costs
//     matrix time, so the code counts them;
//   - stage B (conv1 .. conv4, 84 % of the work) and conv5, conv6 behind it are ONE CHAIN IN
//     REGISTERS (stage_b_chain): every layer runs TRANSPOSED (M = output channels = the weights as
//     the A operand, N = the wave's 16 quads), so that an MFMA leaves each lane with the output
//     channels of its own quad that the next layer's k-steps want from it; output transform, bias,
//     ReLU, the next input transform are in-lane but for one halo position each side (a DPP row
//     shift; at a wave's two ends 2 x 48 floats through LDS and a counter word the neighbours
//     poll).  No activation image and no workgroup barrier from the top of a window to conv6's
//     last store; conv1 (one input channel) is computed inside conv2's first tile the same way;
//   - the k = 3 layers with enough positions run as Winograd: F(4,3) for conv2,3,4 (L = 512, one
//     tile of 16 quads per wave, N tile by N tile with the transformed inputs in registers) and
//     conv7 (L = 256, a tile per wave PAIR, split by output channels), F(2,3) for conv6, conv8,
//     conv9, conv13, conv15 - 9,588 MFMAs per window instead of the direct form's 16,452;
//   - from conv7 on: A fragments (activations): ds_read_b64 from the [position][channel] LDS image
//     (row pitch 50 floats), B fragments (weights): ds_read_b128 / b64 from fragment-ordered copies
//     that LDS-DMA brought in a phase ahead; both issued from inline asm one step ahead with
//     hand-counted waits.  conv17's weights, used once per window, go to registers by buffer loads;
//   - there, outputs are stored in place once every wave has read its inputs (one barrier
//     mid-layer), the epilogue of an N tile inside the MFMA steps of the next one;
//   - activations are held times 2^-60 so that ReLU is the clamp modifier of the instruction that
//     produces a value (dbh_layout.h: kActScale);
//   - the average pooling in front of conv10 is applied to conv10's OUTPUT (a 1x1 convolution
//     commutes with it), in registers;
//   - the last three layers run for eight windows at a time, one wave per window (batched tail);
//   - the next window's samples, statistics and first weights are fetched under the current one's
//     last stages; memory requests are ordered for the wait counts hipcc derives (it does not see
//     the inline-asm LDS-DMA requests, and merges control flow pessimistically).
#include <hip/hip_runtime.h>
#include <type_traits>

#include "dbh_layout.h"

// (A/B switches of tools/ab_variants.sh; the defaults are what ships)
#ifndef DBH_EXP_A_DMA_UNDER_MFMA
#define DBH_EXP_A_DMA_UNDER_MFMA 1
#endif
// timing-only ablations of stage_b_chain (results wrong on purpose; tools/ab_variants.sh):
// 1 = no polls, 2 = no edge rows / posts, 4 = no LDS-DMA requests, 8 = no output / input
// transforms of conv2 -> conv3 -> conv4, 16 = no epilogue of conv4, 32 = no arrivals, 64 = no halo polls, 256 = no DPP row shifts (adds instead),
// 128 = no tile-word polls
#ifndef DBH_ABL
#define DBH_ABL 0
#endif

// This file is compiled twice by dbh_api.hip: as namespace dbh with DBH_TIMELINE 0 (the product)
// and as namespace dbh_timeline with DBH_TIMELINE 1 (cycle stamps for tools/timeline.py).  The
// stamps are global stores, and on gfx9 a store shares the vmcnt counter with the loads: one
// conditional store anywhere makes hipcc wait for vmcnt(0) at every later use of a prefetched
// register, so they must not even be compiled into the production kernel.
#ifndef DBH_FORWARD_NS
#define DBH_FORWARD_NS dbh
#endif
#ifndef DBH_TIMELINE
#define DBH_TIMELINE 0
#endif

// where stage F's requests are made inside the chain: 0 = in front of conv1d_16's MFMAs, 1 = in
// front of conv1d_15 (an experiment)
// the next group's FIRST window staged and its statistics made under stage F like the other three
// (1: +0.4 %), or carried in registers to that group's stage A (0: every wave summing and dividing
// behind that stage's first barrier)
#ifndef DBH_STATS0_IN_F
#define DBH_STATS0_IN_F 1
#endif
// stage D's operands of the earlier windows asked for behind conv7's mid-layer barrier (1: +0.13 %,
// the group's last conv7 12.9 k -> 12.0 k cycles) or behind its last MFMAs (0)
#ifndef DBH_Y_EARLY
#define DBH_Y_EARLY 1
#endif
#ifndef DBH_F_AHEAD_EARLY
#define DBH_F_AHEAD_EARLY 0
#endif
namespace DBH_FORWARD_NS {
using namespace dbh;

typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f2 __attribute__((ext_vector_type(2)));

constexpr int kWaves = 8;
constexpr int kThreads = kWaves * 64;

__device__ __forceinline__ f4 mfma4(float a, float b, f4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}

// ---------------------------------------------------------------------------------------------
// acc[m][t] += sum over (tap, sp, e) of A-tile(m) x B-tile(t).
//   a_lane: this lane's A address for tile 0, tap 0, sp 0:
//           region + (first_row + (lane&15)*row_step) * S + 2*(lane>>4)
//   b_lane: this lane's B address for tap 0, sp 0, tile 0:  weights + t0*128 + lane*2
//   MROWS : physical rows between consecutive M tiles (16 * conv stride)
//   SPTOT : C_in/8 of the whole layer (stride of the tap index in the weight image);
//           SP <= SPTOT is how many channel-pairs groups this call walks (split-K).
// All offsets are compile-time so every access is base + immediate.
// ---------------------------------------------------------------------------------------------
// Cycle stamps for the timeline mode (debug_stage 300): ts is non-null in timeline launches and
// points at ONE register of the wave whose lane `id` takes stamp `id` (v_writelane: no memory
// traffic - a global store per stamp made every vmcnt(0) of the kernel, the arrivals of stage B
// among them, wait for it); the register goes out as one coalesced store at the end of the window
// (flush_marks).  Stamps are the low 32 bits of the counters.
// The constant 100 MHz clock beside the shader clock of mark(): two of these per window give the
// shader clock's frequency while the kernel runs (tools/timeline.py) - it is below the 2.4 GHz the
// peak figures assume.
__device__ __forceinline__ void mark_realtime(unsigned* ts, int id) {
#if DBH_TIMELINE
    if (ts) {
        const unsigned now = (unsigned)__builtin_amdgcn_s_memrealtime();
        asm volatile("v_writelane_b32 %0, %1, %2" : "+v"(*ts) : "s"(now), "n"(id));
    }
#else
    (void)ts;
    (void)id;
#endif
}
__device__ __forceinline__ void mark(unsigned* ts, int id) {
#if DBH_TIMELINE
    if (ts) {
        const unsigned now = (unsigned)__builtin_readcyclecounter();
        asm volatile("v_writelane_b32 %0, %1, %2" : "+v"(*ts) : "s"(now), "n"(id));
    }
#else
    (void)ts;
    (void)id;
#endif
}
__device__ __forceinline__ void flush_marks(unsigned* ts, long long* out, int lane) {
#if DBH_TIMELINE
    if (ts) {
        // (a window's row is written in three goes - stages A-C, D, E-F: only what was stamped)
        if (*ts != 0u) out[lane] = (long long)*ts;
        *ts = 0u;
    }
#else
    (void)ts;
    (void)out;
    (void)lane;
#endif
}

// All-lanes reduction over a 16-lane row with DPP moves (no LDS round trips, unlike
// __shfl_xor -> ds_bpermute): xor 1, xor 2, half-row mirror, row mirror.
template <int CTRL>
__device__ __forceinline__ float dpp_move(float v) {
    return __builtin_bit_cast(
        float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, true));
}
__device__ __forceinline__ float row16_max(float v) {
    v = fmaxf(v, dpp_move<0xB1>(v));    // quad_perm [1,0,3,2]
    v = fmaxf(v, dpp_move<0x4E>(v));    // quad_perm [2,3,0,1]
    v = fmaxf(v, dpp_move<0x141>(v));   // row_half_mirror
    return fmaxf(v, dpp_move<0x140>(v));  // row_mirror
}
__device__ __forceinline__ float row16_sum(float v) {
    v += dpp_move<0xB1>(v);
    v += dpp_move<0x4E>(v);
    v += dpp_move<0x141>(v);
    return v + dpp_move<0x140>(v);
}
__device__ __forceinline__ float lane_value(float v, int src_lane) {
    return __builtin_bit_cast(float,
                              __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), src_lane));
}

// Compile-time step index handed to a layer's "side job": extra memory requests (next stage's
// weights) that are trickled out a few per MFMA step instead of as one burst - a burst fills the
// vector-memory queue and stalls the issuing wave in front of its own MFMAs.
template <int I>
struct IntC {
    static constexpr int value = I;
};
struct NoSide {
    template <int I>
    __device__ __forceinline__ void operator()(IntC<I>) const {}
};
// begin(): a layer's one-off requests (LDS-DMA for a later layer), issued behind the first step's
// fragment reads
struct NoBegin {
    __device__ __forceinline__ void operator()() const {}
};
struct NoHook {
    __device__ __forceinline__ void operator()() const {}
};
// Marks side work made of plain global loads to registers: a Winograd step issues those BETWEEN
// its MFMAs, one after every second MFMA, because a bunch of vector-memory instructions holds
// the wave (and its in-order MFMAs) ~35 cycles each, a lone one ~9 (tools/microbench/
// mfma_issue.hip).  LDS-DMA requests do not gain from this (M0 set-up) and stay up front.
template <class F>
struct Interleaved {
    F f;
    template <int I>
    __device__ __forceinline__ void operator()(IntC<I> t) const { f(t); }
};
template <class F>
__device__ __forceinline__ Interleaved<F> interleaved(F f) { return Interleaved<F>{f}; }
template <class T>
struct is_interleaved : std::false_type {};
template <class F>
struct is_interleaved<Interleaved<F>> : std::true_type {};

// Wave-wide sum of a non-negative 32-bit integer per lane (result < 2^31) with DPP row
// reductions + three readlanes instead of six rounds of ds_bpermute.
__device__ __forceinline__ int wave_sum_i32(int v) {
    v += __builtin_amdgcn_update_dpp(0, v, 0xB1, 0xF, 0xF, true);
    v += __builtin_amdgcn_update_dpp(0, v, 0x4E, 0xF, 0xF, true);
    v += __builtin_amdgcn_update_dpp(0, v, 0x141, 0xF, 0xF, true);
    v += __builtin_amdgcn_update_dpp(0, v, 0x140, 0xF, 0xF, true);
    return __builtin_amdgcn_readlane(v, 0) + __builtin_amdgcn_readlane(v, 16) +
           __builtin_amdgcn_readlane(v, 32) + __builtin_amdgcn_readlane(v, 48);
}

// The two waves of a SIMD share its matrix pipe, and the older one wins every tie: it runs ahead,
// finishes its share of a phase early, and the younger one then runs ALONE - at 57 % of the pipe's
// rate (a lone wave cannot keep the fp32 matrix pipe fed: tools/microbench/tile_loop.hip).  So a
// wave's priority falls as it gets further into the stretch between two barriers (STEP of STEPS):
// whoever is behind catches up and both reach the barrier together.  (+0.6 % on the whole kernel.)
template <int STEP, int STEPS>
__device__ __forceinline__ void progress_priority() {
    __builtin_amdgcn_s_setprio(3 - (4 * STEP) / STEPS);
}

// how many of the eight waves issue stage B's LDS-DMA requests (the low ones: they run ahead of
// their partners on the same SIMDs anyway, and a request is ~100 cycles of issue during which the
// partner has the matrix pipe to itself)
#ifndef DBH_DMA_WAVES
#define DBH_DMA_WAVES 4
#endif
// where in a layer's 18 steps the priority schedule of stage B starts over (the step whose
// priority is 3 again).  At that wrap the wave in front has priority 3 against the 0 of the one
// behind and pulls away, so it must not lie at the layer's first steps, where the halo rows of the
// neighbours are needed: there the schedule should take a normal step DOWN, which lets whoever is
// behind catch up.
#ifndef DBH_CATCHUP
#define DBH_CATCHUP 0
#endif
#ifndef DBH_PRIO_PHASE
#define DBH_PRIO_PHASE 12
#endif
template <int STEP, int STEPS>
__device__ __forceinline__ void progress_priority_pair(bool) {
    __builtin_amdgcn_s_setprio(3 - (4 * ((STEP + DBH_PRIO_PHASE) % STEPS)) / STEPS);
}

template <int MT, int NT>
struct Frags {
    f2 a[MT];
    f2 b[NT];
};

// LDS byte address of a pointer into the __shared__ arena (generic -> local address space).
__device__ __forceinline__ unsigned lds_addr(const float* p) {
    return (unsigned)(size_t)(const __attribute__((address_space(3))) float*)p;
}

// A workgroup barrier that orders LDS traffic only.  __syncthreads() is fence + s_barrier, and
// the fence drains EVERYTHING the wave has in flight - s_waitcnt vmcnt(0) - including the global
// prefetches this kernel deliberately keeps running across layers (the next window's samples,
// conv17's weight fragments, LDS-DMA pieces for a later layer): each such barrier then exposes
// an L2/HBM round trip.  Use this where the waves exchange data through LDS alone and nothing
// that was DMA'd since the last full barrier is read before the next one.
// VM = how many of the wave's most recent vector-memory requests may stay in flight (63 = all).
// The full barrier, spelled out: the LDS-DMA requests below are inline asm, which the compiler's
// own wait-count insertion does not see - a __syncthreads() would NOT wait for them.
__device__ __forceinline__ void full_barrier() {
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
}
template <int VM = 63>
__device__ __forceinline__ void lds_barrier() {
    static_assert(VM >= 0 && VM <= 63, "vmcnt is a 6-bit field");
    asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" ::"n"(VM) : "memory");
}

// A per-lane LDS pointer computed ONCE and kept in a register (made opaque as an integer, so that
// the compiler neither recomputes it at every use - a 64-bit multiply-add each - nor forgets that
// it points into LDS: through a generic pointer the stores would be flat_store).
typedef __attribute__((address_space(3))) float lds_float;
__device__ __forceinline__ lds_float* lds_pinned(const float* p) {
    unsigned a = lds_addr(p);
    asm volatile("" : "+v"(a));
    return (lds_float*)(size_t)a;
}

// One ds_read_b64 the compiler cannot see.  hipcc's own wait-count pass drains ALL outstanding LDS
// reads (lgkmcnt(0)) in front of every second MFMA group of this loop, exposing a full LDS
// round trip each time; issuing the reads from inline asm and counting them by hand
// (frag_wait) keeps the next step's fragments in flight under the current step's MFMAs.
template <int OFFSET_BYTES>
__device__ __forceinline__ f2 ds_read_f2(unsigned addr) {
    f2 v;
    asm volatile("ds_read_b64 %0, %1 offset:%2"
                 : "=v"(v)
                 : "v"(addr), "i"(OFFSET_BYTES)
                 : "memory");
    return v;
}

template <int OFFSET_BYTES>
__device__ __forceinline__ f4 ds_read_f4(unsigned addr) {
    f4 v;
    asm volatile("ds_read_b128 %0, %1 offset:%2"
                 : "=v"(v)
                 : "v"(addr), "i"(OFFSET_BYTES)
                 : "memory");
    return v;
}

template <int TAPS, int SP, int SPTOT, int MT, int NT, int NTTOT, int S, int MROWS, int IT>
__device__ __forceinline__ void load_frags(Frags<MT, NT>& f, unsigned a_addr, unsigned b_addr) {
    constexpr int tap = IT / SP, sp = IT % SP;
    static_assert(((MT - 1) * MROWS + TAPS - 1) * S * 4 + SP * 32 < 65536, "ds offset range");
    static_assert(((TAPS * SPTOT) * NTTOT + NT) * 512 < 65536, "ds offset range");
    if constexpr (MT > 0) f.a[0] = ds_read_f2<((0 * MROWS + tap) * S + sp * 8) * 4>(a_addr);
    if constexpr (MT > 1) f.a[1] = ds_read_f2<((1 * MROWS + tap) * S + sp * 8) * 4>(a_addr);
    if constexpr (MT > 2) f.a[2] = ds_read_f2<((2 * MROWS + tap) * S + sp * 8) * 4>(a_addr);
    if constexpr (MT > 3) f.a[3] = ds_read_f2<((3 * MROWS + tap) * S + sp * 8) * 4>(a_addr);
    static_assert(MT <= 4 && NT <= 3, "extend load_frags");
    if constexpr (NT > 0) f.b[0] = ds_read_f2<(((tap * SPTOT + sp) * NTTOT + 0) * 128) * 4>(b_addr);
    if constexpr (NT > 1) f.b[1] = ds_read_f2<(((tap * SPTOT + sp) * NTTOT + 1) * 128) * 4>(b_addr);
    if constexpr (NT > 2) f.b[2] = ds_read_f2<(((tap * SPTOT + sp) * NTTOT + 2) * 128) * 4>(b_addr);
}

// Wait until at most PENDING of this wave's LDS reads are outstanding, and make the fragment
// registers depend on the wait so no consumer can be scheduled above it.
template <int PENDING, int MT, int NT>
__device__ __forceinline__ void frag_wait(Frags<MT, NT>& f) {
    asm volatile("s_waitcnt lgkmcnt(%0)" : : "i"(PENDING) : "memory");
#pragma unroll
    for (int m = 0; m < MT; ++m) asm volatile("" : "+v"(f.a[m]));
#pragma unroll
    for (int t = 0; t < NT; ++t) asm volatile("" : "+v"(f.b[t]));
}

template <int TAPS, int SP, int SPTOT, int MT, int NT, int NTTOT, int S, int MROWS, int IT,
          class Side>
__device__ __forceinline__ void conv_step(unsigned a_addr, unsigned b_addr, Frags<MT, NT> (&buf)[2],
                                          f4 (&acc)[MT][NT], const Side& side) {
    constexpr int NIT = TAPS * SP;
    if constexpr (IT + 1 < NIT) {
        load_frags<TAPS, SP, SPTOT, MT, NT, NTTOT, S, MROWS, IT + 1>(buf[(IT + 1) & 1], a_addr,
                                                                     b_addr);
        side(IntC<IT>{});
        frag_wait<MT + NT>(buf[IT & 1]);     // the MT+NT reads just issued may stay in flight
    } else {
        side(IntC<IT>{});
        frag_wait<0>(buf[IT & 1]);
    }
    __builtin_amdgcn_sched_barrier(0);
    progress_priority<IT, NIT>();
    const Frags<MT, NT>& f = buf[IT & 1];
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int t = 0; t < NT; ++t) acc[m][t] = mfma4(f.a[m].x, f.b[t].x, acc[m][t]);
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int t = 0; t < NT; ++t) acc[m][t] = mfma4(f.a[m].y, f.b[t].y, acc[m][t]);
    // MFMA intrinsics are pure: without these pins hipcc sinks most of them to the end of the
    // layer and spills the fragments they keep alive
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int t = 0; t < NT; ++t) asm volatile("" : "+v"(acc[m][t]));
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (IT + 1 < NIT)
        conv_step<TAPS, SP, SPTOT, MT, NT, NTTOT, S, MROWS, IT + 1>(a_addr, b_addr, buf, acc, side);
}

// acc[m][t] += sum over (tap, sp, e) of A-tile(m) x B-tile(t), software-pipelined: the fragments
// of step it+1 are requested before the MFMAs of step it are issued.  a_lane / b_lane must
// point into LDS.
template <int TAPS, int SP, int SPTOT, int MT, int NT, int NTTOT, int S, int MROWS,
          class Side = NoSide>
__device__ __forceinline__ void conv_tiles(const float* a_lane, const float* b_lane,
                                           f4 (&acc)[MT][NT], const Side& side = Side()) {
    const unsigned a_addr = lds_addr(a_lane), b_addr = lds_addr(b_lane);
    Frags<MT, NT> buf[2];
    load_frags<TAPS, SP, SPTOT, MT, NT, NTTOT, S, MROWS, 0>(buf[0], a_addr, b_addr);
    conv_step<TAPS, SP, SPTOT, MT, NT, NTTOT, S, MROWS, 0>(a_addr, b_addr, buf, acc, side);
}

template <int MT, int NT>
__device__ __forceinline__ void zero_acc(f4 (&acc)[MT][NT]) {
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int t = 0; t < NT; ++t) acc[m][t] = f4{0.f, 0.f, 0.f, 0.f};
}

// ---------------------------------------------------------------------------------------------
// Fused epilogue.  An accumulator register r of lane l holds position 4*(l>>4)+r of the tile and
// output channel (l&15) of the N tile, so MaxPool2 pairs (r0,r1),(r2,r3) are in-register.
//   out_lane  : region + (first_out_row + ROWQ*(lane>>4)) * S_OUT + channel_base + (lane&15)
//               with ROWQ = 2 when pooling, 4 otherwise
//   bias_lane : packed + bias_offset + t0*16 + (lane&15); scale/shift likewise (BN channel!)
// Order per element: +bias, ReLU, [max over the position pair], [x*scale + shift]  —
// conv -> ReLU -> MaxPool -> BatchNorm exactly as network_architecture.py:34-40 orders them.
// ---------------------------------------------------------------------------------------------
// Per-lane epilogue constants, fetched from L2 BEFORE the layer's MFMA loop so their latency
// hides under it (a load issued after the barrier would stall every layer by an L2 round trip).
template <int NT, bool BN>
struct EpiParams {
    float b[NT], sc[NT], sh[NT];
    __device__ __forceinline__ void load(const float* __restrict__ bias_lane,
                                         const float* __restrict__ scale_lane,
                                         const float* __restrict__ shift_lane) {
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            b[t] = bias_lane[t * 16];
            sc[t] = BN ? scale_lane[t * 16] : 1.f;
            sh[t] = BN ? shift_lane[t * 16] : 0.f;
        }
    }
};

// Start the accumulators at the bias instead of zero: the MFMA chain then delivers conv + bias
// and the epilogue saves one VALU add per output (ADD_BIAS = false below).
// Where epilogue parameter OFF of the packed image lives: in the LDS copy (kParams) if it is
// part of it, else in global memory.
template <int OFF>
__device__ __forceinline__ const float* param_ptr(const float* lds, const float* packed) {
    if constexpr (OFF >= kTabBias0 && OFF < kTabBias1)
        return lds + kParams + (OFF - kTabBias0);
    else if constexpr (OFF >= kTabBn0 && OFF < kTabBn1)
        return lds + kParams + (kTabBias1 - kTabBias0) + (OFF - kTabBn0);
    else
        return packed + OFF;
}
// bias / BN scale / BN shift of layer CONV (BN index BNI, -1 = none) for channel n of tile 0
template <int CONV, int BNI, int NT>
__device__ __forceinline__ void load_epi(EpiParams<NT, (BNI >= 0)>& ep, const float* lds,
                                         const float* packed, int n) {
    constexpr int B = BNI >= 0 ? BNI : 0;
    ep.load(param_ptr<bias_offset(CONV)>(lds, packed) + n,
            param_ptr<bn_scale_offset(B)>(lds, packed) + n,
            param_ptr<bn_shift_offset(B)>(lds, packed) + n);
}

template <int MT, int NT, bool BN>
__device__ __forceinline__ void bias_acc(f4 (&acc)[MT][NT], const EpiParams<NT, BN>& ep) {
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int t = 0; t < NT; ++t) acc[m][t] = f4{ep.b[t], ep.b[t], ep.b[t], ep.b[t]};
}

template <int MT, int NT, int S_OUT, bool POOL, bool BN, bool ADD_BIAS = true>
__device__ __forceinline__ void epilogue(const f4 (&acc)[MT][NT], float* out_lane,
                                         const EpiParams<NT, BN>& ep) {
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        const float b = ADD_BIAS ? ep.b[t] : 0.f;
        const float sc = ep.sc[t], sh = ep.sh[t];
#pragma unroll
        for (int m = 0; m < MT; ++m) {
            float v0 = fmaxf(acc[m][t].x + b, 0.f);
            float v1 = fmaxf(acc[m][t].y + b, 0.f);
            float v2 = fmaxf(acc[m][t].z + b, 0.f);
            float v3 = fmaxf(acc[m][t].w + b, 0.f);
            if (POOL) {
                float o0 = fmaxf(v0, v1);
                float o1 = fmaxf(v2, v3);
                if (BN) {
                    o0 = fmaf(o0, sc, sh);
                    o1 = fmaf(o1, sc, sh);
                }
                out_lane[(m * 8 + 0) * S_OUT + t * 16] = o0;
                out_lane[(m * 8 + 1) * S_OUT + t * 16] = o1;
            } else {
                if (BN) {
                    v0 = fmaf(v0, sc, sh);
                    v1 = fmaf(v1, sc, sh);
                    v2 = fmaf(v2, sc, sh);
                    v3 = fmaf(v3, sc, sh);
                }
                out_lane[(m * 16 + 0) * S_OUT + t * 16] = v0;
                out_lane[(m * 16 + 1) * S_OUT + t * 16] = v1;
                out_lane[(m * 16 + 2) * S_OUT + t * 16] = v2;
                out_lane[(m * 16 + 3) * S_OUT + t * 16] = v3;
            }
        }
    }
}

// Weights of the NEXT layer are copied HBM/L2 -> LDS by the DMA path (global_load_lds_dwordx4:
// 64 lanes x 16 B = one 1 KiB piece per wave-instruction, destination = wave-uniform base +
// lane*16) while the current layer's MFMAs run; no VGPRs, no ds_write.  The __syncthreads()
// that ends the layer carries the vmcnt(0) that retires them.
// A read-only view of a global array for buffer loads: scalar base + per-lane 32-bit byte offset +
// scalar byte offset, all in the instruction - where a global_load needs a 64-bit vector add per
// address that its 13-bit immediate cannot reach.  (No bounds: the range covers any image.)
__device__ __forceinline__ __amdgpu_buffer_rsrc_t buffer_view(const float* base) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(base), 0, 0x7fffffff, 0x00020000);
}
__device__ __forceinline__ f2 buffer_load_f2(__amdgpu_buffer_rsrc_t view, unsigned lane_bytes,
                                             unsigned uniform_bytes) {
    // (two dword loads: this hipcc lowers the b64 / b128 forms of the builtin to ONE dword, splat)
    const unsigned lo = __builtin_amdgcn_raw_buffer_load_b32(view, (int)lane_bytes, (int)uniform_bytes, 0);
    const unsigned hi = __builtin_amdgcn_raw_buffer_load_b32(view, (int)lane_bytes + 4, (int)uniform_bytes, 0);
    return f2{__builtin_bit_cast(float, lo), __builtin_bit_cast(float, hi)};
}

// One 1 KiB piece: 64 lanes x 16 B from the wave-uniform global address `g_piece` (in scalar
// registers: the instruction's SADDR form, the lane's 16 lane bytes as its 32-bit offset - no
// 64-bit vector add per piece) to the wave-uniform LDS address in M0 + lane * 16.
// Inline asm: the compiler neither counts this request (see full_barrier) nor knows M0 changes -
// nothing else in this kernel uses M0.
__device__ __forceinline__ void dma_piece(const float* g_piece, const float* lds_piece,
                                          unsigned lane_bytes) {
    if (DBH_ABL & 4) return;
    const unsigned m0v = lds_addr(lds_piece);
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2"
                 :
                 : "s"(m0v), "v"(lane_bytes), "s"(g_piece)
                 : "memory");
}

// (DBH_DMA_ALL_WAVES: how many waves share a block copy - A/B knob; stage B's own requests have
// theirs, DBH_DMA_WAVES)
#ifndef DBH_DMA_ALL_WAVES
#define DBH_DMA_ALL_WAVES 4
#endif
#ifndef DBH_CONV8_DMA
#define DBH_CONV8_DMA 1
#endif
template <int NFLOATS, int NW = DBH_DMA_ALL_WAVES>
__device__ __forceinline__ void dma_weights(const float* __restrict__ g, float* lds_dst, int lane,
                                            int wave) {
    static_assert(NFLOATS % 256 == 0, "weight blocks are whole 1 KiB pieces");
    constexpr int kPieces = NFLOATS / 256;
    const unsigned lane_bytes = (unsigned)lane * 16u;
    if (wave >= NW) return;
#pragma unroll
    for (int i = 0; i < (kPieces + NW - 1) / NW; ++i) {
        const int piece = wave + NW * i;   // wave-uniform
        if (piece < kPieces) dma_piece(g + piece * 256, lds_dst + piece * 256, lane_bytes);
    }
}

// The i-th piece of this wave's share of the same copy (i < ceil(pieces / waves)): for callers
// that spread the requests between their own instructions.
template <int NFLOATS, int NW = kWaves>
__device__ __forceinline__ void dma_weights_one(const float* __restrict__ g, float* lds_dst,
                                                int lane, int wave, int i) {
    constexpr int kPieces = NFLOATS / 256;
    const int piece = wave + NW * i;
    if (wave < NW && piece < kPieces)
        dma_piece(g + piece * 256, lds_dst + piece * 256, (unsigned)lane * 16u);
}

// The same copy, spread over the NIT steps of the running layer (step IT issues its share).
template <int NFLOATS, int IT, int NIT, int NW = DBH_DMA_ALL_WAVES>
__device__ __forceinline__ void dma_weights_slice(const float* __restrict__ g, float* lds_dst,
                                                  int lane, int wave) {
    constexpr int kPieces = NFLOATS / 256;
    constexpr int kPerWave = (kPieces + NW - 1) / NW;
    const unsigned lane_bytes = (unsigned)lane * 16u;
    if (wave >= NW) return;
#pragma unroll
    for (int i = IT * kPerWave / NIT; i < (IT + 1) * kPerWave / NIT; ++i) {
        const int piece = wave + NW * i;
        if (piece < kPieces) dma_piece(g + piece * 256, lds_dst + piece * 256, lane_bytes);
    }
}

__device__ __forceinline__ void zero_row(float* region, int row, int stride, int channels,
                                         int tid) {
    if (tid < channels) region[row * stride + tid] = 0.f;
}

__device__ __forceinline__ void dump_stage(const float* region, int stride, int rows,
                                           int channels, float* __restrict__ out, int tid) {
    for (int idx = tid; idx < rows * channels; idx += kThreads) {
        const int r = idx / channels, c = idx - r * channels;
        out[idx] = region[(r + 1) * stride + c] * kActUnscale;      // (see kActScale)
    }
}

// a + b / a - b on both halves of a register pair.  (Written as asm because hipcc splits a plain
// f2 add into two scalar ones whenever it likes the register allocation better.)
__device__ __forceinline__ f2 pk_add(f2 a, f2 b) {
    f2 r;
    asm("v_pk_add_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ f2 pk_sub(f2 a, f2 b) {
    f2 r;
    asm("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
// ReLU for free: the `clamp` output modifier ([0, 1]) on the instruction that produces the value -
// activations are held times 2^-60 (dbh_layout.h: kActScale), so 1.0 is out of their reach.
__device__ __forceinline__ f2 pk_add_relu(f2 a, f2 b) {
    f2 r;
    asm("v_pk_add_f32 %0, %1, %2 clamp" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ f2 pk_sub_relu(f2 a, f2 b) {
    f2 r;
    asm("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1] clamp" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
// (k: a constant pair, from scalar registers)
__device__ __forceinline__ f2 pk_fma_relu(f2 k, f2 b, f2 c) {
    f2 r;
    asm("v_pk_fma_f32 %0, %1, %2, %3 clamp" : "=v"(r) : "s"(k), "v"(b), "v"(c));
    return r;
}
// MaxPool of two values the kernel made itself: v_max_f32 as it is.  (fmaxf costs three: hipcc
// canonicalises both operands first - `v_max_f32 x, x, x` - unless it can prove them quiet.)  Same
// result for everything but a signalling NaN.
__device__ __forceinline__ float max_raw(float a, float b) {
    float r;
    asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ f2 max_raw2(f2 a, f2 b) { return f2{max_raw(a.x, b.x), max_raw(a.y, b.y)}; }
// NB: the operands of these must come from instructions the compiler can see.  hipcc pads the
// distance between an MFMA and the first VALU instruction that reads its result with s_nop, but
// it does not look into inline asm: an accumulator read HERE straight after the MFMA chain (the
// exposed epilogue of a layer's last tile) arrives stale.

// ---------------------------------------------------------------------------------------------
// F(2,3) with 16 input channels out of an LDS image (conv13, conv15 of the inception block; conv6,
// 16 -> 48 at L = 256, ran this way until round 5 and is now part of stage_b_chain): 2/3 of the
// direct form's MFMAs.  One tile of 16 pairs per wave as in conv7, but with two channel groups a
// tile is only two steps of eight MFMAs, so the three N tiles run as ONE six-step pipeline: steps
// 0-1 build U (eight register pairs) and multiply for tile 0, steps 2-3 tile 1 with tile 0's
// outputs stored inside them, steps 4-5 tile 2 with tile 1's.  Input and output live in different
// buffers (IN_OFF at pitch kS16, OUT_OFF at pitch kS48): no barrier before the stores.
// Weights by N tile: [t][sp][matrix pair][lane][matrix of the pair][e].
// ---------------------------------------------------------------------------------------------
struct W23U16 {
    f2 u[4][2];      // [xi][sp]
};
struct W23Pipe16 {
    f2 rows[2][4];   // [step parity][input row]
    f4 b[2][2];      // [step parity][matrix pair]
};

// NSTEPS: two per N tile the wave works on (b_addr = the first of them).
template <int G, int NSTEPS, class Side>
__device__ __forceinline__ void w23c16_step(W23U16& U, unsigned a_addr, unsigned b_addr,
                                            W23Pipe16& pipe, f4 (&acc)[3][4], const float (&bias)[3],
                                            const Side& side) {
    constexpr int T = G / 2, SP = G % 2;
    auto loads = [&](auto step_tag) {
        constexpr int N = decltype(step_tag)::value;
        if constexpr (N < 2) {
            pipe.rows[N & 1][0] = ds_read_f2<(0 * kS16 + N * 8) * 4>(a_addr);
            pipe.rows[N & 1][1] = ds_read_f2<(1 * kS16 + N * 8) * 4>(a_addr);
            pipe.rows[N & 1][2] = ds_read_f2<(2 * kS16 + N * 8) * 4>(a_addr);
            pipe.rows[N & 1][3] = ds_read_f2<(3 * kS16 + N * 8) * 4>(a_addr);
        }
        pipe.b[N & 1][0] = ds_read_f4<((N * 2 + 0) * 256) * 4>(b_addr);
        pipe.b[N & 1][1] = ds_read_f4<((N * 2 + 1) * 256) * 4>(b_addr);
    };
    if constexpr (G == 0) loads(IntC<0>{});
    if constexpr (G + 1 < NSTEPS) {
        loads(IntC<G + 1>{});
        if constexpr (G + 1 < 2) asm volatile("s_waitcnt lgkmcnt(6)" ::: "memory");
        else asm volatile("s_waitcnt lgkmcnt(2)" ::: "memory");
    } else {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    f4(&b)[2] = pipe.b[G & 1];
#pragma unroll
    for (int p = 0; p < 2; ++p) asm volatile("" : "+v"(b[p]));
    progress_priority<G, NSTEPS>();
    if constexpr (T == 0) {
        f2(&d)[4] = pipe.rows[G & 1];
#pragma unroll
        for (int k = 0; k < 4; ++k) asm volatile("" : "+v"(d[k]));
        __builtin_amdgcn_sched_barrier(0);
        U.u[0][SP] = pk_sub(d[0], d[2]);
        U.u[1][SP] = pk_add(d[1], d[2]);
        U.u[2][SP] = pk_sub(d[2], d[1]);
        U.u[3][SP] = pk_sub(d[1], d[3]);
#pragma unroll
        for (int x = 0; x < 4; ++x) asm volatile("" : "+v"(U.u[x][SP]));
    }
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (SP == 0) {
        // M0 starts at +bias, M3 at -bias (even = M0+M1+M2, odd = M1-M2-M3 both get the bias)
        const f4 zero = f4{0.f, 0.f, 0.f, 0.f};
        const float bv = bias[T];
        acc[T][0] = mfma4(U.u[0][SP].x, b[0][0], f4{bv, bv, bv, bv});
        acc[T][1] = mfma4(U.u[1][SP].x, b[0][2], zero);
        acc[T][2] = mfma4(U.u[2][SP].x, b[1][0], zero);
        acc[T][3] = mfma4(U.u[3][SP].x, b[1][2], f4{-bv, -bv, -bv, -bv});
    } else {
#pragma unroll
        for (int p = 0; p < 2; ++p) {
            acc[T][2 * p] = mfma4(U.u[2 * p][SP].x, b[p][0], acc[T][2 * p]);
            acc[T][2 * p + 1] = mfma4(U.u[2 * p + 1][SP].x, b[p][2], acc[T][2 * p + 1]);
        }
    }
#pragma unroll
    for (int p = 0; p < 2; ++p) {
        acc[T][2 * p] = mfma4(U.u[2 * p][SP].y, b[p][1], acc[T][2 * p]);
        acc[T][2 * p + 1] = mfma4(U.u[2 * p + 1][SP].y, b[p][3], acc[T][2 * p + 1]);
    }
#pragma unroll
    for (int x = 0; x < 4; ++x) asm volatile("" : "+v"(acc[T][x]));
    __builtin_amdgcn_sched_barrier(0);
    side(IntC<G>{});
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (G + 1 < NSTEPS)
        w23c16_step<G + 1, NSTEPS>(U, a_addr, b_addr, pipe, acc, bias, side);
}

// ---------------------------------------------------------------------------------------------
// Winograd F(4,3) convolution (48 -> 48 channels, k = 3, 'same', stride 1) at L = 512, in place.
//   For the output quad (4j .. 4j+3) and d = x[4j-1 .. 4j+4]:
//     U0 = 4d0-5d2+d4        U1 = (d3+d4)-4(d1+d2)    U2 = (d4-d3)+4(d1-d2)
//     U3 = (d4-d2)+2(d3-d1)  U4 = (d4-d2)-2(d3-d1)    U5 = 4d1-5d3+d5        (VALU, registers)
//     M_xi = U_xi . V_xi over the 48 input channels                           (six GEMMs, MFMA)
//     y0 = M0+M1+M2+M3+M4   y1 = M1-M2+2(M3-M4)   y2 = M1+M2+4(M3+M4)   y3 = M1-M2+8(M3-M4)+M5
//   6 products per output quad instead of 12: HALF the MFMAs of the direct convolution.  The
//   extra fp32 round-off is ~2x the direct form's (end-to-end |dp| 1.7e-6 vs 0.8e-6 against the
//   fp64 oracle, tolerance 1e-4).
// One wave owns one tile of 16 quads (64 positions).  How a layer runs (timeline evidence in
// with the transform inside the MFMA loop and one epilogue per layer, all eight waves
// reached their epilogues together, the LDS store path serialised them for ~1.2k cycles and the
// matrix pipe idled ~3k cycles per layer):
//   1. U PHASE: the wave reads ALL its input rows (36 ds_read_b64) and keeps the six transformed
//      inputs of all 48 channels in registers (U[xi][sp]: 72 VGPRs).  After the barrier that
//      follows, nobody reads the activation buffer again during this layer - so the outputs may be
//      stored in place at any time, in any layout, with no hazard and no halo rows.
//   2. Three N TILES of 16 output channels, one after the other: 72 MFMAs each (six matrices x
//      twelve k-steps) whose inner loop is nothing but three ds_read_b128 of B fragments and
//      twelve MFMAs per step - no VALU beside the fp32 MFMAs, which share the vector ALUs.
//   3. The epilogue of tile t (output transform, ReLU, [pool, BN], stores) is cut in two pieces
//      and issued inside tile t+1's MFMA steps; only the last tile's epilogue is exposed.
// Weights: "third" t of a layer's image = everything N tile t needs ([sp][matrix pair][lane]
// [matrix of the pair][e]: one ds_read_b128 = the B fragments of two matrices for two k-steps),
// always in LDS slot t.  Slot t of the NEXT layer is requested once every wave has left tile t
// (split barrier below), one tile later, so that the wait is never a wait.
// ---------------------------------------------------------------------------------------------
struct W43U {
    f2 u[6][6];      // [xi][sp]: channels 8 sp + 2 (lane >> 4) + {0, 1} of quad pm(lane & 15)
};

template <int SP>
__device__ __forceinline__ void w43_load_rows(f2 (&d)[6], unsigned a_addr) {
    d[0] = ds_read_f2<(0 * kS48 + SP * 8) * 4>(a_addr);
    d[1] = ds_read_f2<(1 * kS48 + SP * 8) * 4>(a_addr);
    d[2] = ds_read_f2<(2 * kS48 + SP * 8) * 4>(a_addr);
    d[3] = ds_read_f2<(3 * kS48 + SP * 8) * 4>(a_addr);
    d[4] = ds_read_f2<(4 * kS48 + SP * 8) * 4>(a_addr);
    d[5] = ds_read_f2<(5 * kS48 + SP * 8) * 4>(a_addr);
}

// U[.][SP] from the six input rows of channel group SP: both components of a fragment at once
// (v_pk_fma_f32 / v_pk_add_f32), as ONE block of VALU ahead of the step's MFMAs - beside fp32
// MFMAs (same ALUs) a VALU instruction costs ~4 cycles in a block, ~12 on its own
// (tools/microbench/mfma_issue.hip).  Shared sub-expressions:
//   a = d4-4d2, b = d3-4d1: U1 = a+b, U2 = a-b;   c = d4-d2, g = d3-d1: U3 = c+2g, U4 = c-2g
template <int SP>
__device__ __forceinline__ void w43_transform(W43U& U, const f2 (&d)[6]) {
#ifdef DBH_EXP_SCALAR_TRANSFORM
    // A/B knob (tools/ab_variants.sh): the same arithmetic one component at a time - twice the
    // instructions, none of them packed (MI355X_MICROARCH.md lists packed fp32 VALU beside MFMAs
    // as an anti-lever; measured in THIS kernel: see DESIGN.md section 4)
#pragma unroll
    for (int e = 0; e < 2; ++e) {
        const float d0 = d[0][e], d1 = d[1][e], d2 = d[2][e], d3 = d[3][e], d4 = d[4][e], d5 = d[5][e];
        const float a = fmaf(-4.f, d2, d4), b = fmaf(-4.f, d1, d3);
        const float c = d4 - d2, g = d3 - d1;
        U.u[0][SP][e] = fmaf(4.f, d0, fmaf(-5.f, d2, d4));
        U.u[1][SP][e] = a + b;
        U.u[2][SP][e] = a - b;
        U.u[3][SP][e] = fmaf(2.f, g, c);
        U.u[4][SP][e] = fmaf(-2.f, g, c);
        U.u[5][SP][e] = fmaf(4.f, d1, fmaf(-5.f, d3, d5));
    }
#pragma unroll
    for (int x = 0; x < 6; ++x) asm volatile("" : "+v"(U.u[x][SP]));
    return;
#endif
    const f2 m4 = f2{-4.f, -4.f}, p2 = f2{2.f, 2.f}, m2 = f2{-2.f, -2.f};
    const f2 p4 = f2{4.f, 4.f}, m5 = f2{-5.f, -5.f};
    const f2 a = __builtin_elementwise_fma(m4, d[2], d[4]), b = __builtin_elementwise_fma(m4, d[1], d[3]);
    const f2 c = d[4] - d[2], g = d[3] - d[1];
    U.u[0][SP] = __builtin_elementwise_fma(p4, d[0], __builtin_elementwise_fma(m5, d[2], d[4]));
    U.u[1][SP] = a + b;
    U.u[2][SP] = a - b;
    U.u[3][SP] = __builtin_elementwise_fma(p2, g, c);
    U.u[4][SP] = __builtin_elementwise_fma(m2, g, c);
    U.u[5][SP] = __builtin_elementwise_fma(p4, d[1], __builtin_elementwise_fma(m5, d[3], d[5]));
#pragma unroll
    for (int x = 0; x < 6; ++x) asm volatile("" : "+v"(U.u[x][SP]));
}

// the B fragments of one step: [matrix pair p][lane] -> {V(2p).e0, V(2p).e1, V(2p+1).e0, .e1}
template <int SP>
__device__ __forceinline__ void w43_load_b(f4 (&b)[3], unsigned b_addr) {
    b[0] = ds_read_f4<((SP * 3 + 0) * 256) * 4>(b_addr);
    b[1] = ds_read_f4<((SP * 3 + 1) * 256) * 4>(b_addr);
    b[2] = ds_read_f4<((SP * 3 + 2) * 256) * 4>(b_addr);
}

// Half (h = rows 2h, 2h+1 of every accumulator = quads 2q + 8h, 2q + 1 + 8h of the wave's tile)
// of the epilogue of N tile T: output transform, ReLU (+ MaxPool2 + BatchNorm), stores in place.
// MFMA row m = 4q'+r' of the wave's tile works on quad pm(m) = 2q' + (r'&1) + 8(r'>>1), so that the
// four lane groups of one store write quads 2 apart = 16-bank-aligned quarters of the LDS banks.
template <int T, bool POOL, bool BN>
__device__ __forceinline__ void w43_epilogue_half(const f4 (&acc)[6], int h, float sc, float sh,
                                                  lds_float* out_q) {
    constexpr int NV = POOL ? 2 : 4;
    const f2 k2 = f2{2.f, 2.f}, k4 = f2{4.f, 4.f}, k8 = f2{8.f, 8.f};
    const f2 a0 = f2{acc[0][2 * h], acc[0][2 * h + 1]};
    const f2 a1 = f2{acc[1][2 * h], acc[1][2 * h + 1]};
    const f2 a2 = f2{acc[2][2 * h], acc[2][2 * h + 1]};
    const f2 a3 = f2{acc[3][2 * h], acc[3][2 * h + 1]};
    const f2 a4 = f2{acc[4][2 * h], acc[4][2 * h + 1]};
    const f2 a5 = f2{acc[5][2 * h], acc[5][2 * h + 1]};
    const f2 s12 = a1 + a2, d12 = a1 - a2, s34 = a3 + a4, d34 = a3 - a4;
    // (ReLU = the clamp modifier of each output's last instruction)
    const f2 y0 = pk_add_relu(a0 + s12, s34);
    const f2 y1 = pk_fma_relu(k2, d34, d12);
    const f2 y2 = pk_fma_relu(k4, s34, s12);
    const f2 y3 = pk_fma_relu(k8, d34, d12 + a5);      // (a5 through a visible add: see pk_fma_relu)
#pragma unroll
    for (int e = 0; e < 2; ++e) {
        float v0 = y0[e], v1 = y1[e], v2 = y2[e], v3 = y3[e];
        // quad j = wave*16 + 2q + e + 8h = pm(4q + 2h + e); out_q = this lane's place in quad
        // wave*16 + 2q (one address per layer; the rest are immediate offsets of the stores)
        lds_float* dst = out_q + NV * (e + 8 * h) * kS48 + T * 16;
        if constexpr (POOL) {
            f2 p = f2{fmaxf(v0, v1), fmaxf(v2, v3)};
            if (BN) p = __builtin_elementwise_fma(p, f2{sc, sc}, f2{sh, sh});
            dst[0] = p.x;
            dst[kS48] = p.y;
        } else {
            if (BN) {
                v0 = fmaf(v0, sc, sh);
                v1 = fmaf(v1, sc, sh);
                v2 = fmaf(v2, sc, sh);
                v3 = fmaf(v3, sc, sh);
            }
            dst[0] = v0;
            dst[kS48] = v1;
            dst[2 * kS48] = v2;
            dst[3 * kS48] = v3;
        }
    }
}

// The same epilogue (MaxPool2 + BatchNorm) into conv1d_7's PARK in global memory (round 6): the
// layout stage D's chain loads its operands from - [half of the window][row i of the quad][channel
// group g][lane (q, n) of the reading wave][r], channels 16g + 4q + r of position 4 (16 half + n) + i
// (dbh_layout.h: kPark7Floats).  park_lane: this lane's place for pooled row 0 of quad 2q of its
// tile, channel n of N tile 0 (w43_nsplit_half); pooled row i = 2e + pp of quad offset e + 8h lies
// i * 768 + h * 16 floats on, N tile T another T * 256.
template <int T, class Ptr>
__device__ __forceinline__ void w43_epilogue_half_park(const f4 (&acc)[6], int h, float sc, float sh,
                                                       Ptr park_lane) {
    const f2 k2 = f2{2.f, 2.f}, k4 = f2{4.f, 4.f}, k8 = f2{8.f, 8.f};
    const f2 a0 = f2{acc[0][2 * h], acc[0][2 * h + 1]};
    const f2 a1 = f2{acc[1][2 * h], acc[1][2 * h + 1]};
    const f2 a2 = f2{acc[2][2 * h], acc[2][2 * h + 1]};
    const f2 a3 = f2{acc[3][2 * h], acc[3][2 * h + 1]};
    const f2 a4 = f2{acc[4][2 * h], acc[4][2 * h + 1]};
    const f2 a5 = f2{acc[5][2 * h], acc[5][2 * h + 1]};
    const f2 s12 = a1 + a2, d12 = a1 - a2, s34 = a3 + a4, d34 = a3 - a4;
    const f2 y0 = pk_add_relu(a0 + s12, s34);
    const f2 y1 = pk_fma_relu(k2, d34, d12);
    const f2 y2 = pk_fma_relu(k4, s34, s12);
    const f2 y3 = pk_fma_relu(k8, d34, d12 + a5);      // (a5 through a visible add: see pk_fma_relu)
#pragma unroll
    for (int e = 0; e < 2; ++e) {
        f2 p = f2{fmaxf(y0[e], y1[e]), fmaxf(y2[e], y3[e])};
        p = __builtin_elementwise_fma(p, f2{sc, sc}, f2{sh, sh});
        Ptr dst = park_lane + (2 * e) * 768 + T * 256 + h * 16;
        dst[0] = p.x;
        dst[768] = p.y;
    }
}

// The twelve MFMAs of one step of a stage-B tile, TRANSPOSED (round 5): the weight fragments are
// the A operand (M = 16 output channels), the transformed inputs the B operand (N = the wave's 16
// quads), so that lane (quad n, q) ends up with output channels 16t + 4q + r of ITS OWN quad in
// registers r = 0..3 of each of the six accumulators - and everything that follows (output
// transform, ReLU, the next layer's input transform) is in-lane.  Step 0 starts the chains: five
// from the MFMA's constant 0, M1's from the bias of the lane's four channels (every output of the
// transform takes M1 with weight 1).
template <int SP>
__device__ __forceinline__ void w43t_mfmas(const W43U& U, const f4 (&b)[3], f4 (&acc)[6], f4 bias4) {
    if constexpr (SP == 0) {
        const f4 zero = f4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int p = 0; p < 3; ++p) {
            acc[2 * p] = mfma4(b[p][0], U.u[2 * p][SP].x, zero);
            acc[2 * p + 1] = mfma4(b[p][2], U.u[2 * p + 1][SP].x, p == 0 ? bias4 : zero);
        }
    } else {
#pragma unroll
        for (int p = 0; p < 3; ++p) {
            acc[2 * p] = mfma4(b[p][0], U.u[2 * p][SP].x, acc[2 * p]);
            acc[2 * p + 1] = mfma4(b[p][2], U.u[2 * p + 1][SP].x, acc[2 * p + 1]);
        }
    }
#pragma unroll
    for (int p = 0; p < 3; ++p) {
        acc[2 * p] = mfma4(b[p][1], U.u[2 * p][SP].y, acc[2 * p]);
        acc[2 * p + 1] = mfma4(b[p][3], U.u[2 * p + 1][SP].y, acc[2 * p + 1]);
    }
#pragma unroll
    for (int x = 0; x < 6; ++x) asm volatile("" : "+v"(acc[x]));
}

// ---------------------------------------------------------------------------------------------
// STAGE A INSIDE conv1d_2's TILE 0.  conv1d_1 (k = 3, stride 2, one input channel) + ReLU + BN1
// used to be a stage of its own: twelve MFMAs per wave, then 98 KB of outputs through the LDS store
// path (~85 B/clk), a barrier, and tile 0 of conv1d_2 reading them back - 4.7k cycles per window
// for 0.8k cycles of matrix work (profiles/r03_v1/timeline_5120_fused.txt).  Here the wave that
// owns quads j = 16 wave + pm(n) computes the six conv1d_1 rows each of its quads needs
// (positions 4j - 1 .. 4j + 4) itself, TRANSPOSED: M = 16 output channels, N = 16 quads,
// K = 3 taps (+ a zero), one MFMA per (row t, channel group g):
//     A[m][k] = w[k][16g + m]        (lane (m, k): one register per channel group, per workgroup)
//     B[k][n] = x[8 j(n) - 2 + 2t + k]     (lane (n, k): six samples per window)
//     D: lane (n, q) gets channels 16g + 4q + r of row t of ITS OWN quad in registers r = 0..3
// - which is where conv1d_2's input transform wants them: ReLU (clamp), BN1 and the F(4,3)
// transform run on registers, and the result is conv1d_2's A fragment as it stands, because the
// packer stores conv1d_2's matrices with k-step 4g + r <-> channels {16g + 4q + r} (dbh_layout.h:
// frag_cin).  18 MFMAs per wave instead of 12 (every quad's two halo rows are computed twice), no
// LDS traffic at all, one barrier less.  The rows that are conv1d_2's zero padding (position -1 of
// quad 0, position 512 of quad 127) are set to zero, not to BN1(ReLU(bias)).
// ---------------------------------------------------------------------------------------------
struct ConvAIn {
    float xs[6];          // x[8j - 2 + 2t + q], t = 0..5: normalised, times kActScale; 0 outside
    float w[3];           // w[q][16g + n] of conv1d_1 (0 for q = 3)
    // conv1d_2's 'same' padding: row -1 of quad 0 and row 512 of quad 127 are zeros, not
    // BN1(ReLU(bias)).  Row 0 of a quad enters the input transform in U0 = 4 d0 - 5 d2 + d4 alone,
    // row 5 in U5 = 4 d1 - 5 d3 + d5 alone: the first quad's lanes multiply d0 by 0 instead of 4
    // (the constant of an fma that is there anyway), the last quad's d5 by 0 instead of 1.
    f2 p4_edge;           // {4, 4}; {0, 0} in the lanes of quad 0
    f2 one_edge;          // {1, 1}; {0, 0} in the lanes of quad 127
    float* dump;          // debug_stage 0: where this lane's 4 x 12 outputs go
    bool dump_on;         // (wave-uniform)
    bool stop;            // debug_stage 0 / 100: the kernel ends behind tile 0
    bool wave_hi;         // wave >= 4 (progress_priority_pair)
    bool dump_b;          // debug_stage 1: stage B's output is dumped (from registers)
};

// ReLU and BN1 of the six rows of one channel pair: x * 1 with the clamp modifier (activations are
// held times 2^-60: dbh_layout.h), then x * scale + shift - twelve packed instructions in one
// block.  The inputs are results of MFMAs, and hipcc does not pad the distance between an MFMA
// and an inline-asm reader of its result (see pk_fma_relu): the block starts with the wait states
// itself (the callers also keep twelve MFMAs between a conv1d_1 product and this block).
__device__ __forceinline__ void relu_bn_rows(f2 (&d)[6], const f2 (&v)[6], f2 sc, f2 sh) {
    // (s_nop 7, s_nop 2: the eleven wait states between an 8-pass MFMA and a vector instruction
    // that reads its result, which hipcc would insert itself if it could see into this block -
    // correctness does not rest on how far away the callers keep the conv1d_1 MFMAs.  66 cycles
    // per window.)
    asm volatile(
        "s_nop 7\n\ts_nop 2\n\t"
        "v_pk_mul_f32 %0, %6, 1.0 op_sel_hi:[1,0] clamp\n\t"
        "v_pk_mul_f32 %1, %7, 1.0 op_sel_hi:[1,0] clamp\n\t"
        "v_pk_mul_f32 %2, %8, 1.0 op_sel_hi:[1,0] clamp\n\t"
        "v_pk_mul_f32 %3, %9, 1.0 op_sel_hi:[1,0] clamp\n\t"
        "v_pk_mul_f32 %4, %10, 1.0 op_sel_hi:[1,0] clamp\n\t"
        "v_pk_mul_f32 %5, %11, 1.0 op_sel_hi:[1,0] clamp\n\t"
        "v_pk_fma_f32 %0, %0, %12, %13\n\t"
        "v_pk_fma_f32 %1, %1, %12, %13\n\t"
        "v_pk_fma_f32 %2, %2, %12, %13\n\t"
        "v_pk_fma_f32 %3, %3, %12, %13\n\t"
        "v_pk_fma_f32 %4, %4, %12, %13\n\t"
        "v_pk_fma_f32 %5, %5, %12, %13"
        : "=&v"(d[0]), "=&v"(d[1]), "=&v"(d[2]), "=&v"(d[3]), "=&v"(d[4]), "=&v"(d[5])
        : "v"(v[0]), "v"(v[1]), "v"(v[2]), "v"(v[3]), "v"(v[4]), "v"(v[5]), "v"(sc), "v"(sh));
}

// bias of conv1d_1 for this lane's four channels of group G, BN1 scale / shift pairs of step SP:
// byte offsets from this lane's place in the LDS parameter table (kParams + 4q)
constexpr int kTabBn1ScaleBytes = ((kTabBias1 - kTabBias0) + (bn_scale_offset(0) - kTabBn0)) * 4;
constexpr int kTabBn1ShiftBytes = ((kTabBias1 - kTabBias0) + (bn_shift_offset(0) - kTabBn0)) * 4;
static_assert(bias_offset(0) == kTabBias0, "conv1d_1's bias opens the parameter table");

template <int SP>
__device__ __forceinline__ void w43a_load_params(f2 (&par)[2], unsigned p_addr) {
    constexpr int ch = 16 * (SP >> 1) + 2 * (SP & 1);
    par[0] = ds_read_f2<kTabBn1ScaleBytes + ch * 4>(p_addr);
    par[1] = ds_read_f2<kTabBn1ShiftBytes + ch * 4>(p_addr);
}

// the six conv1d_1 products of channel group G (their accumulators start at the bias)
template <int G>
__device__ __forceinline__ void conv_a_group(const ConvAIn& in, f4 bias4, f4 (&a1)[6]) {
#pragma unroll
    for (int t = 0; t < 6; ++t) a1[t] = mfma4(in.w[G], in.xs[t], bias4);
}

// Step SP of tile 0: channel pair H = SP & 1 of group G = SP >> 1 - six rows of conv1d_1 outputs
// become U[.][SP] (24 packed + 4 plain vector instructions in one block), then the step's twelve
// conv1d_2 MFMAs.  Group 2's conv1d_1 MFMAs ride in front of step 2's.
template <int SP, class Side>
__device__ __forceinline__ void w43a_step(W43U& U, const ConvAIn& in, f4 (&a1)[3][6], f4 bias4_g2,
                                          unsigned p_addr, unsigned b_addr, f2 (&pbuf)[2][2],
                                          f4 (&buf)[2][3], f4 (&acc)[6], f4 bias,
                                          const Side& side) {
    constexpr int G = SP >> 1, H = SP & 1;
    if constexpr (SP + 1 < 6) {
        w43a_load_params<SP + 1>(pbuf[(SP + 1) & 1], p_addr);
        w43_load_b<SP + 1>(buf[(SP + 1) & 1], b_addr);
        asm volatile("s_waitcnt lgkmcnt(5)" ::: "memory");
    } else {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    f4(&b)[3] = buf[SP & 1];
    f2(&par)[2] = pbuf[SP & 1];
#pragma unroll
    for (int p = 0; p < 3; ++p) asm volatile("" : "+v"(b[p]));
    asm volatile("" : "+v"(par[0]), "+v"(par[1]));
    progress_priority_pair<SP, 18>(in.wave_hi);
    __builtin_amdgcn_sched_barrier(0);
    f2 d[6], v[6];
#pragma unroll
    for (int t = 0; t < 6; ++t)
        v[t] = H ? f2{a1[G][t].z, a1[G][t].w} : f2{a1[G][t].x, a1[G][t].y};
    relu_bn_rows(d, v, par[0], par[1]);
    if (in.dump_on) {                     // debug_stage 0: the quad's own four rows
#pragma unroll
        for (int t = 1; t < 5; ++t) {
            in.dump[(t - 1) * 48 + 16 * G + 2 * H] = d[t].x * kActUnscale;
            in.dump[(t - 1) * 48 + 16 * G + 2 * H + 1] = d[t].y * kActUnscale;
        }
    }
    {   // the F(4,3) input transform (w43_transform) with the two padding rows masked
        const f2 m4 = f2{-4.f, -4.f}, p2 = f2{2.f, 2.f}, m2 = f2{-2.f, -2.f};
        const f2 p4 = f2{4.f, 4.f}, m5 = f2{-5.f, -5.f};
        const f2 d5 = d[5] * in.one_edge;
        const f2 a = __builtin_elementwise_fma(m4, d[2], d[4]), b2 = __builtin_elementwise_fma(m4, d[1], d[3]);
        const f2 c = d[4] - d[2], g = d[3] - d[1];
        U.u[0][SP] = __builtin_elementwise_fma(in.p4_edge, d[0], __builtin_elementwise_fma(m5, d[2], d[4]));
        U.u[1][SP] = a + b2;
        U.u[2][SP] = a - b2;
        U.u[3][SP] = __builtin_elementwise_fma(p2, g, c);
        U.u[4][SP] = __builtin_elementwise_fma(m2, g, c);
        U.u[5][SP] = __builtin_elementwise_fma(p4, d[1], __builtin_elementwise_fma(m5, d[3], d5));
#pragma unroll
        for (int x = 0; x < 6; ++x) asm volatile("" : "+v"(U.u[x][SP]));
    }
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (SP == 2) {
        conv_a_group<2>(in, bias4_g2, a1[2]);
#pragma unroll
        for (int t = 0; t < 6; ++t) asm volatile("" : "+v"(a1[2][t]));
    }
    w43t_mfmas<SP>(U, b, acc, bias);
    __builtin_amdgcn_sched_barrier(0);
    side(IntC<SP>{});
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (SP + 1 < 6)
        w43a_step<SP + 1>(U, in, a1, bias4_g2, p_addr, b_addr, pbuf, buf, acc, bias, side);
}

// Tile 0 of conv1d_2 with conv1d_1 inside.  between(i): the caller's i-th request of the twelve it
// may place between the first conv1d_1 MFMAs (LDS-DMA pieces for slots 1 and 2: a request costs
// ~100 cycles of issue, a bunch of them in front of the MFMAs keeps the matrix pipe idle).
template <class Between, class Side>
__device__ __forceinline__ void w43a_tile0(W43U& U, const ConvAIn& in, float* lds, const float* third0,
                                           int lane, f4 (&acc)[6], f4 bias, const Between& between,
                                           const Side& side) {
    const int q = lane >> 4;
    const f4* bias4 = reinterpret_cast<const f4*>(lds + kParams + 4 * q);   // + 4 g: group g
    const f4 b0 = bias4[0], b1 = bias4[4], b2 = bias4[8];
    f4 a1[3][6];
#pragma unroll
    for (int t = 0; t < 6; ++t) {
        a1[0][t] = mfma4(in.w[0], in.xs[t], b0);
        between(t);
    }
#pragma unroll
    for (int t = 0; t < 6; ++t) {
        a1[1][t] = mfma4(in.w[1], in.xs[t], b1);
        between(6 + t);
    }
#pragma unroll
    for (int g = 0; g < 2; ++g)
#pragma unroll
        for (int t = 0; t < 6; ++t) asm volatile("" : "+v"(a1[g][t]));
    __builtin_amdgcn_sched_barrier(0);
    const unsigned p_addr = lds_addr(lds + kParams + 4 * q);
    const unsigned b_addr = lds_addr(third0 + lane * 4);     // (the tile's third of conv2's weights)
    f2 pbuf[2][2];
    f4 buf[2][3];
    w43a_load_params<0>(pbuf[0], p_addr);
    w43_load_b<0>(buf[0], b_addr);
    w43a_step<0>(U, in, a1, b2, p_addr, b_addr, pbuf, buf, acc, bias, side);
}

// =============================================================================================
// STAGE B CHAINED IN REGISTERS (round 5): conv1d_2 -> conv1d_3 -> conv1d_4 without an LDS image
// of any activation in between.  All three layers run TRANSPOSED (w43t_mfmas): lane (n, q) of wave
// w owns quad j = 16 w + n and, after the six steps of N tile t, holds output channels 16t + 4q + r
// of that quad's six Winograd products.  The output transform (in-lane), bias (the start of M1's
// chain), ReLU (clamp) leave Y[t][h][i] = positions 4j + i, channels 16t + 4q + 2h + {0, 1} - and
// the next layer's input transform needs exactly those channels in this lane (its k-step 4t + r
// contracts over channels {16t + 4q + r}: dbh_layout.h: frag_cin) plus ONE halo position each
// side: position 4j - 1 is lane n - 1's row 3, position 4j + 4 lane n + 1's row 0 - a DPP row
// shift; at the two ends of a wave's row of 16 quads they come from the neighbouring wave through
// 2 x 48 floats of LDS per wave and layer, announced on a counter word per wave (halo_post /
// halo_wait: no workgroup barrier).  What this removes per layer: 98 KB of LDS stores, the 36
// A-fragment reads per wave that brought them back, both workgroup barriers and the pipeline
// fill behind each.  Nine tiles of six steps run back to back; the epilogue of tile T (output
// transform -> Y, edge rows, post) rides inside tile T + 1's steps, U of the next layer is built
// step by step inside its tile 0 (from Y of channel group g in steps 2g, 2g + 1), conv1d_4's
// epilogue pools, applies BN2 and stores [position][channel] rows for stage C.
//   Weights: a ring of six slots - conv2's thirds in slots 0..2 of the weight area, conv3's in the
// (idle) activation buffer, conv4's in slots 0..2 again as conv2 leaves them, conv7's behind
// conv4's.  "Every wave has left tile t" / "every wave's pieces have landed" is a count of
// arrivals on one word per tile index (chain_arrive: after the wave's own requests have landed),
// polled where the answer is needed, always at least a tile after it became true for a wave in
// step with the others.
// =============================================================================================
typedef __attribute__((address_space(3))) f2 lds_f2;
// pieces of a third (18) per wave: waves 0 and 1 take three, the others two
constexpr int kThirdPieces = (kWinoHalf / 256 + kWaves - 1) / kWaves;
static_assert(kThirdPieces == 3, "");

// (no lane mask on the adds and edge stores below: changing EXEC right behind a block of MFMAs
// cost 0.7 % of the kernel - profiles/r05_ablation.txt - so every lane takes part, the lanes that
// have nothing to say with an address of their own in a 1 KB scratch, kChainDummy)
// arrive_addr: lane 0 -> word 0 of kSyncTiles, the others -> scratch (chain_addresses)
__device__ __forceinline__ void chain_arrive(unsigned arrive_addr, int t) {
    if (DBH_ABL & 32) return;
    // release: this wave's LDS reads are done and its LDS-DMA pieces have landed
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\tds_add_u32 %0, %1 offset:%2" ::"v"(arrive_addr), "v"(1u),
                 "n"(t * 4)
                 : "memory");
}
template <int BASE = kSyncTiles>
__device__ __forceinline__ void chain_wait(float* lds, int t, unsigned target) {
    if (DBH_ABL & (1 | 32 | 128)) return;
    const unsigned addr = lds_addr(lds + BASE + t);
    for (;;) {
        unsigned seen;
        asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(seen) : "v"(addr) : "memory");
        if ((int)(__builtin_amdgcn_readfirstlane(seen) - target) >= 0) break;
        __builtin_amdgcn_s_sleep(1);
    }
}
// The polls above cost an LDS round trip each with nothing else issued by the wave; where the
// answer is almost always "yes" the word is PEEKED a step ahead - the read rides in front of a
// step's fragment requests, whose hand-counted wait retires it - and only looked at here.
template <int BASE = kSyncTiles>
__device__ __forceinline__ unsigned chain_peek(float* lds, int t) {
    unsigned seen;
    asm volatile("ds_read_b32 %0, %1" : "=v"(seen) : "v"(lds_addr(lds + BASE + t)) : "memory");
    return seen;
}
template <int BASE = kSyncTiles>
__device__ __forceinline__ void chain_check(float* lds, int t, unsigned peeked, unsigned target) {
    if (DBH_ABL & (1 | 32 | 128)) return;
    asm volatile("" : "+v"(peeked));     // (the use stays behind the wait that retired the read)
    if ((int)(__builtin_amdgcn_readfirstlane(peeked) - target) >= 0) return;
    chain_wait<BASE>(lds, t, target);
}
// Halo posts: wave w owns the word pair kSyncHalo + 2w = {posts of wave w - 1, posts of wave w + 1}
// (an outer wave stands in for its missing neighbour itself), so that one 8-byte read answers
// "have both my neighbours stored the edge rows of tile g".  post_addr: lane 0 -> the right
// neighbour's first word, lane 1 -> the left neighbour's second (halo_post_address).
// (plain LDS instructions: a wave's requests are served in order, so the adds land behind the edge
// rows stored before them)
__device__ __forceinline__ unsigned halo_post_address(float* lds, int wave, int lane) {
    const int right = wave < kWaves - 1 ? 2 * (wave + 1) : 2 * wave + 1;
    const int left = wave > 0 ? 2 * (wave - 1) + 1 : 0;
    return lds_addr(lane < 2 ? lds + kSyncHalo + (lane == 0 ? right : left)
                             : lds + kChainDummy + 176 + lane);
}
__device__ __forceinline__ void halo_post(unsigned post_addr) {
    asm volatile("ds_add_u32 %0, %1" ::"v"(post_addr), "v"(1u) : "memory");
}
typedef unsigned u2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ bool halo_ready(u2 seen, unsigned target) {
    const int da = (int)(__builtin_amdgcn_readfirstlane(seen.x) - target);
    const int db = (int)(__builtin_amdgcn_readfirstlane(seen.y) - target);
    return da >= 0 && db >= 0;
}
template <int BASE = kSyncHalo>
__device__ __forceinline__ void halo_wait(float* lds, int wave, unsigned target) {
    if (DBH_ABL & (1 | 2 | 8 | 64)) return;
    const unsigned addr = lds_addr(lds + BASE + 2 * wave);
    for (;;) {
        u2 seen;
        asm volatile("ds_read_b64 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(seen) : "v"(addr) : "memory");
        if (halo_ready(seen, target)) break;
        __builtin_amdgcn_s_sleep(1);
    }
}
template <int BASE = kSyncHalo>
__device__ __forceinline__ u2 halo_peek(float* lds, int wave) {
    u2 seen;
    asm volatile("ds_read_b64 %0, %1" : "=v"(seen) : "v"(lds_addr(lds + BASE + 2 * wave)) : "memory");
    return seen;
}
template <int BASE = kSyncHalo>
__device__ __forceinline__ void halo_check(float* lds, int wave, u2 peeked, unsigned target) {
    if (DBH_ABL & (1 | 2 | 8 | 64)) return;
    asm volatile("" : "+v"(peeked));
    if (halo_ready(peeked, target)) return;
    halo_wait<BASE>(lds, wave, target);
}

// lane n <- lane n - 1 / n + 1 of its row of 16; the row's first / last lane keeps `edge`
__device__ __forceinline__ float row_from_left(float edge, float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, edge),
                                                                 __builtin_bit_cast(int, v), 0x111,
                                                                 0xF, 0xF, false));   // row_shr:1
}
__device__ __forceinline__ float row_from_right(float edge, float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, edge),
                                                                 __builtin_bit_cast(int, v), 0x101,
                                                                 0xF, 0xF, false));   // row_shl:1
}

// The four outputs of half H (channels 4q + 2H, + 1 of the N tile) of a transposed accumulator
// set: output transform + ReLU (the clamp of each output's last instruction).
template <int H>
__device__ __forceinline__ void w43t_outputs(const f4 (&acc)[6], f2 (&y)[4]) {
    const f2 k2 = f2{2.f, 2.f}, k4 = f2{4.f, 4.f}, k8 = f2{8.f, 8.f};
    const f2 a0 = f2{acc[0][2 * H], acc[0][2 * H + 1]};
    const f2 a1 = f2{acc[1][2 * H], acc[1][2 * H + 1]};
    const f2 a2 = f2{acc[2][2 * H], acc[2][2 * H + 1]};
    const f2 a3 = f2{acc[3][2 * H], acc[3][2 * H + 1]};
    const f2 a4 = f2{acc[4][2 * H], acc[4][2 * H + 1]};
    const f2 a5 = f2{acc[5][2 * H], acc[5][2 * H + 1]};
    const f2 s12 = a1 + a2, d12 = a1 - a2, s34 = a3 + a4, d34 = a3 - a4;
    y[0] = pk_add_relu(a0 + s12, s34);
    y[1] = pk_fma_relu(k2, d34, d12);
    y[2] = pk_fma_relu(k4, s34, s12);
    y[3] = pk_fma_relu(k8, d34, d12 + a5);      // (a5 through a visible add: see pk_fma_relu)
}

// halo rows of step SP (channels 16 g + 4q + 2h, + 1) of the layer output at HOFF.
// HLAY 0: stage B's arrays (dbh_layout.h: kHalo); 1: stage D's (kDHalo: [wave][side][48], h_addr =
// the wave's left row)
template <int SP, int HOFF, int HLAY>
__device__ __forceinline__ void w43t_load_halo(f2 (&hb)[2], unsigned h_addr) {
    constexpr int c = 16 * (SP >> 1) + 2 * (SP & 1);
    if constexpr (HLAY == 0) {
        hb[0] = ds_read_f2<(HOFF + kHaloRows + 48 + c) * 4>(h_addr);     // A3[w][1]: position 4j - 1
        hb[1] = ds_read_f2<(HOFF + 96 + c) * 4>(h_addr);                 // A0[w + 1][0]: position 4j + 4
    } else {
        hb[0] = ds_read_f2<(HOFF + c) * 4>(h_addr);                      // position 4j - 1
        hb[1] = ds_read_f2<(HOFF + 48 + c) * 4>(h_addr);                 // position 4j + 4
    }
}

// One step of a chained tile.  HOFF >= 0: tile 0 of conv3 / conv4 - the step first turns
// Y[g][h] (g = SP >> 1, h = SP & 1) and its two halo positions into U[.][SP].  pre(SP) runs in
// front of the step's LDS requests (polls), side(SP) behind its MFMAs.
template <int HOFF, int STEP0, int STEPS, int SP, int CATCH, int HLAY, class Pre, class Side>
__device__ __forceinline__ void w43t_step(W43U& U, f2 (&Y)[3][2][4], unsigned h_addr,
                                          unsigned b_addr, f2 (&hbuf)[2][2], f4 (&buf)[2][3],
                                          f4 (&acc)[6], f4 bias4, bool wave_hi, const Pre& pre,
                                          const Side& side) {
    constexpr bool BUILD = HOFF >= 0 && !(DBH_ABL & 8);
    pre(IntC<SP>{});
    if constexpr (SP + 1 < 6) {
        if constexpr (BUILD) w43t_load_halo<SP + 1, (BUILD ? HOFF : 0), HLAY>(hbuf[(SP + 1) & 1], h_addr);
        w43_load_b<SP + 1>(buf[(SP + 1) & 1], b_addr);
        if constexpr (BUILD) asm volatile("s_waitcnt lgkmcnt(5)" ::: "memory");
        else asm volatile("s_waitcnt lgkmcnt(3)" ::: "memory");
    } else {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    f4(&b)[3] = buf[SP & 1];
#pragma unroll
    for (int p = 0; p < 3; ++p) asm volatile("" : "+v"(b[p]));
    if constexpr (SP < CATCH) {
        // the last tile before a workgroup barrier: the younger wave of the SIMD, which runs ~1k
        // cycles behind its partner through all of stage B (the older one wins every tie), gets
        // the pipe first for a few steps, so that the two reach the barrier together instead of
        // the older one waiting there while the younger one finishes alone
        if (wave_hi) __builtin_amdgcn_s_setprio(3);
        else __builtin_amdgcn_s_setprio(0);
    } else {
        progress_priority_pair<STEP0 + SP, STEPS>(wave_hi);
    }
    if constexpr (BUILD) {
        f2(&hb)[2] = hbuf[SP & 1];
        asm volatile("" : "+v"(hb[0]), "+v"(hb[1]));
        __builtin_amdgcn_sched_barrier(0);
        const f2(&yy)[4] = Y[SP >> 1][SP & 1];
        const f2 d0 = (DBH_ABL & 256) ? hb[0] + yy[3]
                                      : f2{row_from_left(hb[0].x, yy[3].x), row_from_left(hb[0].y, yy[3].y)};
        const f2 d5 = (DBH_ABL & 256) ? hb[1] + yy[0]
                                      : f2{row_from_right(hb[1].x, yy[0].x), row_from_right(hb[1].y, yy[0].y)};
        const f2 m4 = f2{-4.f, -4.f}, p2 = f2{2.f, 2.f}, m2 = f2{-2.f, -2.f};
        const f2 p4 = f2{4.f, 4.f}, m5 = f2{-5.f, -5.f};
        const f2 a = __builtin_elementwise_fma(m4, yy[1], yy[3]), b2 = __builtin_elementwise_fma(m4, yy[0], yy[2]);
        const f2 c = yy[3] - yy[1], g = yy[2] - yy[0];
        U.u[0][SP] = __builtin_elementwise_fma(p4, d0, __builtin_elementwise_fma(m5, yy[1], yy[3]));
        U.u[1][SP] = a + b2;
        U.u[2][SP] = a - b2;
        U.u[3][SP] = __builtin_elementwise_fma(p2, g, c);
        U.u[4][SP] = __builtin_elementwise_fma(m2, g, c);
        U.u[5][SP] = __builtin_elementwise_fma(p4, yy[0], __builtin_elementwise_fma(m5, yy[2], d5));
#pragma unroll
        for (int x = 0; x < 6; ++x) asm volatile("" : "+v"(U.u[x][SP]));
    }
    __builtin_amdgcn_sched_barrier(0);
    w43t_mfmas<SP>(U, b, acc, bias4);
    __builtin_amdgcn_sched_barrier(0);
    side(IntC<SP>{});
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (SP + 1 < 6)
        w43t_step<HOFF, STEP0, STEPS, SP + 1, CATCH, HLAY>(U, Y, h_addr, b_addr, hbuf, buf, acc, bias4, wave_hi,
                                                           pre, side);
}

// bias_addr: this lane's place in the LDS parameter table (kParams + 4q); BIAS_OFF: floats from
// there to the tile's four biases - requested first, so that step 0's hand-counted wait retires it
// (a load the compiler sees would be waited for with lgkmcnt(0), fragment requests and all)
template <int HOFF, int STEP0, int STEPS, int BIAS_OFF, int CATCH = 0, int HLAY = 0, class Pre, class Side>
__device__ __forceinline__ void w43t_tile(W43U& U, f2 (&Y)[3][2][4], unsigned h_addr,
                                          unsigned bias_addr, const float* slot_lane,
                                          f4 (&acc)[6], bool wave_hi, const Pre& pre,
                                          const Side& side) {
    const unsigned b_addr = lds_addr(slot_lane);
    f2 hbuf[2][2];
    f4 buf[2][3];
    const f4 bias4 = ds_read_f4<BIAS_OFF * 4>(bias_addr);
    if constexpr (HOFF >= 0 && !(DBH_ABL & 8)) w43t_load_halo<0, (HOFF >= 0 ? HOFF : 0), HLAY>(hbuf[0], h_addr);
    w43_load_b<0>(buf[0], b_addr);
    w43t_step<HOFF, STEP0, STEPS, 0, CATCH, HLAY>(U, Y, h_addr, b_addr, hbuf, buf, acc, bias4, wave_hi, pre, side);
}

struct NoPre {
    template <int I>
    __device__ __forceinline__ void operator()(IntC<I>) const {}
};

// in_a: conv1d_1's operands (w43a_tile0); thirds_here: thirds 1 and 2 of conv2's weights are
// requested by between_a (between conv1d_1's MFMAs) instead of having landed before the window's
// first barrier; after_first(): the caller's work behind this wave's first arrival (everything it
// asked of global memory has landed there).
template <class BetweenA, class AfterFirst, class DumpBase>
__device__ __forceinline__ void stage_b_chain(float* lds, const float* __restrict__ packed, int tid,
                                              int lane, int wave, unsigned* ts,
                                              unsigned& chain_windows, const ConvAIn& in_a,
                                              bool thirds_here, const BetweenA& between_a,
                                              const AfterFirst& after_first,
                                              const DumpBase& dump_b_base, const float* third0) {
    const int n = lane & 15, q = lane >> 4;
    const unsigned tiles0 = chain_windows * 24u, halos0 = chain_windows * 7u;
    chain_windows += 1;
    const unsigned h_addr = lds_addr(lds + kHalo + wave * 96 + 4 * q);
    // where a lane's row 0 (A0) and row 3 (A3) go per layer output: the halo arrays for the lanes
    // that hold the wave's first and last quad, scratch for the others
    // (one compare: written as n == 0 || n == 15 hipcc lowers the selects below to a switch over
    // exec masks - and lost the first of them)
    const bool is_edge = ((n + 1) & 15) < 2;
    const unsigned edge_addr = lds_addr(lds + kHalo + (wave * 2 + (n == 15 ? 1 : 0)) * 48 + 4 * q);
    const unsigned dummy_addr = lds_addr(lds + kChainDummy + 2 * lane);
    unsigned a0_addr[2], a3_addr[2];
#pragma unroll
    for (int L = 0; L < 2; ++L) {
        a0_addr[L] = is_edge ? edge_addr + L * 2 * kHaloRows * 4 : dummy_addr;
        a3_addr[L] = is_edge ? edge_addr + (L * 2 * kHaloRows + kHaloRows + 96) * 4 : dummy_addr;
        asm volatile("" : "+v"(a0_addr[L]), "+v"(a3_addr[L]));
    }
    const unsigned post_addr = halo_post_address(lds, wave, lane);
    const unsigned arrive_addr =
        lds_addr(lane == 0 ? lds + kSyncTiles : lds + kChainDummy + 176 + 64 + lane);
    const f4* tab4 = reinterpret_cast<const f4*>(lds + kParams + 4 * q);
    const unsigned bias_addr = lds_addr(lds + kParams + 4 * q);
    static_assert((bias_offset(1) - kTabBias0) % 4 == 0 && (bn_scale_offset(1) - kTabBn0) % 4 == 0 &&
                  (kTabBias1 - kTabBias0) % 4 == 0, "");
    constexpr int B2 = bias_offset(1) - kTabBias0, B3 = bias_offset(2) - kTabBias0,
                  B4 = bias_offset(3) - kTabBias0;
    const bool wave_hi = wave >= 4;
    W43U U;
    f4 acc[2][6];
    f2 Y[3][2][4];
    // half h of the epilogue of tile g of conv2 (L = 0) / conv3 (L = 1): outputs to Y, the wave's
    // two edge rows to the halo arrays, and the post that announces a finished tile
    auto finish = [&](auto L_tag, auto g_tag, auto h_tag, const f4(&a)[6]) {
        constexpr int L = decltype(L_tag)::value, g = decltype(g_tag)::value, h = decltype(h_tag)::value;
        if (DBH_ABL & 8) return;
        w43t_outputs<h>(a, Y[g][h]);
        if (DBH_ABL & 2) return;
        asm volatile("ds_write_b64 %0, %2 offset:%4\n\tds_write_b64 %1, %3 offset:%4"
                     :
                     : "v"(a0_addr[L]), "v"(a3_addr[L]), "v"(Y[g][h][0]), "v"(Y[g][h][3]),
                       "n"((16 * g + 2 * h) * 4)
                     : "memory");
        if constexpr (h == 1) halo_post(post_addr);
    };
    // half h of the epilogue of tile g of conv4: outputs, MaxPool2 -> X[g][h][p]: channels 16g +
    // 4q + 2h, + 1 of pooled positions 2j + p - conv5's B operand as it stands
    f2 X[3][2][2];
    auto store = [&](auto g_tag, auto h_tag, const f4(&a)[6]) {
        constexpr int g = decltype(g_tag)::value, h = decltype(h_tag)::value;
        if (DBH_ABL & 16) return;
        f2 y[4];
        w43t_outputs<h>(a, y);
        // (BN2 is folded into conv5's weights and bias by the packer - dbh_api.hip: pack_weights -
        // so the pooled values go to conv5 as they are; only the debug dump applies it)
        X[g][h][0] = max_raw2(y[0], y[1]);
        X[g][h][1] = max_raw2(y[2], y[3]);
        if (!DBH_FOLD_BN2) {
            const f4 sc4 = tab4[((kTabBias1 - kTabBias0) + (bn_scale_offset(1) - kTabBn0)) / 4 + 4 * g];
            const f4 sh4 = tab4[((kTabBias1 - kTabBias0) + (bn_shift_offset(1) - kTabBn0)) / 4 + 4 * g];
            const f2 sc = h ? f2{sc4.z, sc4.w} : f2{sc4.x, sc4.y}, sh = h ? f2{sh4.z, sh4.w} : f2{sh4.x, sh4.y};
            X[g][h][0] = __builtin_elementwise_fma(X[g][h][0], sc, sh);
            X[g][h][1] = __builtin_elementwise_fma(X[g][h][1], sc, sh);
        }
        if (in_a.dump_b) {          // debug_stage 1 (wave-uniform)
            const f4 sc4 = tab4[((kTabBias1 - kTabBias0) + (bn_scale_offset(1) - kTabBn0)) / 4 + 4 * g];
            const f4 sh4 = tab4[((kTabBias1 - kTabBias0) + (bn_shift_offset(1) - kTabBn0)) / 4 + 4 * g];
            const f2 sc = h ? f2{sc4.z, sc4.w} : f2{sc4.x, sc4.y}, sh = h ? f2{sh4.z, sh4.w} : f2{sh4.x, sh4.y};
            float* dst = dump_b_base() + 2 * (wave * 16 + n) * 48 + 4 * q;
#pragma unroll
            for (int pp = 0; pp < 2; ++pp) {
                const f2 v = DBH_FOLD_BN2 ? __builtin_elementwise_fma(X[g][h][pp], sc, sh) : X[g][h][pp];
                dst[pp * 48 + 16 * g + 2 * h] = v.x * kActUnscale;
                dst[pp * 48 + 16 * g + 2 * h + 1] = v.y * kActUnscale;
            }
        }
    };
    // conv5 (1x1, 48 -> 16) on X, transposed like everything here: A = its weights (M = its 16
    // output channels), B = X of one pooled position per quad - 24 MFMAs per wave, two chains;
    // channel group g's eight as soon as X[g] exists (inside conv4's last tile for g = 0, 1)
    f2 w5[6];
    f4 acc5[2];
    auto conv5_group = [&](auto g_tag) {
        constexpr int g = decltype(g_tag)::value;
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int pp = 0; pp < 2; ++pp) {
                if (g == 0 && h == 0)
                    acc5[pp] = mfma4(w5[0].x, X[0][0][pp].x, tab4[(bias_offset(4) - kTabBias0) / 4]);
                else
                    acc5[pp] = mfma4(w5[2 * g + h].x, X[g][h][pp].x, acc5[pp]);
                acc5[pp] = mfma4(w5[2 * g + h].y, X[g][h][pp].y, acc5[pp]);
            }
        asm volatile("" : "+v"(acc5[0]), "+v"(acc5[1]));
    };
    static_assert((bias_offset(4) - kTabBias0) % 4 == 0 && kConv[4].taps == 1 && kConv[4].cout_pad == 16, "");
    static_assert((bn_shift_offset(1) - kTabBn0) % 4 == 0 && kS48 % 2 == 0, "");
    const IntC<0> c0;
    const IntC<1> c1;
    const IntC<2> c2;
    constexpr int NW = DBH_DMA_WAVES;
    constexpr int kThirdSteps = (kWinoHalf / 256 + NW - 1) / NW;       // 3 (5 with four waves)
#ifndef DBH_THIRD_PACK
#define DBH_THIRD_PACK 1
#endif
    // request(s) of step `step` of a third: with four requesting waves a wave has five pieces; one
    // per step would put the last behind step 4, a step in front of the arrival that waits for it
    // to land - two per step are through by step 2
    auto third = [&](int conv, int t, float* dst, int step) {
        const float* src = packed + weight_offset(conv) + t * kWinoHalf;
        if (DBH_THIRD_PACK && NW < 8) {
            if (2 * step < kThirdSteps) dma_weights_one<kWinoHalf, NW>(src, dst, lane, wave, 2 * step);
            if (2 * step + 1 < kThirdSteps) dma_weights_one<kWinoHalf, NW>(src, dst, lane, wave, 2 * step + 1);
        } else {
            dma_weights_one<kWinoHalf, NW>(src, dst, lane, wave, step);
        }
    };
    // conv3's 54 pieces: request i of this wave (7 per wave with eight requesters, 14 with four)
    auto conv3_piece = [&](int i) {
        dma_weights_one<3 * kWinoHalf, NW>(packed + weight_offset(2), lds + kChainW3, lane, wave, i);
    };
    // words peeked a step ahead of where they are looked at (chain_peek / halo_peek)
    unsigned pk_a = 0, pk_b = 0;
    u2 pk_h = u2{0u, 0u};

    // ---- conv2 (its tile 0 computes conv1d_1 itself).  conv3's 54 pieces go to the idle
    // activation buffer one behind each of the first seven steps.
    {
        const f4 bias2 = tab4[B2 / 4];
        w43a_tile0(U, in_a, lds, third0, lane, acc[0], bias2, between_a, [&](auto tag) {
            constexpr int SP = decltype(tag)::value;
            if constexpr (NW == 8) {
                conv3_piece(SP);
            } else {          // 2, 2, 2, 1, 1, 1
                if constexpr (SP < 3) {
                    conv3_piece(2 * SP);
                    conv3_piece(2 * SP + 1);
                } else {
                    conv3_piece(SP + 3);
                }
            }
        });
    }
    mark(ts, 2);
    if (in_a.stop) return;     // debug_stage 0 / 100: stage A only
    chain_arrive(arrive_addr, 0);
    after_first();
    if (thirds_here) chain_wait(lds, 0, tiles0 + 8);      // slots 1, 2: every wave's pieces landed
    mark(ts, 3);
    w43t_tile<-1, 6, 18, B2 + 16>(U, Y, h_addr, bias_addr, lds + kSlot1 + lane * 4, acc[1], wave_hi, NoPre(),
                                  [&](auto tag) {
                                      constexpr int SP = decltype(tag)::value;
                                      if constexpr (NW == 8) {
                                          if constexpr (SP == 0) conv3_piece(6);
                                      } else {          // 2, 1, 1, 1
                                          if constexpr (SP == 0) conv3_piece(9);
                                          if constexpr (SP < 4) conv3_piece(10 + SP);
                                      }
                                      if constexpr (SP == 1) finish(c0, c0, c0, acc[0]);
                                      if constexpr (SP == 3) finish(c0, c0, c1, acc[0]);
                                  });
    chain_arrive(arrive_addr, 1);
    mark(ts, 4);
    w43t_tile<-1, 12, 18, B2 + 32>(
        U, Y, h_addr, bias_addr, lds + kSlot2 + lane * 4, acc[0], wave_hi,
        [&](auto tag) {
            if constexpr (decltype(tag)::value == 5) {
                pk_a = chain_peek(lds, 1);
                pk_h = halo_peek(lds, wave);
            }
        },
        [&](auto tag) {
            constexpr int SP = decltype(tag)::value;
            if constexpr (SP == 1) finish(c0, c1, c0, acc[1]);
            if constexpr (SP == 3) finish(c0, c1, c1, acc[1]);
        });
    chain_arrive(arrive_addr, 2);
    mark(ts, 5);

    // ---- conv3.  Tile 2 of conv2 is finished inside the first two steps (its Y is needed in
    // steps 4 and 5, its edge rows by the neighbours in theirs); conv4's third t follows conv2's
    // out of slot t.
    chain_check(lds, 1, pk_a, tiles0 + 8);       // conv3's weights have landed (slots 0, 1 are free)
    halo_check(lds, wave, pk_h, halos0 + 2);
    mark(ts, 6);
    w43t_tile<0, 0, 18, B3>(
        U, Y, h_addr, bias_addr, lds + kChainW3 + lane * 4, acc[1], wave_hi,
        [&](auto tag) {
            constexpr int SP = decltype(tag)::value;
            if constexpr (SP == 2) pk_h = halo_peek(lds, wave);
            if constexpr (SP == 3) halo_check(lds, wave, pk_h, halos0 + 3);
        },
        [&](auto tag) {
            constexpr int SP = decltype(tag)::value;
            if constexpr (SP == 0) finish(c0, c2, c0, acc[0]);
            if constexpr (SP == 1) finish(c0, c2, c1, acc[0]);
            if constexpr (SP < kThirdSteps) third(3, 0, lds + kSlot0, SP);
        });
    chain_arrive(arrive_addr, 0);
    mark(ts, 7);
    w43t_tile<-1, 6, 18, B3 + 16>(
        U, Y, h_addr, bias_addr, lds + kChainW3 + kWinoHalf + lane * 4, acc[0], wave_hi,
        [&](auto tag) {
            if constexpr (decltype(tag)::value == 5) pk_a = chain_peek(lds, 2);
        },
        [&](auto tag) {
            constexpr int SP = decltype(tag)::value;
            if constexpr (SP == 1) finish(c1, c0, c0, acc[1]);
            if constexpr (SP == 3) finish(c1, c0, c1, acc[1]);
            if constexpr (SP < kThirdSteps) third(3, 1, lds + kSlot1, SP);
        });
    chain_arrive(arrive_addr, 1);
    mark(ts, 8);
    w43t_tile<-1, 12, 18, B3 + 32>(
        U, Y, h_addr, bias_addr, lds + kChainW3 + 2 * kWinoHalf + lane * 4, acc[1], wave_hi,
        [&](auto tag) {
            constexpr int SP = decltype(tag)::value;
            // slot 2: every wave has left conv2's tile 2
            if constexpr (SP == 0) chain_check(lds, 2, pk_a, tiles0 + 8);
            if constexpr (SP == 5) {
                pk_a = chain_peek(lds, 0);
                pk_h = halo_peek(lds, wave);
            }
        },
        [&](auto tag) {
            constexpr int SP = decltype(tag)::value;
            if constexpr (SP == 1) finish(c1, c1, c0, acc[0]);
            if constexpr (SP == 3) finish(c1, c1, c1, acc[0]);
            if constexpr (SP < kThirdSteps) third(3, 2, lds + kSlot2, SP);
        });
    chain_arrive(arrive_addr, 2);
    mark(ts, 9);

    // ---- conv4 + MaxPool + BN2 -> rows 1..256 of the activation buffer.  Its first store (and
    // conv5's / conv6's weights) must find every wave out of conv3, whose weights lie there.
    chain_check(lds, 0, pk_a, tiles0 + 16);      // conv4's first third has landed
    halo_check(lds, wave, pk_h, halos0 + 5);
    mark(ts, 10);
    w43t_tile<2 * kHaloRows, 0, 18, B4>(
        U, Y, h_addr, bias_addr, lds + kSlot0 + lane * 4, acc[0], wave_hi,
        [&](auto tag) {
            constexpr int SP = decltype(tag)::value;
            if constexpr (SP == 2) pk_h = halo_peek(lds, wave);
            if constexpr (SP == 3) halo_check(lds, wave, pk_h, halos0 + 6);
            if constexpr (SP == 5) {
                pk_a = chain_peek(lds, 1);
                pk_b = chain_peek(lds, 2);
            }
        },
        [&](auto tag) {
            constexpr int SP = decltype(tag)::value;
            if constexpr (SP == 0) finish(c1, c2, c0, acc[1]);
            if constexpr (SP == 1) finish(c1, c2, c1, acc[1]);
            // conv5's weights (three pieces, waves 0-2): its first MFMAs run in tile 2
            if constexpr (SP == 2)
                dma_weights<conv_weight_floats(4)>(packed + weight_offset(4), lds + kW5, lane, wave);
        });
    chain_arrive(arrive_addr, 0);
    mark(ts, 11);
    chain_check(lds, 1, pk_a, tiles0 + 16);
    chain_check(lds, 2, pk_b, tiles0 + 16);      // nobody reads conv3's weights any more
    mark(ts, 57);
    w43t_tile<-1, 6, 18, B4 + 16>(
        U, Y, h_addr, bias_addr, lds + kSlot1 + lane * 4, acc[1], wave_hi,
        [&](auto tag) {
            if constexpr (decltype(tag)::value == 5) pk_a = chain_peek(lds, 0);
        },
        [&](auto tag) {
            constexpr int SP = decltype(tag)::value;
            // conv6's weights (9 pieces), to the upper buffer
            if constexpr (SP == 0)
                dma_weights_one<conv_weight_floats(5), NW>(packed + weight_offset(5), lds + kW6, lane, wave, 0);
            if constexpr (SP == (DBH_THIRD_PACK ? 1 : 2))
                dma_weights_one<conv_weight_floats(5), NW>(packed + weight_offset(5), lds + kW6, lane, wave, 1);
            if constexpr (SP == (DBH_THIRD_PACK ? 2 : 4) && NW < 8)
                dma_weights_one<conv_weight_floats(5), NW>(packed + weight_offset(5), lds + kW6, lane, wave, 2);
            if constexpr (SP == 1) store(c0, c0, acc[0]);
            if constexpr (SP == 3) store(c0, c1, acc[0]);
        });
    chain_arrive(arrive_addr, 1);
    mark(ts, 12);
    w43t_tile<-1, 12, 18, B4 + 32, DBH_CATCHUP>(
        U, Y, h_addr, bias_addr, lds + kSlot2 + lane * 4, acc[0], wave_hi,
        [&](auto tag) {
            constexpr int SP = decltype(tag)::value;
            // slot 0: every wave has left conv4's tile 0 - conv7's first third goes there, and
            // conv5's weights (asked for in that tile) have landed: its six fragment pairs ride in
            // front of this step's requests
            if constexpr (SP == 0) {
                chain_check(lds, 0, pk_a, tiles0 + 24);
                const unsigned w5_addr = lds_addr(lds + kW5 + lane * 2);
                w5[0] = ds_read_f2<0 * 512>(w5_addr);
                w5[1] = ds_read_f2<1 * 512>(w5_addr);
                w5[2] = ds_read_f2<2 * 512>(w5_addr);
                w5[3] = ds_read_f2<3 * 512>(w5_addr);
                w5[4] = ds_read_f2<4 * 512>(w5_addr);
                w5[5] = ds_read_f2<5 * 512>(w5_addr);
            }
        },
        [&](auto tag) {
            constexpr int SP = decltype(tag)::value;
            if constexpr (SP == 0) {
#pragma unroll
                for (int k = 0; k < 6; ++k) asm volatile("" : "+v"(w5[k]));
            }
            if constexpr (SP == 1) store(c1, c0, acc[1]);
            if constexpr (SP == 3) store(c1, c1, acc[1]);
            if constexpr (SP == 2) conv5_group(c0);
            if constexpr (SP == 5) conv5_group(c1);
            if constexpr (SP < kThirdSteps) third(6, 0, lds + kSlot0, SP);
        });
    mark(ts, 58);
    store(c2, c0, acc[0]);
    store(c2, c1, acc[0]);
    conv5_group(c2);
    chain_arrive(arrive_addr, 2);
    mark(ts, 13);
    if (in_a.dump_b) return;      // debug_stage 1: stage B's output is out
    // ---- conv6 (F(2,3), 16 -> 48 channels, L = 256, ReLU) on conv5's output, which never leaves
    // the registers either: lane (n, q) holds channels 4q + r of pooled positions 2j, 2j + 1 =
    // pair j's own two inputs d1, d2; d0 and d3 are the neighbouring lanes' (the neighbouring
    // waves' at the ends: one more halo exchange).  U1 = d1 + d2 and U2 = d2 - d1 need no halo:
    // their 24 MFMAs run while the neighbours' rows arrive, then U0 = d0 - d2, U3 = d1 - d3 and
    // the other 24.  The weights (by N tile: [t][sp][matrix pair][lane][matrix][e], k-step
    // 2 sp + e <-> channels {4q + 2 sp + e}) are read once, twelve 16-byte fragments.
    chain_check(lds, 1, pk_a, tiles0 + 24);    // conv6's weights have landed; every wave has left slot 1
    f4 wf[3][2][2];
    {
        const unsigned w6_addr = lds_addr(lds + kW6 + lane * 4);
        wf[0][0][0] = ds_read_f4<0 * 1024>(w6_addr);
        wf[0][0][1] = ds_read_f4<1 * 1024>(w6_addr);
        wf[0][1][0] = ds_read_f4<2 * 1024>(w6_addr);
        wf[0][1][1] = ds_read_f4<3 * 1024>(w6_addr);
        wf[1][0][0] = ds_read_f4<4 * 1024>(w6_addr);
        wf[1][0][1] = ds_read_f4<5 * 1024>(w6_addr);
        wf[1][1][0] = ds_read_f4<6 * 1024>(w6_addr);
        wf[1][1][1] = ds_read_f4<7 * 1024>(w6_addr);
        wf[2][0][0] = ds_read_f4<8 * 1024>(w6_addr);
        wf[2][0][1] = ds_read_f4<9 * 1024>(w6_addr);
        wf[2][1][0] = ds_read_f4<10 * 1024>(w6_addr);
        wf[2][1][1] = ds_read_f4<11 * 1024>(w6_addr);
    }
    static_assert(wino2_by_tile(5) && kConv[5].cin == 16 && kConv[5].cout_pad == 48, "");
    // conv7's second third follows conv4's out of slot 1
    dma_weights<kWinoHalf>(packed + weight_offset(6) + kWinoHalf, lds + kSlot1, lane, wave);
    f4 Z[2];
#pragma unroll
    for (int pp = 0; pp < 2; ++pp) {
        const f4 v = acc5[pp];
        Z[pp] = f4{fmaxf(v.x, 0.f), fmaxf(v.y, 0.f), fmaxf(v.z, 0.f), fmaxf(v.w, 0.f)};
    }
    {   // the wave's first and last position to its neighbours (conv2's halo arrays, long dead)
        const f2 z0l = f2{Z[0].x, Z[0].y}, z0h = f2{Z[0].z, Z[0].w};
        const f2 z1l = f2{Z[1].x, Z[1].y}, z1h = f2{Z[1].z, Z[1].w};
        asm volatile("ds_write_b64 %0, %2\n\tds_write_b64 %0, %3 offset:8\n\t"
                     "ds_write_b64 %1, %4\n\tds_write_b64 %1, %5 offset:8"
                     :
                     : "v"(a0_addr[0]), "v"(a3_addr[0]), "v"(z0l), "v"(z0h), "v"(z1l), "v"(z1h)
                     : "memory");
        halo_post(post_addr);
    }
    pk_b = chain_peek(lds, 2);
    mark(ts, 14);
    const f4 U1 = Z[0] + Z[1], U2 = Z[1] - Z[0];
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
    for (int t = 0; t < 3; ++t)
#pragma unroll
        for (int sp = 0; sp < 2; ++sp) asm volatile("" : "+v"(wf[t][sp][0]), "+v"(wf[t][sp][1]));
    f4 M[3][4];
    const f4 zero4 = f4{0.f, 0.f, 0.f, 0.f};
    f4 hl, hr;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
#pragma unroll
        for (int t = 0; t < 3; ++t) {
            M[t][1] = mfma4(wf[t][k >> 1][0][2 + (k & 1)], U1[k], k == 0 ? zero4 : M[t][1]);
            M[t][2] = mfma4(wf[t][k >> 1][1][k & 1], U2[k], k == 0 ? zero4 : M[t][2]);
        }
        if (k == 1) {
            // half-way: have the neighbours posted?  Their rows are asked for in the same breath -
            // if the answer (the older of the two) is yes, what comes back is what they stored
            __builtin_amdgcn_sched_barrier(0);
            pk_h = halo_peek(lds, wave);
            hl = ds_read_f4<(kHaloRows + 48) * 4>(h_addr);     // A3[w][1]: position 2j - 1
            hr = ds_read_f4<96 * 4>(h_addr);                   // A0[w + 1][0]: position 2j + 2
            __builtin_amdgcn_sched_barrier(0);
        }
    }
#pragma unroll
    for (int t = 0; t < 3; ++t) asm volatile("" : "+v"(M[t][1]), "+v"(M[t][2]));
    mark(ts, 15);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    asm volatile("" : "+v"(hl), "+v"(hr), "+v"(pk_h));
    if (!(DBH_ABL & (1 | 2 | 8 | 64)) && !halo_ready(pk_h, halos0 + 7)) {
        halo_wait(lds, wave, halos0 + 7);
        hl = ds_read_f4<(kHaloRows + 48) * 4>(h_addr);
        hr = ds_read_f4<96 * 4>(h_addr);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        asm volatile("" : "+v"(hl), "+v"(hr));
    }
    mark(ts, 16);
    f4 U0, U3;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        U0[k] = row_from_left(hl[k], Z[1][k]) - Z[1][k];
        U3[k] = Z[0][k] - row_from_right(hr[k], Z[0][k]);
    }
    // conv7's last third follows conv4's out of slot 2, once every wave has left its tile 2
    // (asked for in front of the layer's first MFMAs instead: no difference, profiles/r06_steps)
    chain_check(lds, 2, pk_b, tiles0 + 24);
    dma_weights<kWinoHalf>(packed + weight_offset(6) + 2 * kWinoHalf, lds + kSlot2, lane, wave);
#pragma unroll
    for (int k = 0; k < 4; ++k)
#pragma unroll
        for (int t = 0; t < 3; ++t) {
            const f4 b6 = tab4[(bias_offset(5) - kTabBias0) / 4 + 4 * t];
            M[t][0] = mfma4(wf[t][k >> 1][0][k & 1], U0[k], k == 0 ? b6 : M[t][0]);
            M[t][3] = mfma4(wf[t][k >> 1][1][2 + (k & 1)], U3[k], k == 0 ? -b6 : M[t][3]);
        }
    static_assert((bias_offset(5) - kTabBias0) % 4 == 0, "");
    mark(ts, 17);
    mark(ts, 18);
    // even = M0 + M1 + M2, odd = M1 - M2 - M3 (M0 started at the bias, M3 at minus the bias), ReLU,
    // rows 2j and 2j + 1 of the image conv7 reads
    {
        lds_f2* out6 = (lds_f2*)lds_pinned(lds + kActOff + (1 + 2 * (wave * 16 + n)) * kS48 + 4 * q);
#pragma unroll
        for (int t = 0; t < 3; ++t)
#pragma unroll
            for (int h = 0; h < 2; ++h) {       // (packed; ReLU = the clamp of the last instruction)
                const f2 m0 = f2{M[t][0][2 * h], M[t][0][2 * h + 1]}, m1 = f2{M[t][1][2 * h], M[t][1][2 * h + 1]};
                const f2 m2 = f2{M[t][2][2 * h], M[t][2][2 * h + 1]}, m3 = f2{M[t][3][2 * h], M[t][3][2 * h + 1]};
                // (the two sums the compiler can see read the youngest accumulators, M0 and M3: it
                // pads the distance to their MFMAs; it would not for an inline-asm reader)
                const f2 s01 = m0 + m1, s23 = m2 + m3;
                out6[(16 * t) / 2 + h] = pk_add_relu(s01, m2);
                out6[(kS48 + 16 * t) / 2 + h] = pk_sub_relu(m1, s23);
            }
    }
    zero_row(lds + kActOff, 0, kS48, 48, tid);
    zero_row(lds + kActOff, 257, kS48, 48, tid);
    mark(ts, 19);
    mark(ts, 20);
    full_barrier();     // conv7 reads everybody's rows; its weights have landed
    mark(ts, 21);
}

// conv1d_7's output of a wave's sixteen quads in stage D's operand order, from a park (global
// memory, or the LDS image of the group's last window): twelve 16-byte loads per lane, each
// wave-instruction one contiguous KiB.  src = the window's park + half * 3072 + lane * 4.
template <class Ptr>
__device__ __forceinline__ void d_load_y(f2 (&Y)[3][2][4], Ptr src) {
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int g = 0; g < 3; ++g) {
            const f4 v = *reinterpret_cast<const f4*>(src + (i * 3 + g) * 256);
            Y[g][0][i] = f2{v.x, v.y};
            Y[g][1][i] = f2{v.z, v.w};
        }
}

// =============================================================================================
// STAGE D FOR FOUR WINDOWS AT A TIME (round 6): conv1d_8 -> conv1d_9 -> MaxPool -> BN4 as Winograd
// F(4,3), chained in registers exactly like conv1d_3 -> conv1d_4 above.  One window has 32 quads
// at L = 128 = two tiles of 16: for ONE window eight waves had to split a tile's matrices between
// partner waves and exchange partial sums through LDS behind two barriers per layer (F(2,3), 576
// MFMAs per layer, 67 % of the matrix pipe's time).  A GROUP of kGroup = 4 windows is eight tiles:
// wave w owns half hf = w & 1 of window k = w >> 1 - lane (n, q) quad 16 hf + n - from conv1d_7's
// parked output to BN4's, 216 MFMAs per layer and wave (432 per window and layer instead of 576),
// no workgroup barrier inside.  Halo rows cross between the two waves of a window only (the
// window's two ends are zero rows the wave writes itself); the tile words count all eight waves,
// because the weight slots are shared.
//   conv1d_7's park (global memory, written by w43_epilogue_half_park in operand order) -> Y:
// twelve 16-byte loads per lane, each wave-instruction one contiguous KiB.
//   Weights: four slots of a third each.  conv8's thirds 0 and 1 were requested while the group's
// last conv1d_7 ran, third 2 and conv9's third 0 are requested in tile 0 here; conv9's thirds 1 and
// 2 follow conv8's out of slots 0 and 1.  The inception block's weights (conv10 .. conv15) arrive
// in their stage-E homes meanwhile - once per GROUP.
//   BN4's output X (66 x 50 image with its two zero rows): window 0's straight into LDS, the
// others' into their parks in global memory (LDS-DMA brings each in while the window before it
// runs its stage E2 / E3 / F).
// =============================================================================================
template <class Ahead>
__device__ __forceinline__ void stage_d_chain(float* lds, const float* __restrict__ packed,
                                              float* __restrict__ wg_scratch, int lane, int wave,
                                              unsigned* ts, unsigned& d_groups, f2 (&Y)[3][2][4],
                                              float* dump3, bool stop3, int cat_base,
                                              const Ahead& ahead) {
    const int n = lane & 15, q = lane >> 4;
    const int k = wave >> 1, hf = wave & 1;
    const unsigned tiles0 = d_groups * 8u, halos0 = d_groups * 7u;
    d_groups += 1;
    constexpr int NW = DBH_DMA_WAVES;
    static_assert(NW == 4, "the request schedule below deals pieces to four waves");
    // request(s) of step `step` of a copy of NFLOATS: two per step
    auto copy_step = [&](auto nfloats_tag, const float* src, float* dst, int step) {
        constexpr int NFLOATS = decltype(nfloats_tag)::value;
        constexpr int per_wave = (NFLOATS / 256 + NW - 1) / NW;
        if (2 * step < per_wave) dma_weights_one<NFLOATS, NW>(src, dst, lane, wave, 2 * step);
        if (2 * step + 1 < per_wave) dma_weights_one<NFLOATS, NW>(src, dst, lane, wave, 2 * step + 1);
    };
    const IntC<kWinoHalf> third_floats;
    // (Y: conv1d_7's output of this wave's sixteen quads - rows i = 0..3, channels 16g + 4q + {0..3} -
    // loaded by the caller: d_load_y)
    // this wave's halo rows: kDHalo + L * kDHaloLayer + (wave * 2 + side) * 48; the one at the
    // window's end (side hf: left of an even wave, right of an odd one) is 'same' padding
    if (lane < 48) {
        lds[kDHalo + (wave * 2 + hf) * 48 + lane] = 0.f;
        lds[kDHalo + kDHaloLayer + (wave * 2 + hf) * 48 + lane] = 0.f;
    }
    const unsigned h_addr = lds_addr(lds + kDHalo + wave * 96 + 4 * q);
    // where a lane's row 0 (an odd wave's first quad: its partner's right halo) and row 3 (an even
    // wave's last quad: its partner's left halo) go - entry (wave ^ 1, side hf) - scratch for the rest
    const bool is_edge = n == (hf ? 0 : 15);
    const unsigned edge_addr = lds_addr(lds + kDHalo + ((wave ^ 1) * 2 + hf) * 48 + 4 * q);
    const unsigned dummy_addr = lds_addr(lds + kDDummy + 2 * lane);
    unsigned a0_addr[2], a3_addr[2];
#pragma unroll
    for (int L = 0; L < 2; ++L) {
        a0_addr[L] = (is_edge && hf == 1) ? edge_addr + L * kDHaloLayer * 4 : dummy_addr;
        a3_addr[L] = (is_edge && hf == 0) ? edge_addr + L * kDHaloLayer * 4 : dummy_addr;
        asm volatile("" : "+v"(a0_addr[L]), "+v"(a3_addr[L]));
    }
    // posts: word pair kSyncDHalo + 2w = {left neighbour's, right neighbour's}; lane 0 -> the
    // partner's word for this wave, lane 1 -> this wave's own word for the neighbour it has not
    const unsigned post_addr =
        lds_addr(lane == 0   ? lds + kSyncDHalo + 2 * (wave ^ 1) + hf
                 : lane == 1 ? lds + kSyncDHalo + 2 * wave + hf
                             : lds + kDDummy + 176 + lane);
    const unsigned arrive_addr =
        lds_addr(lane == 0 ? lds + kSyncDTiles : lds + kDDummy + 176 + 64 + lane);
    const f4* tab4 = reinterpret_cast<const f4*>(lds + kParams + 4 * q);
    const unsigned bias_addr = lds_addr(lds + kParams + 4 * q);
    constexpr int B8 = bias_offset(7) - kTabBias0, B9 = bias_offset(8) - kTabBias0;
    static_assert(B8 % 4 == 0 && B9 % 4 == 0 && (bn_scale_offset(3) - kTabBn0) % 4 == 0 &&
                  bn_scale_offset(4) == kTabBn1, "");
    const bool wave_hi = wave >= 4;
    // the edge rows of conv1d_7's output to the partner wave (the loads above have landed: the
    // compiler waits in front of the first use)
#pragma unroll
    for (int g = 0; g < 3; ++g)
#pragma unroll
        for (int h = 0; h < 2; ++h)
            asm volatile("ds_write_b64 %0, %2 offset:%4\n\tds_write_b64 %1, %3 offset:%4"
                         :
                         : "v"(a0_addr[0]), "v"(a3_addr[0]), "v"(Y[g][h][0]), "v"(Y[g][h][3]),
                           "n"((16 * g + 2 * h) * 4)
                         : "memory");
    halo_post(post_addr);
    mark(ts, 26);
    W43U U;
    f4 acc[2][6];
    // half h of the epilogue of tile g of conv8: outputs to Y, the edge rows to the partner, post
    auto finish = [&](auto g_tag, auto h_tag, const f4(&a)[6]) {
        constexpr int g = decltype(g_tag)::value, h = decltype(h_tag)::value;
        w43t_outputs<h>(a, Y[g][h]);
        asm volatile("ds_write_b64 %0, %2 offset:%4\n\tds_write_b64 %1, %3 offset:%4"
                     :
                     : "v"(a0_addr[1]), "v"(a3_addr[1]), "v"(Y[g][h][0]), "v"(Y[g][h][3]),
                       "n"((16 * g + 2 * h) * 4)
                     : "memory");
        if constexpr (h == 1) halo_post(post_addr);
    };
    // half h of the epilogue of tile g of conv9: outputs, MaxPool2, BN4 -> X[g][h][p]: channels 16g +
    // 4q + 2h, + 1 of pooled positions 2j + p (j = 16 hf + n: the lane's PAIR of the window's 64) -
    // the B operand of the inception block's 1x1 convolutions as it stands
    f2 X[3][2][2];
    auto store = [&](auto g_tag, auto h_tag, const f4(&a)[6]) {
        constexpr int g = decltype(g_tag)::value, h = decltype(h_tag)::value;
        f2 y[4];
        w43t_outputs<h>(a, y);
        const f4 sc4 = tab4[((kTabBias1 - kTabBias0) + (bn_scale_offset(3) - kTabBn0)) / 4 + 4 * g];
        const f4 sh4 = tab4[((kTabBias1 - kTabBias0) + (bn_shift_offset(3) - kTabBn0)) / 4 + 4 * g];
        const f2 sc = h ? f2{sc4.z, sc4.w} : f2{sc4.x, sc4.y}, sh = h ? f2{sh4.z, sh4.w} : f2{sh4.x, sh4.y};
        X[g][h][0] = __builtin_elementwise_fma(max_raw2(y[0], y[1]), sc, sh);
        X[g][h][1] = __builtin_elementwise_fma(max_raw2(y[2], y[3]), sc, sh);
        if (dump3 != nullptr) {        // debug_stage 3 (wave-uniform): dense [64][48]
#pragma unroll
            for (int pp = 0; pp < 2; ++pp) {
                float* dst = dump3 + (32 * hf + 2 * n + pp) * 48 + 16 * g + 4 * q + 2 * h;
                dst[0] = X[g][h][pp].x * kActUnscale;
                dst[1] = X[g][h][pp].y * kActUnscale;
            }
        }
    };
    const IntC<0> c0;
    const IntC<1> c1;
    const IntC<2> c2;
    unsigned pk_a = 0;
    u2 pk_h = u2{0u, 0u};
    const float* w8 = packed + weight_offset(7);
    const float* w9 = packed + weight_offset(8);
    const float* we = packed + weight_offset(9);        // conv10 .. conv15, contiguous: 48 pieces

    // ---- conv8.  Tile 0 builds U from Y and the partner's edge rows.
    halo_wait<kSyncDHalo>(lds, wave, halos0 + 1);
    w43t_tile<0, 0, 18, B8, 0, 1>(
        U, Y, h_addr, bias_addr, lds + kDS0 + lane * 4, acc[1], wave_hi, NoPre(), [&](auto tag) {
            constexpr int SP = decltype(tag)::value;
            // conv8's third 2 -> slot 2, conv9's third 0 -> slot 3 (conv1d_7's N tiles 1 and 2 lay there)
            if constexpr (SP < 3) copy_step(third_floats, w8 + 2 * kWinoHalf, lds + kDS2, SP);
            else copy_step(third_floats, w9, lds + kDS3, SP - 3);
        });
    chain_arrive(arrive_addr, 0);
    mark(ts, 27);
    w43t_tile<-1, 6, 18, B8 + 16, 0, 1>(
        U, Y, h_addr, bias_addr, lds + kDS1 + lane * 4, acc[0], wave_hi,
        [&](auto tag) {
            if constexpr (decltype(tag)::value == 5) pk_a = chain_peek<kSyncDTiles>(lds, 0);
        },
        [&](auto tag) {
            constexpr int SP = decltype(tag)::value;
            if constexpr (SP == 1) finish(c0, c0, acc[1]);
            if constexpr (SP == 3) finish(c0, c1, acc[1]);
        });
    chain_arrive(arrive_addr, 1);
    mark(ts, 28);
    // slots 2 and 3 have landed (requested in tile 0), and every wave has left slot 0
    chain_check<kSyncDTiles>(lds, 0, pk_a, tiles0 + 8);
    w43t_tile<-1, 12, 18, B8 + 32, 0, 1>(
        U, Y, h_addr, bias_addr, lds + kDS2 + lane * 4, acc[1], wave_hi,
        [&](auto tag) {
            if constexpr (decltype(tag)::value == 5) {
                pk_a = chain_peek<kSyncDTiles>(lds, 1);
                pk_h = halo_peek<kSyncDHalo>(lds, wave);
            }
        },
        [&](auto tag) {
            constexpr int SP = decltype(tag)::value;
            if constexpr (SP == 1) finish(c1, c0, acc[0]);
            if constexpr (SP == 3) finish(c1, c1, acc[0]);
            // conv9's third 1 follows conv8's third 0 out of slot 0; conv15 (12 pieces) to its home
            // ... then the first half of conv10 .. conv15 (48 pieces, twelve per wave) to the front of
            // the arena - where the group's last window had its conv7 output until its two waves read
            // it (before their arrival behind tile 0, which every wave has seen by now)
            if constexpr (SP < 3) copy_step(third_floats, w9 + kWinoHalf, lds + kDS0, SP);
            else copy_step(IntC<kEWEnd>{}, we, lds + kEW10, SP - 3);
        });
    chain_arrive(arrive_addr, 2);
    mark(ts, 29);

    // ---- conv9 + MaxPool + BN4.  Tile 2 of conv8 is finished inside the first two steps.
    halo_check<kSyncDHalo>(lds, wave, pk_h, halos0 + 3);
    w43t_tile<kDHaloLayer, 0, 18, B9, 0, 1>(
        U, Y, h_addr, bias_addr, lds + kDS3 + lane * 4, acc[0], wave_hi,
        [&](auto tag) {
            constexpr int SP = decltype(tag)::value;
            // slot 1: every wave has left conv8's tile 1
            if constexpr (SP == 0) chain_check<kSyncDTiles>(lds, 1, pk_a, tiles0 + 8);
            if constexpr (SP == 2) pk_h = halo_peek<kSyncDHalo>(lds, wave);
            if constexpr (SP == 3) halo_check<kSyncDHalo>(lds, wave, pk_h, halos0 + 4);
            if constexpr (SP == 5) pk_a = chain_peek<kSyncDTiles>(lds, 2);
        },
        [&](auto tag) {
            constexpr int SP = decltype(tag)::value;
            if constexpr (SP == 0) finish(c2, c0, acc[1]);
            if constexpr (SP == 1) finish(c2, c1, acc[1]);
            // conv9's third 2 follows conv8's third 1 out of slot 1; then the rest of conv10 .. conv15
            if constexpr (SP < 3) copy_step(third_floats, w9 + 2 * kWinoHalf, lds + kDS1, SP);
            else copy_step(IntC<kEWEnd>{}, we, lds + kEW10, SP);
        });
    chain_arrive(arrive_addr, 3);
    mark(ts, 30);
    chain_check<kSyncDTiles>(lds, 2, pk_a, tiles0 + 8);      // conv9's third 1 has landed in slot 0
    w43t_tile<-1, 6, 18, B9 + 16, 0, 1>(
        U, Y, h_addr, bias_addr, lds + kDS0 + lane * 4, acc[1], wave_hi,
        [&](auto tag) {
            if constexpr (decltype(tag)::value == 5) pk_a = chain_peek<kSyncDTiles>(lds, 3);
        },
        [&](auto tag) {
            constexpr int SP = decltype(tag)::value;
            if constexpr (SP == 1) store(c0, c0, acc[0]);
            if constexpr (SP == 3) store(c0, c1, acc[0]);
            // conv16's first two matrices' worth -> slot 2 (every wave has left conv8's tile 2: the
            // check in front of this tile), and BN5's parameters
            if constexpr (SP < 3) copy_step(third_floats, packed + weight_offset(15), lds + kDS2, SP);
            if constexpr (SP == 3)
                dma_weights<512, 2>(packed + bn_scale_offset(4), lds + kEBn5, lane, wave);
        });
    chain_arrive(arrive_addr, 4);
    mark(ts, 31);
    chain_check<kSyncDTiles>(lds, 3, pk_a, tiles0 + 8);      // conv9's third 2 has landed in slot 1
    w43t_tile<-1, 12, 18, B9 + 32, 0, 1>(
        U, Y, h_addr, bias_addr, lds + kDS1 + lane * 4, acc[0], wave_hi,
        [&](auto tag) {
            if constexpr (decltype(tag)::value == 5) pk_a = chain_peek<kSyncDTiles>(lds, 4);
        },
        [&](auto tag) {
            constexpr int SP = decltype(tag)::value;
            if constexpr (SP == 1) store(c1, c0, acc[1]);
            if constexpr (SP == 3) store(c1, c1, acc[1]);
            // conv16's other half -> slot 3 (every wave has left conv9's tile 0: the check in front
            // of this tile)
            if constexpr (SP < 3)
                copy_step(third_floats, packed + weight_offset(15) + kWinoHalf, lds + kDS3, SP);
        });
    store(c2, c0, acc[0]);
    store(c2, c1, acc[0]);
    chain_arrive(arrive_addr, 5);
    mark(ts, 32);
    if (stop3) {        // debug_stage 3 / 103: stage D only
        ahead();
        full_barrier();
        return;
    }

    // =========================================================================================
    // STAGE E ON THE SAME REGISTERS: the inception block (network_architecture.py:54-74) for this
    // lane's pair of positions 2j, 2j + 1 of its window's 64 - transposed like everything in the
    // chain (M = 16 output channels = the weights as the A operand, N = the wave's 16 pairs):
    //   conv11 (1x1, 48 -> 48)                          72 MFMAs  -> concat 48..95
    //   conv10 (1x1 on the 3-tap average of X = the 3-tap average of its products: a 1x1 convolution
    //           commutes with a pooling along the positions)      72         -> concat 0..47
    //   conv12 (1x1 -> 16) -> conv13 (k3, F(2,3) as conv6 runs it) 24 + 48   -> concat 96..143
    //   conv14 (1x1 -> 16) -> conv15 (k3, F(2,3)) -> conv16 (k3, 48 -> 48, F(2,3): 144 instead of
    //           the direct form's 216)                            24 + 48 + 144 -> concat 144..191
    // each followed by ReLU, MaxPool2 - the max of the pair's two outputs, in-lane - and BN5.  What
    // crosses lanes: one position each side for the k = 3 layers and the average - a DPP row shift;
    // at a wave's end that meets its partner, 16 or 48 floats through LDS (the window's two ends
    // are zero rows the wave writes itself), announced on the chain's post words.  432 MFMAs per
    // wave = 864 per window, where the block as three barrier-separated phases out of LDS images
    // took 1,008 at two thirds of the pipe's rate.
    //   The output - lane (n, q): pooled position j = 16 hf + n, channels 16t + 4q + r of each branch -
    // goes to the window's concat image (34 x 196, rows 0 and 33 zero): window 0's in LDS buffer A,
    // the others' in their parks.
    // =========================================================================================
    // BN5's parameters have landed (requested in conv9's tile 1, whose arrival every wave has made)
    chain_check<kSyncDTiles>(lds, 4, pk_a, tiles0 + 8);
    const unsigned ehalos0 = halos0 + 4u;
    // (zero rows at the window's ends: the 16-channel array, written here; the 48-channel ones are
    // stage D's, whose end rows are zero already)
    if (lane < 32) lds[kEHaloZ + (wave * 2 + hf) * 32 + lane] = 0.f;
    const unsigned ez_edge = is_edge ? lds_addr(lds + kEHaloZ + ((wave ^ 1) * 2 + hf) * 32 + 4 * q) : dummy_addr;
    // (the edge lane's slot of stage D's arrays, or the scratch: exactly one of a0 / a3 is the slot)
    const unsigned z48_addr = hf ? a0_addr[0] : a3_addr[0];      // array 0: conv10's products
    const unsigned y48_addr = hf ? a0_addr[1] : a3_addr[1];      // array 1: conv15's output
    const float third = 1.f / 3.f;
    const int j = 16 * hf + n;
    // This lane's 48 outputs - pooled position j, channels 48 b + 16 t + 4q + r of branch b - stay in
    // registers to the end of the block: the concat images lie where its weights do.
    f4 OUT[4][3];
    const f4* bn5 = reinterpret_cast<const f4*>(lds + kEBn5 + 4 * q);        // + ch / 4: scale; + 48: shift
    auto pooled_out = [&](f4& dst, int ch, f4 a, f4 b) {       // ReLU'd pair -> MaxPool2 -> BN5
        const f4 sc = bn5[ch / 4], sh = bn5[48 + ch / 4];
#pragma unroll
        for (int r = 0; r < 4; ++r) dst[r] = fmaf(max_raw(a[r], b[r]), sc[r], sh[r]);
        // (pinned: the result is used under a condition at the end of the block, and the compiler
        // sinks its whole computation there - keeping four raw accumulators alive per quadruple
        // instead of the quadruple: 87 spilled registers)
        asm volatile("" : "+v"(dst));
    };
    auto relu4 = [](f4 v) { return f4{fmaxf(v.x, 0.f), fmaxf(v.y, 0.f), fmaxf(v.z, 0.f), fmaxf(v.w, 0.f)}; };
    constexpr int TB = kTabBias0;

    // ---- conv12, conv14 (1x1, 48 -> 16): Z3, Z4 = channels 4q + r of the pair's two positions
    f4 Z3[2], Z4[2];
    {
        f2 w12[6], w14[6];
        const f4 b12 = tab4[(bias_offset(11) - TB) / 4], b14 = tab4[(bias_offset(13) - TB) / 4];
#pragma unroll
        for (int sp = 0; sp < 6; ++sp) {
            w12[sp] = *reinterpret_cast<const f2*>(lds + kEW12 + sp * 128 + lane * 2);
            w14[sp] = *reinterpret_cast<const f2*>(lds + kEW14 + sp * 128 + lane * 2);
        }
#pragma unroll
        for (int sp = 0; sp < 6; ++sp)
#pragma unroll
            for (int pp = 0; pp < 2; ++pp) {
                const f2 x = X[sp >> 1][sp & 1][pp];
                Z3[pp] = mfma4(w12[sp].x, x.x, sp == 0 ? b12 : Z3[pp]);
                Z4[pp] = mfma4(w14[sp].x, x.x, sp == 0 ? b14 : Z4[pp]);
                Z3[pp] = mfma4(w12[sp].y, x.y, Z3[pp]);
                Z4[pp] = mfma4(w14[sp].y, x.y, Z4[pp]);
            }
#pragma unroll
        for (int pp = 0; pp < 2; ++pp) {
            Z3[pp] = relu4(Z3[pp]);
            Z4[pp] = relu4(Z4[pp]);
        }
        // the position that meets the partner wave: an even wave's last (2j + 1 of its lane 15),
        // an odd wave's first (2j of its lane 0)
        const f4 s3 = hf ? Z3[0] : Z3[1], s4 = hf ? Z4[0] : Z4[1];
        asm volatile("ds_write_b128 %0, %1\n\tds_write_b128 %0, %2 offset:64" ::"v"(ez_edge), "v"(s3), "v"(s4)
                     : "memory");
        halo_post(post_addr);
    }
    __builtin_amdgcn_sched_barrier(0);
    mark(ts, 34);
    // ---- conv10's products z = W10 . x for the two positions (no bias yet: the average comes first)
    f4 z10[3][2];
    {
#pragma unroll
        for (int t = 0; t < 3; ++t) {
            f2 w[6];
#pragma unroll
            for (int sp = 0; sp < 6; ++sp)
                w[sp] = *reinterpret_cast<const f2*>(lds + kEW10 + (sp * 3 + t) * 128 + lane * 2);
#pragma unroll
            for (int sp = 0; sp < 6; ++sp)
#pragma unroll
                for (int pp = 0; pp < 2; ++pp) {
                    const f2 x = X[sp >> 1][sp & 1][pp];
                    z10[t][pp] = mfma4(w[sp].x, x.x, sp == 0 ? f4{0.f, 0.f, 0.f, 0.f} : z10[t][pp]);
                    z10[t][pp] = mfma4(w[sp].y, x.y, z10[t][pp]);
                }
        }
#pragma unroll
        for (int t = 0; t < 3; ++t) {
            const f4 sz = hf ? z10[t][0] : z10[t][1];
            asm volatile("ds_write_b128 %0, %1 offset:%2" ::"v"(z48_addr), "v"(sz), "n"(t * 64) : "memory");
        }
        halo_post(post_addr);
    }
    __builtin_amdgcn_sched_barrier(0);
    mark(ts, 35);
    // the F(2,3) layers on a 16-channel pair (conv13 on Z3, conv15 on Z4), as conv6 runs at the end
    // of stage B's chain: U1 = d1 + d2, U2 = d2 - d1 in-lane, U0 = d0 - d2 and U3 = d1 - d3 with the
    // neighbours' positions; M0 starts at the bias, M3 at minus the bias; even = M0 + M1 + M2,
    // odd = M1 - M2 - M3
    auto wino16 = [&](const float* w_lds, int bias_off, const f4(&Z)[2], int zoff, f4(&even)[3], f4(&odd)[3]) {
        const f4 hl = *reinterpret_cast<const f4*>(lds + kEHaloZ + (wave * 2 + 0) * 32 + zoff + 4 * q);
        const f4 hr = *reinterpret_cast<const f4*>(lds + kEHaloZ + (wave * 2 + 1) * 32 + zoff + 4 * q);
        f4 U0, U3;
        const f4 U1 = Z[0] + Z[1], U2 = Z[1] - Z[0];
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            U0[c] = row_from_left(hl[c], Z[1][c]) - Z[1][c];
            U3[c] = Z[0][c] - row_from_right(hr[c], Z[0][c]);
        }
        // (one N tile at a time - four fragments, four chains of four MFMAs: all three at once held
        // 96 registers of fragments and accumulators beside what the block keeps)
#pragma unroll
        for (int t = 0; t < 3; ++t) {
            f4 wf[2][2];
#pragma unroll
            for (int sp = 0; sp < 2; ++sp)
#pragma unroll
                for (int pr = 0; pr < 2; ++pr)
                    wf[sp][pr] = *reinterpret_cast<const f4*>(w_lds + ((t * 2 + sp) * 2 + pr) * 256 + lane * 4);
            const f4 b = tab4[(bias_off - TB) / 4 + 4 * t];
            const f4 zero4 = f4{0.f, 0.f, 0.f, 0.f};
            f4 M0, M1, M2, M3;
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                M0 = mfma4(wf[c >> 1][0][c & 1], U0[c], c == 0 ? b : M0);
                M1 = mfma4(wf[c >> 1][0][2 + (c & 1)], U1[c], c == 0 ? zero4 : M1);
                M2 = mfma4(wf[c >> 1][1][c & 1], U2[c], c == 0 ? zero4 : M2);
                M3 = mfma4(wf[c >> 1][1][2 + (c & 1)], U3[c], c == 0 ? -b : M3);
            }
            even[t] = relu4(M0 + M1 + M2);
            odd[t] = relu4(M1 - M2 - M3);
        }
    };
    // ---- conv11 (1x1, 48 -> 48) -> concat 48..95
    {
#pragma unroll
        for (int t = 0; t < 3; ++t) {
            f2 w[6];
#pragma unroll
            for (int sp = 0; sp < 6; ++sp)
                w[sp] = *reinterpret_cast<const f2*>(lds + kEW11 + (sp * 3 + t) * 128 + lane * 2);
            const f4 b = tab4[(bias_offset(10) - TB) / 4 + 4 * t];
            f4 a[2];
#pragma unroll
            for (int sp = 0; sp < 6; ++sp)
#pragma unroll
                for (int pp = 0; pp < 2; ++pp) {
                    const f2 x = X[sp >> 1][sp & 1][pp];
                    a[pp] = mfma4(w[sp].x, x.x, sp == 0 ? b : a[pp]);
                    a[pp] = mfma4(w[sp].y, x.y, a[pp]);
                }
            pooled_out(OUT[1][t], 48 + 16 * t, relu4(a[0]), relu4(a[1]));
        }
    }
    __builtin_amdgcn_sched_barrier(0);
    mark(ts, 39);
    // ---- conv13 -> concat 96..143
    halo_wait<kSyncDHalo>(lds, wave, ehalos0 + 1);
    {
        f4 ev[3], od[3];
        wino16(lds + kEW13, bias_offset(12), Z3, 0, ev, od);
#pragma unroll
        for (int t = 0; t < 3; ++t) pooled_out(OUT[2][t], 96 + 16 * t, ev[t], od[t]);
    }
    __builtin_amdgcn_sched_barrier(0);
    mark(ts, 36);
#if DBH_F_AHEAD_EARLY
    ahead();
    __builtin_amdgcn_sched_barrier(0);
#endif
    // ---- conv15 -> the pair of conv16's input positions (48 channels: Y16[t][p] = channels 16t + 4q + r)
    f4 Y16[3][2];
    {
        f4 ev[3], od[3];
        wino16(lds + kEW15, bias_offset(14), Z4, 16, ev, od);
#pragma unroll
        for (int t = 0; t < 3; ++t) {
            Y16[t][0] = ev[t];
            Y16[t][1] = od[t];
            const f4 sy = hf ? ev[t] : od[t];
            asm volatile("ds_write_b128 %0, %1 offset:%2" ::"v"(y48_addr), "v"(sy), "n"(t * 64) : "memory");
        }
        halo_post(post_addr);
    }
    __builtin_amdgcn_sched_barrier(0);
    mark(ts, 37);
    // ---- conv10: the average over positions p - 1, p, p + 1 of its products (TensorFlow's valid-count
    // divisor: two taps at a window's end), bias, ReLU -> concat 0..47
    halo_wait<kSyncDHalo>(lds, wave, ehalos0 + 2);
    {
        const float inv_first = j == 0 ? 0.5f : third, inv_last = j == 31 ? 0.5f : third;
#pragma unroll
        for (int t = 0; t < 3; ++t) {
            const f4 hl = *reinterpret_cast<const f4*>(lds + kDHalo + (wave * 2 + 0) * 48 + 16 * t + 4 * q);
            const f4 hr = *reinterpret_cast<const f4*>(lds + kDHalo + (wave * 2 + 1) * 48 + 16 * t + 4 * q);
            const f4 b = tab4[(bias_offset(9) - TB) / 4 + 4 * t];
            f4 y0, y1;
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const float prev = row_from_left(hl[c], z10[t][1][c]);       // position 2j - 1
                const float next = row_from_right(hr[c], z10[t][0][c]);      // position 2j + 2
                const float mid = z10[t][0][c] + z10[t][1][c];
                y0[c] = fmaxf(fmaf(prev + mid, inv_first, b[c]), 0.f);
                y1[c] = fmaxf(fmaf(mid + next, inv_last, b[c]), 0.f);
            }
            pooled_out(OUT[0][t], 16 * t, y0, y1);
        }
    }
    __builtin_amdgcn_sched_barrier(0);
    mark(ts, 38);
    // ---- conv16 (k3, 48 -> 48) as F(2,3) on the pair -> concat 144..191.  Its four matrices lie in
    // stage D's slots 2 and 3 (requested in its last two tiles; every wave's pieces have landed when
    // every wave has arrived behind them).
    chain_wait<kSyncDTiles>(lds, 5, tiles0 + 8);
    halo_wait<kSyncDHalo>(lds, wave, ehalos0 + 3);
    {
        f4 U0[3], U1[3], U2[3], U3[3];
#pragma unroll
        for (int t = 0; t < 3; ++t) {
            const f4 hl = *reinterpret_cast<const f4*>(lds + kDHalo + kDHaloLayer + (wave * 2 + 0) * 48 + 16 * t + 4 * q);
            const f4 hr = *reinterpret_cast<const f4*>(lds + kDHalo + kDHaloLayer + (wave * 2 + 1) * 48 + 16 * t + 4 * q);
            U1[t] = Y16[t][0] + Y16[t][1];
            U2[t] = Y16[t][1] - Y16[t][0];
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                U0[t][c] = row_from_left(hl[c], Y16[t][1][c]) - Y16[t][1][c];
                U3[t][c] = Y16[t][0][c] - row_from_right(hr[c], Y16[t][0][c]);
            }
        }
        // the caller's requests for stage F (conv17's fragments, the next group's offsets): this
        // wave has nothing in flight and waits for nothing of global memory before the closing
        // barrier, which retires them
#pragma unroll
        for (int t = 0; t < 3; ++t) asm volatile("" : "+v"(U0[t]), "+v"(U1[t]), "+v"(U2[t]), "+v"(U3[t]));
#if !DBH_F_AHEAD_EARLY
        ahead();
#endif
        // The fragments of a third of the contraction at a time (four 16-byte reads: twelve
        // registers' worth less to hold beside this lane's 48 outputs), asked for ONE THIRD AHEAD
        // from inline asm and waited for by count: as plain loads hipcc put each read and an
        // lgkmcnt(0) right in front of the two to six MFMAs that use it - 36 exposed LDS round
        // trips per wave (profiles/r06_steps).
        const unsigned w16_addr = lds_addr(lds + kEW16 + lane * 4);
        f4 wbuf[2][2][2];      // [parity][s2][pr]
        auto ask = [&](auto it_tag) {
            constexpr int IT = decltype(it_tag)::value;      // t * 3 + third_k
            constexpr int T16 = IT / 3, TK = IT % 3;
            wbuf[IT & 1][0][0] = ds_read_f4<(((T16 * 6 + 2 * TK + 0) * 2 + 0) * 256) * 4>(w16_addr);
            wbuf[IT & 1][0][1] = ds_read_f4<(((T16 * 6 + 2 * TK + 0) * 2 + 1) * 256) * 4>(w16_addr);
            wbuf[IT & 1][1][0] = ds_read_f4<(((T16 * 6 + 2 * TK + 1) * 2 + 0) * 256) * 4>(w16_addr);
            wbuf[IT & 1][1][1] = ds_read_f4<(((T16 * 6 + 2 * TK + 1) * 2 + 1) * 256) * 4>(w16_addr);
        };
        f4 M0, M1, M2, M3;
        f4 b16 = tab4[(bias_offset(15) - TB) / 4];
        auto third_of = [&](auto it_tag, auto&& self) -> void {
            constexpr int IT = decltype(it_tag)::value;
            constexpr int T16 = IT / 3, TK = IT % 3;
            if constexpr (IT + 1 < 9) {
                ask(IntC<IT + 1>{});
                asm volatile("s_waitcnt lgkmcnt(4)" ::: "memory");
            } else {
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            }
            f4(&wf)[2][2] = wbuf[IT & 1];
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2) asm volatile("" : "+v"(wf[s2][0]), "+v"(wf[s2][1]));
            const f4 zero4 = f4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
                for (int e = 0; e < 2; ++e) {
                    // k-step 2 sp + e <-> channels 16 tc + 4q + r, sp = 2 third_k + s2
                    constexpr int tc = TK;
                    const int r = 2 * s2 + e;
                    const bool first = TK == 0 && s2 == 0 && e == 0;
                    M0 = mfma4(wf[s2][0][e], U0[tc][r], first ? b16 : M0);
                    M1 = mfma4(wf[s2][0][2 + e], U1[tc][r], first ? zero4 : M1);
                    M2 = mfma4(wf[s2][1][e], U2[tc][r], first ? zero4 : M2);
                    M3 = mfma4(wf[s2][1][2 + e], U3[tc][r], first ? -b16 : M3);
                }
            if constexpr (TK == 2) {
                pooled_out(OUT[3][T16], 144 + 16 * T16, relu4(M0 + M1 + M2), relu4(M1 - M2 - M3));
                if constexpr (T16 + 1 < 3) b16 = tab4[(bias_offset(15) - TB) / 4 + 4 * (T16 + 1)];
            }
            if constexpr (IT + 1 < 9) self(IntC<IT + 1>{}, self);
        };
        ask(IntC<0>{});
        third_of(IntC<0>{}, third_of);
    }
    __builtin_amdgcn_sched_barrier(0);
    mark(ts, 40);
    // every wave is through with the block's weights (and stage D's slots): the concat images go
    // over them - row 1 + j, channels 48 b + 16 t + 4q + r - with their two zero rows (conv1d_17's
    // 'same' padding); cat_base < 0: a window the group does not have
    lds_barrier();
    if (cat_base >= 0) {
        lds_float* cat = lds_pinned(lds + cat_base + (1 + j) * kS192 + 4 * q);
#pragma unroll
        for (int br = 0; br < 4; ++br)
#pragma unroll
            for (int t = 0; t < 3; ++t)
                *reinterpret_cast<__attribute__((address_space(3))) f4*>(cat + 48 * br + 16 * t) = OUT[br][t];
        if (lane < 49) {
            f4* row = reinterpret_cast<f4*>(lds + cat_base + (hf ? 33 * kS192 : 0));
            float zero = 0.f;
            asm volatile("" : "+v"(zero));      // (made here: hoisted out of the persistent loop, four registers of zeros were spilled)
            row[lane] = f4{zero, zero, zero, zero};
        }
    }
    // the images are out (and what stage F's first window needs of global memory has landed)
    full_barrier();
    mark(ts, 33);
}

struct NoBetween {
    __device__ __forceinline__ void operator()(int) const {}
};

// ---------------------------------------------------------------------------------------------
// F(4,3) at L = 256 (conv7 + MaxPool + BN): 64 quads = four tiles of 16 for eight waves, 864 MFMAs
// instead of F(2,3)'s 1,152.  The two waves of a SIMD, w and w + 4, share tile w & 3 and split its
// OUTPUT CHANNELS: w takes N tile 0 and the first three channel groups of N tile 1's sum, w + 4
// N tile 2 and the last three - 108 MFMAs each.  Both build all of U (72 VGPRs) while they
// multiply for their own N tile (six steps of twelve MFMAs); behind the mid-layer barrier (every
// input row read: outputs may go in place) come the three steps of the shared N tile, with the
// own tile's epilogue inside them.  The output transform is linear, so each wave applies it to its
// partial sums of the shared tile and only two of the four outputs per quad - two f4 per lane -
// cross LDS each way, announced on a per-pair counter word: w finishes outputs 0,1 (the first
// pooled position of a quad), w + 4 outputs 2,3 (the second).
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ void pair_signal(float* lds, int word, int lane) {
    // (plain LDS instructions: requests of one wave are served in order, so the counter add lands
    // after the data stored before it; a C++ release atomic would also wait for the wave's global
    // prefetches)
    const unsigned addr = lds_addr(lds + kPairSync + word);
    if (lane == 0) asm volatile("ds_add_u32 %0, %1" ::"v"(addr), "v"(1u) : "memory");
}
__device__ __forceinline__ void pair_wait(float* lds, int word, unsigned target) {
    const unsigned addr = lds_addr(lds + kPairSync + word);
    for (;;) {
        unsigned seen;
        asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(seen) : "v"(addr) : "memory");
        if ((int)(__builtin_amdgcn_readfirstlane(seen) - target) >= 0) break;
        __builtin_amdgcn_s_sleep(1);
    }
}

struct W43nsPipe {
    f2 rows[2][6];   // [step parity][input row]
    f4 b[2][3];      // [step parity][matrix pair]
};

// How the six channel groups of conv7's shared N tile are split between the two waves of a SIMD:
// the older one (w < 4) takes the first DBH_CONV7_LOW, the younger one the rest.  (3 / 3 until
// round 5; the older wave wins every tie for the matrix pipe and reached the exchange first.)
#ifndef DBH_CONV7_LOW
#define DBH_CONV7_LOW 3
#endif
// Step G of a wave's 6 + n: G < 6 = channel group G of its own N tile (TOWN) and of U; G >= 6 =
// channel group SP0 + G - 6 of the shared N tile 1.  One software pipeline.
template <int TOWN, int SP0, bool WITH_BIAS, int G, class Side>
__device__ __forceinline__ void w43ns_step(W43U& U, unsigned a_addr, unsigned b_addr,
                                           W43nsPipe& pipe, f4 (&own)[6], f4 (&shared)[6],
                                           float bias_own, float bias_shared, const Side& side) {
    auto loads = [&](auto step_tag) {
        constexpr int N = decltype(step_tag)::value;
        if constexpr (N < 6) w43_load_rows<N>(pipe.rows[N & 1], a_addr);
        constexpr int T = N < 6 ? TOWN : 1, SP = N < 6 ? N : SP0 + N - 6;
        pipe.b[N & 1][0] = ds_read_f4<(T * kWinoHalf + (SP * 3 + 0) * 256) * 4>(b_addr);
        pipe.b[N & 1][1] = ds_read_f4<(T * kWinoHalf + (SP * 3 + 1) * 256) * 4>(b_addr);
        pipe.b[N & 1][2] = ds_read_f4<(T * kWinoHalf + (SP * 3 + 2) * 256) * 4>(b_addr);
    };
    constexpr int GEND = 6 + (SP0 == 0 ? DBH_CONV7_LOW : 6 - DBH_CONV7_LOW);
    if constexpr (G == 0) loads(IntC<0>{});
    if constexpr (G + 1 < GEND) {
        loads(IntC<G + 1>{});
        if constexpr (G + 1 < 6) asm volatile("s_waitcnt lgkmcnt(9)" ::: "memory");
        else asm volatile("s_waitcnt lgkmcnt(3)" ::: "memory");
    } else {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    f4(&b)[3] = pipe.b[G & 1];
#pragma unroll
    for (int p = 0; p < 3; ++p) asm volatile("" : "+v"(b[p]));
    if constexpr (G < 6) progress_priority<G, 6>();
    else progress_priority<G - 6, GEND - 6>();
    constexpr int SP = G < 6 ? G : SP0 + G - 6;      // which channel group of U this step uses
    if constexpr (G < 6) {
        f2(&d)[6] = pipe.rows[G & 1];
#pragma unroll
        for (int k = 0; k < 6; ++k) asm volatile("" : "+v"(d[k]));
        __builtin_amdgcn_sched_barrier(0);
        w43_transform<G>(U, d);
    }
    __builtin_amdgcn_sched_barrier(0);
    f4(&acc)[6] = G < 6 ? own : shared;
    if constexpr (G == 0 || G == 6) {
        // the chains start here: five from the MFMA's constant 0, M1's from the bias (every output
        // of the transform takes M1 with weight 1) - for the shared tile in ONE of the two waves
        const f4 zero = f4{0.f, 0.f, 0.f, 0.f};
        const float bv = G == 0 ? bias_own : bias_shared;
        const bool with_bias = G == 0 || WITH_BIAS;
#pragma unroll
        for (int p = 0; p < 3; ++p) {
            acc[2 * p] = mfma4(U.u[2 * p][SP].x, b[p][0], zero);
            acc[2 * p + 1] = mfma4(U.u[2 * p + 1][SP].x, b[p][2],
                                   (p == 0 && with_bias) ? f4{bv, bv, bv, bv} : zero);
        }
    } else {
#pragma unroll
        for (int p = 0; p < 3; ++p) {
            acc[2 * p] = mfma4(U.u[2 * p][SP].x, b[p][0], acc[2 * p]);
            acc[2 * p + 1] = mfma4(U.u[2 * p + 1][SP].x, b[p][2], acc[2 * p + 1]);
        }
    }
#pragma unroll
    for (int p = 0; p < 3; ++p) {
        acc[2 * p] = mfma4(U.u[2 * p][SP].y, b[p][1], acc[2 * p]);
        acc[2 * p + 1] = mfma4(U.u[2 * p + 1][SP].y, b[p][3], acc[2 * p + 1]);
    }
#pragma unroll
    for (int x = 0; x < 6; ++x) asm volatile("" : "+v"(acc[x]));
    __builtin_amdgcn_sched_barrier(0);
    side(IntC<G>{});
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (G != 5 && G + 1 < GEND)
        w43ns_step<TOWN, SP0, WITH_BIAS, G + 1>(U, a_addr, b_addr, pipe, own, shared, bias_own,
                                                bias_shared, side);
}

// rows 2h, 2h+1 of partial accumulators -> the partial outputs (no ReLU: they are partial sums)
__device__ __forceinline__ void w43_partial_outputs(const f4 (&acc)[6], int h, f2 (&y)[4]) {
    const f2 k2 = f2{2.f, 2.f}, k4 = f2{4.f, 4.f}, k8 = f2{8.f, 8.f};
    const f2 a0 = f2{acc[0][2 * h], acc[0][2 * h + 1]}, a1 = f2{acc[1][2 * h], acc[1][2 * h + 1]};
    const f2 a2 = f2{acc[2][2 * h], acc[2][2 * h + 1]}, a3 = f2{acc[3][2 * h], acc[3][2 * h + 1]};
    const f2 a4 = f2{acc[4][2 * h], acc[4][2 * h + 1]}, a5 = f2{acc[5][2 * h], acc[5][2 * h + 1]};
    const f2 s12 = a1 + a2, d12 = a1 - a2, s34 = a3 + a4, d34 = a3 - a4;
    y[0] = a0 + s12 + s34;
    y[1] = __builtin_elementwise_fma(k2, d34, d12);
    y[2] = __builtin_elementwise_fma(k4, s34, s12);
    y[3] = __builtin_elementwise_fma(k8, d34, d12) + a5;
}

// One wave's half (HIGH = wave >= 4) of conv7.  side(G): the caller's LDS-DMA requests behind the
// MFMAs of step G (0..5: before the mid-layer barrier; 6..8: behind it, when N tiles 0 and 2 of the
// layer's own weights - slots 0 and 2 - are free).
//   LAST = false: park = this window's park in global memory; the layer ends with an LDS-only
// barrier - its stores, and the caller's requests, are still on their way: the caller retires them
// (a full barrier) before anything reads what they bring.
//   LAST = true (the group's last window, which stage D follows at once): the output goes to the
// same layout in LDS (kPark7Lds: rows of the input image every wave has read by the mid-layer
// barrier) and the layer ends with a full barrier, which then has nothing slow to wait for.
//   after(): the caller's requests behind the layer's last MFMAs (U and the pipeline are dead: ~120
// free registers); returns how many of them - the wave's newest - may still be in flight behind
// the LAST form's closing barrier.
template <int CONV, int BNI, bool HIGH, bool LAST, class Side, class After>
__device__ __forceinline__ void w43_nsplit_half(float* lds, const float* __restrict__ packed,
                                                float* __restrict__ park, int tid, int lane,
                                                int wave, unsigned* ts, int ts_base,
                                                unsigned& pair_rounds, const Side& side,
                                                const After& after) {
    constexpr int TOWN = HIGH ? 2 : 0, SP0 = HIGH ? DBH_CONV7_LOW : 0;
    const int n = lane & 15, q = lane >> 4;
    const int m = wave & 3;
    EpiParams<3, true> ep;
    load_epi<CONV, BNI>(ep, lds, packed, n);
    // quad j = m*16 + pm(n) needs logical rows 4j-1 .. 4j+4 = physical rows 4j .. 4j+5
    const int pm_n = 2 * (n >> 2) + (n & 1) + 8 * ((n >> 1) & 1);
    const unsigned a_addr = lds_addr(lds + kActOff + (m * 64 + 4 * pm_n) * kS48 + 2 * q);
    const unsigned b_addr = lds_addr(lds + kSlot0 + lane * 4);
    // this lane's place in the park: pooled row 0 of quad m*16 + 2q, channel n of N tile 0
    const int park_off = (m >> 1) * 3072 + ((n >> 2) * 16 + (m & 1) * 8 + q) * 4 + (n & 3);
    typedef typename std::conditional<LAST, lds_float*, float*>::type ParkPtr;
    ParkPtr park_lane;
    if constexpr (LAST) park_lane = lds_pinned(lds + kPark7Lds + park_off);
    else park_lane = park + park_off;
    (void)tid;
    float* mine = lds + kX7 + wave * 512 + lane * 4;
    const float* theirs = lds + kX7 + (wave ^ 4) * 512 + lane * 4;
    W43U U;
    W43nsPipe pipe;
    f4 own[6], shared[6];
    w43ns_step<TOWN, SP0, !HIGH, 0>(U, a_addr, b_addr, pipe, own, shared, ep.b[TOWN], ep.b[1], side);
    mark(ts, ts_base);
    // every wave has read all its input rows, and is through with N tiles 0 and 2 of the weights:
    // the exchange below may use those rows, side(6..8) those slots.  The barrier also RETIRES what
    // side(0..5) asked for (vmcnt(0): this wave has nothing else in flight - its park stores of the
    // window before were waited for long ago, this window's come behind this barrier - and the
    // requests are some thousand cycles old): what they bring may be read behind the layer's closing
    // barrier without another wait.
    full_barrier();
    mark(ts, ts_base + 1);
    w43ns_step<TOWN, SP0, !HIGH, 6>(
        U, a_addr, b_addr, pipe, own, shared, ep.b[TOWN], ep.b[1], [&](auto tag) {
            constexpr int G = decltype(tag)::value;
            side(tag);
            if constexpr (G == 6) w43_epilogue_half_park<TOWN, ParkPtr>(own, 0, ep.sc[TOWN], ep.sh[TOWN], park_lane);
            if constexpr (G == 7) w43_epilogue_half_park<TOWN, ParkPtr>(own, 1, ep.sc[TOWN], ep.sh[TOWN], park_lane);
        });
    const bool in_flight = after();
    // the shared N tile: the partial outputs of this wave's three channel groups
    f2 y[2][4];
    w43_partial_outputs(shared, 0, y[0]);
    w43_partial_outputs(shared, 1, y[1]);
    constexpr int kKeep = HIGH ? 2 : 0, kShip = HIGH ? 0 : 2;
    // the two outputs of each quad the partner finishes: rows 0..3 of output kShip, then kShip + 1
    *reinterpret_cast<f4*>(mine) = f4{y[0][kShip].x, y[0][kShip].y, y[1][kShip].x, y[1][kShip].y};
    *reinterpret_cast<f4*>(mine + 256) =
        f4{y[0][kShip + 1].x, y[0][kShip + 1].y, y[1][kShip + 1].x, y[1][kShip + 1].y};
    pair_signal(lds, 2 * m + (HIGH ? 1 : 0), lane);
    if constexpr (!LAST) mark(ts, ts_base + 2);
    pair_wait(lds, 2 * m + (HIGH ? 0 : 1), pair_rounds + 1);
    pair_rounds += 1;
    const f4 t0 = *reinterpret_cast<const f4*>(theirs);
    const f4 t1 = *reinterpret_cast<const f4*>(theirs + 256);
    const float sc = ep.sc[1], sh = ep.sh[1];
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            const float ya = y[h][kKeep][e] + t0[2 * h + e];
            const float yb = y[h][kKeep + 1][e] + t1[2 * h + e];
            float p = fmaxf(fmaxf(ya, yb), 0.f);       // ReLU, then MaxPool over the two positions
            p = fmaf(p, sc, sh);
            // quad offset e + 8h from the lane's first (pm(4q + 2h + e)); pooled row 2e + (HIGH ? 1 : 0)
            park_lane[(2 * e + (HIGH ? 1 : 0)) * 768 + 256 + h * 16] = p;
        }
    if constexpr (LAST) {
        // (the weights stage D's first tiles read have landed: requested before after()'s twelve
        // loads, and loads return in order; this form has no stores in flight)
        (void)in_flight;       // (what was asked for before the mid-layer barrier was retired there)
        lds_barrier();
    } else {
        lds_barrier();
    }
    mark(ts, LAST ? ts_base + 2 : ts_base + 3);      // (the LAST form is stamped 59, 60, 61)
}

struct NoAfter {
    __device__ __forceinline__ bool operator()() const { return false; }
};
template <int CONV, int BNI, bool LAST, class Side, class After = NoAfter>
__device__ __forceinline__ void w43_nsplit_pooled_layer(float* lds, const float* __restrict__ packed,
                                                        float* __restrict__ park, int tid, int lane,
                                                        int wave, unsigned* ts, int ts_base,
                                                        unsigned& pair_rounds, const Side& side,
                                                        const After& after = After()) {
    static_assert(kConv[CONV].wino == 4 && kConv[CONV].cin == 48 && kConv[CONV].cout_pad == 48, "");
    if (wave < 4)
        w43_nsplit_half<CONV, BNI, false, LAST>(lds, packed, park, tid, lane, wave, ts, ts_base, pair_rounds, side, after);
    else
        w43_nsplit_half<CONV, BNI, true, LAST>(lds, packed, park, tid, lane, wave, ts, ts_base, pair_rounds, side, after);
}

// ---------------------------------------------------------------------------------------------
// A small-M layer (one 16-position tile: conv17/18/19).  Each weight is used once per window,
// so B fragments skip LDS: every wave fetches its share from L2 into registers well AHEAD of
// use (SmallMRegs::prefetch).  A wave owns NTW of the 3 N tiles and 1/KS of the contraction:
//   conv17 (K = 576): KS = 8, NTW = 3 -> all 8 waves, 54 MFMAs each, partial tiles summed via LDS;
//   conv18/19 (K = 144): KS = 1, NTW = 1 -> 3 waves, 36 MFMAs each, no reduction phase at all.
// ---------------------------------------------------------------------------------------------
template <int CONV, int KS, int NTW, bool BN>
struct SmallMRegs {
    static constexpr int TAPS = kConv[CONV].taps;
    static constexpr int SPTOT = kConv[CONV].cin / 8;
    static constexpr int SP = SPTOT / KS;
    static constexpr int NGROUPS = 3 / NTW;             // wave groups along N
    static constexpr int ACTIVE = KS * NGROUPS;
    static_assert(SP * KS == SPTOT && NGROUPS * NTW == 3 && ACTIVE <= kWaves, "bad split");
    f2 b[TAPS * SP * NTW];
    EpiParams<1, BN> ep;
    __device__ __forceinline__ void prefetch(const float* __restrict__ packed, int bn_index,
                                             int lane, int wave) {
        if (ACTIVE == kWaves || wave < ACTIVE) {
            const int t0 = (wave % NGROUPS) * NTW, ks = wave / NGROUPS;
            const float* b_lane = packed + weight_offset(CONV) + (ks * SP * 3 + t0) * 128 + lane * 2;
#pragma unroll
            for (int tap = 0; tap < TAPS; ++tap)
#pragma unroll
                for (int sp = 0; sp < SP; ++sp)
#pragma unroll
                    for (int t = 0; t < NTW; ++t)
                        b[(tap * SP + sp) * NTW + t] = *reinterpret_cast<const f2*>(
                            b_lane + ((tap * SPTOT + sp) * 3 + t) * 128);
        }
        prefetch_epilogue(packed, bn_index, lane, wave);
    }
    __device__ __forceinline__ void prefetch_epilogue(const float* __restrict__ packed,
                                                      int bn_index, int lane, int wave) {
        if (wave < 3) {     // the waves that run the epilogue (one N tile each)
            const int ch = wave * 16 + (lane & 15);
            ep.load(packed + bias_offset(CONV) + ch,
                    packed + (BN ? bn_scale_offset(bn_index) : 0) + ch,
                    packed + (BN ? bn_shift_offset(bn_index) : 0) + ch);
        }
    }
    // fragments [K0, K1) only - for trickling the fetch across the steps of an earlier layer
    template <int K0, int K1>
    __device__ __forceinline__ void prefetch_slice(const float* __restrict__ packed, int lane,
                                                   int wave) {
        if (ACTIVE == kWaves || wave < ACTIVE) {
            const int t0 = (wave % NGROUPS) * NTW, ks = wave / NGROUPS;
            // buffer loads: the wave's part of the address is scalar, the lane's one shift
            const __amdgpu_buffer_rsrc_t view = buffer_view(packed + weight_offset(CONV));
            const unsigned wave_bytes = (unsigned)((ks * SP * 3 + t0) * 128) * 4u;
            const unsigned lane_bytes = (unsigned)lane * 8u;
#pragma unroll
            for (int k = K0; k < K1; ++k) {
                const int t = k % NTW, sp = (k / NTW) % SP, tap = k / (NTW * SP);
                b[k] = buffer_load_f2(view, lane_bytes,
                                      wave_bytes + (unsigned)(((tap * SPTOT + sp) * 3 + t) * 128) * 4u);
            }
        }
    }
};

// TO_GLOBAL: out_region is a dense [16][48] block in global memory (row 0 = position 0) instead
// of an LDS activation buffer, and the layer ends without a barrier of its own.
// pre_barrier / post_barrier: work of the caller's that rides on the layer's one barrier (every
// wave calls pre_barrier before it, post_barrier after it).
// between(tap): a request of the caller's behind the MFMAs of tap `tap` (LDS-DMA pieces: one at a
// time between MFMAs instead of a bunch in front of them).
// all_waves (TO_GLOBAL only): the partial tiles are summed by ALL eight waves, 24 of the 192 output
// quadruples each (ep_all: bias / BN of this lane's quadruple: small_m_all_params), instead of by
// waves 0-2 while the others wait for them at the next barrier.
template <int CONV, int S_IN, int STRIDE, int KS, int NTW, bool POOL, bool BN, bool TO_GLOBAL = false,
          class PreBarrier = NoHook, class PostBarrier = NoHook, class Between = NoBetween>
__device__ __forceinline__ void small_m_layer(float* lds, const float* in_region, float* out_region,
                                              float* red, const SmallMRegs<CONV, KS, NTW, BN>& regs, int lane,
                                              int wave, unsigned* ts, int ts_base,
                                              const PreBarrier& pre_barrier = PreBarrier(),
                                              const PostBarrier& post_barrier = PostBarrier(),
                                              const Between& between = Between(),
                                              const EpiParams<1, BN>* ep_all = nullptr) {
    using R = SmallMRegs<CONV, KS, NTW, BN>;
    constexpr int TAPS = R::TAPS, SP = R::SP;
    const int n = lane & 15, q = lane >> 4;
    f4 acc[1][NTW];
    zero_acc(acc);
    if (wave < R::ACTIVE) {
        const int ks = wave / R::NGROUPS;
        // stride-2 'same' pads on the right only: logical row 2p+tap = physical row 2p+tap+1;
        // stride-1 'same' k=3: physical row p+tap.
        const int first = (STRIDE == 2) ? 1 : 0;
        const float* a_lane = in_region + (first + n * STRIDE) * S_IN + 2 * q + ks * SP * 8;
        // even/odd k-steps accumulate separately so consecutive MFMAs never wait on each other
        f4 odd[NTW];
#pragma unroll
        for (int t = 0; t < NTW; ++t) odd[t] = f4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int tap = 0; tap < TAPS; ++tap) {
            f2 a[SP];
#pragma unroll
            for (int sp = 0; sp < SP; ++sp)
                a[sp] = *reinterpret_cast<const f2*>(a_lane + tap * S_IN + sp * 8);
#pragma unroll
            for (int sp = 0; sp < SP; ++sp)
#pragma unroll
                for (int t = 0; t < NTW; ++t) {
                    const f2 bw = regs.b[(tap * SP + sp) * NTW + t];
                    acc[0][t] = mfma4(a[sp].x, bw.x, acc[0][t]);
                    odd[t] = mfma4(a[sp].y, bw.y, odd[t]);
                }
            between(tap);
        }
#pragma unroll
        for (int t = 0; t < NTW; ++t) acc[0][t] += odd[t];
    }
    if constexpr (KS > 1) {
        static_assert(NTW == 3, "split-K path assumes every wave holds all three N tiles");
        if (wave < R::ACTIVE) {
#pragma unroll
            for (int t = 0; t < 3; ++t)
                *reinterpret_cast<f4*>(red + (wave * 3 + t) * 256 + lane * 4) = acc[0][t];
        }
        pre_barrier();
        mark(ts, ts_base);
        lds_barrier();
        mark(ts, ts_base + 1);
        post_barrier();
        if (ep_all != nullptr) {
            if constexpr (TO_GLOBAL) {
                if (lane < 24) {
                    const int e = wave * 24 + lane, t = e >> 6, l2 = e & 63;
                    f4 sum[1][1];
                    sum[0][0] = *reinterpret_cast<const f4*>(red + t * 256 + l2 * 4);
#pragma unroll
                    for (int ks = 1; ks < KS; ++ks)
                        sum[0][0] += *reinterpret_cast<const f4*>(red + (ks * 3 + t) * 256 + l2 * 4);
                    epilogue<1, 1, 48, false, BN>(sum, out_region + 4 * (l2 >> 4) * 48 + t * 16 + (l2 & 15), *ep_all);
                }
            }
        } else if (wave < 3) {
            const int t = wave;
            f4 sum[1][1];
            sum[0][0] = *reinterpret_cast<const f4*>(red + t * 256 + lane * 4);
#pragma unroll
            for (int ks = 1; ks < KS; ++ks)
                sum[0][0] +=
                    *reinterpret_cast<const f4*>(red + (ks * 3 + t) * 256 + lane * 4);
            if constexpr (TO_GLOBAL) {
                static_assert(!POOL, "");
                epilogue<1, 1, 48, false, BN>(sum, out_region + 4 * q * 48 + t * 16 + n, regs.ep);
            } else {
                float* out_lane = out_region + (1 + (POOL ? 2 * q : 4 * q)) * kS48 + t * 16 + n;
                epilogue<1, 1, kS48, POOL, BN>(sum, out_lane, regs.ep);
            }
        }
    } else {
        static_assert(NTW == 1, "direct path: one N tile per wave");
        mark(ts, ts_base);
        mark(ts, ts_base + 1);
        if (wave < 3) {
            float* out_lane = out_region + (1 + (POOL ? 2 * q : 4 * q)) * kS48 + wave * 16 + n;
            epilogue<1, 1, kS48, POOL, BN>(acc, out_lane, regs.ep);
        }
    }
    mark(ts, ts_base + 2);
    if constexpr (!TO_GLOBAL) full_barrier();  
    mark(ts, ts_base + 3);
}

// One wave's share of the 1x1 convolutions of the inception block (4 position tiles x 1 N tile).
// ep: bias (and BN5 scale / shift) of this lane's channel, loaded by the caller ahead of the call
template <int NTTOT, int S_OUT, bool POOLBN>
__device__ __forceinline__ void inception_1x1(const float* in_region, const float* w_lds,
                                              float* out_region, int out_ch,
                                              const EpiParams<1, POOLBN>& ep, int t, int lane) {
    const int n = lane & 15, q = lane >> 4;
    f4 acc[4][1];
    bias_acc(acc, ep);
    conv_tiles<1, 6, 6, 4, 1, NTTOT, kS48, 16>(in_region + (n + 1) * kS48 + 2 * q,
                                               w_lds + t * 128 + lane * 2, acc);
    float* out_lane = out_region + (1 + (POOLBN ? 2 * q : 4 * q)) * S_OUT + out_ch + n;
    epilogue<4, 1, S_OUT, POOLBN, POOLBN, false>(acc, out_lane, ep);
}

// conv10 reads AveragePooling1D(3, stride 1, 'same') of X.  A 1x1 convolution commutes with a
// pooling along the positions: W . (x[p-1] + x[p] + x[p+1]) / c[p] = (z[p-1] + z[p] + z[p+1]) / c[p]
// with z = W . x (no bias, z = 0 outside the window; c[p] = the number of taps inside it:
// TensorFlow's valid-count divisor, oracle/network_ref.py).  So the convolution runs on X itself
// and the pooling on its OUTPUT, in registers: the wave holds all 64 positions of its 16 channels
// (lane (n, q), tile m, register r <-> position 16 m + 4 q + r), the neighbours across the lane
// groups come by ds_bpermute.  No average-pooled copy of X in LDS, no phase of its own (it cost
// ~1.7k cycles per window with its barrier: profiles/r03_v1/timeline_5120_fused.txt, "E0").
// Then bias, ReLU, MaxPool2, BN5 as everywhere.
template <int NTTOT, int S_OUT>
__device__ __forceinline__ void inception_1x1_of_avgpool(const float* in_region, const float* w_lds,
                                                         float* out_region, int out_ch,
                                                         const EpiParams<1, true>& ep, int t,
                                                         int lane) {
    const int n = lane & 15, q = lane >> 4;
    f4 z[4][1];
    zero_acc(z);
    conv_tiles<1, 6, 6, 4, 1, NTTOT, kS48, 16>(in_region + (n + 1) * kS48 + 2 * q,
                                               w_lds + t * 128 + lane * 2, z);
    // position 4q - 1 lives in register 3 of the lane group before (the tile before, for q = 0),
    // position 4q + 4 in register 0 of the lane group behind
    const int from_before = ((lane + 48) & 63) * 4, from_behind = ((lane + 16) & 63) * 4;
    float before[4], behind[4];
#pragma unroll
    for (int m = 0; m < 4; ++m) {
        // (pinned copies: handed z[m][0].w directly, this hipcc sends register x both times)
        float last = z[m][0].w, first = z[m][0].x;
        asm volatile("" : "+v"(last), "+v"(first));
        before[m] = __builtin_bit_cast(
            float, __builtin_amdgcn_ds_bpermute(from_before, __builtin_bit_cast(int, last)));
        behind[m] = __builtin_bit_cast(
            float, __builtin_amdgcn_ds_bpermute(from_behind, __builtin_bit_cast(int, first)));
    }
    const float third = 1.f / 3.f, b = ep.b[0], sc = ep.sc[0], sh = ep.sh[0];
    float* out_lane = out_region + (1 + 2 * q) * S_OUT + out_ch + n;
#pragma unroll
    for (int m = 0; m < 4; ++m) {
        const float prev = q > 0 ? before[m] : (m > 0 ? before[m > 0 ? m - 1 : 0] : 0.f);
        const float next = q < 3 ? behind[m] : (m < 3 ? behind[m < 3 ? m + 1 : 3] : 0.f);
        const f4 v = z[m][0];
        const float t1 = v.x + v.y, t2 = v.z + v.w;
        // x (1 / count), not / count: one ulp of the quotient is far inside the tolerance
        const float inv0 = (m == 0 && q == 0) ? 0.5f : third;
        const float inv3 = (m == 3 && q == 3) ? 0.5f : third;
        const float y0 = fmaxf(fmaf(prev + t1, inv0, b), 0.f);
        const float y1 = fmaxf(fmaf(t1 + v.z, third, b), 0.f);
        const float y2 = fmaxf(fmaf(v.y + t2, third, b), 0.f);
        const float y3 = fmaxf(fmaf(t2 + next, inv3, b), 0.f);
        out_lane[(m * 8 + 0) * S_OUT] = fmaf(fmaxf(y0, y1), sc, sh);
        out_lane[(m * 8 + 1) * S_OUT] = fmaf(fmaxf(y2, y3), sc, sh);
    }
}

// The 16 -> 48, k = 3 convolutions of the inception block (conv13, conv15; L = 64) as Winograd
// F(2,3): pair tile m (16 pairs = 32 positions) x NT channel tiles from tile T0, 16 MFMAs per
// (pair tile, channel tile) instead of the direct form's 24.  POOLBN: the pair's two outputs are
// max-pooled (that IS MaxPool2) and batch-normalised into the concat buffer; otherwise both are
// stored, ReLU'd.  Pointers are for channel tile T0.
template <int NT, int S_OUT, bool POOLBN>
__device__ __forceinline__ void inception_k3_wino(const float* in_region, const float* w_tile0,
                                                  float* out_region, int out_ch,
                                                  const float* __restrict__ bias_lane,
                                                  const float* __restrict__ scale_lane,
                                                  const float* __restrict__ shift_lane, int m,
                                                  int lane) {
    const int n = lane & 15, q = lane >> 4;
    EpiParams<NT, POOLBN> ep;
    ep.load(bias_lane, scale_lane, shift_lane);
    float bias[3] = {ep.b[0], NT > 1 ? ep.b[NT > 1 ? 1 : 0] : 0.f, NT > 2 ? ep.b[NT > 2 ? 2 : 0] : 0.f};
    // pair j = m*16 + n needs logical rows 2j-1 .. 2j+2 = physical rows 2j .. 2j+3
    const unsigned a_addr = lds_addr(in_region + (m * 32 + 2 * n) * kS16 + 2 * q);
    const unsigned b_addr = lds_addr(w_tile0 + lane * 4);
    W23U16 U;
    W23Pipe16 pipe;
    f4 acc[3][4];
    w23c16_step<0, 2 * NT>(U, a_addr, b_addr, pipe, acc, bias, NoSide());
    // pair m*16 + 4q + r of the window -> pooled position (same number) or positions 2j, 2j + 1
    float* out_lane = out_region + (1 + (POOLBN ? 1 : 2) * (m * 16 + 4 * q)) * S_OUT + out_ch + n;
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float even = acc[t][0][r] + acc[t][1][r] + acc[t][2][r];
            const float odd = acc[t][1][r] - acc[t][2][r] - acc[t][3][r];
            if constexpr (POOLBN) {
                const float o = fmaxf(fmaxf(even, odd), 0.f);
                out_lane[r * S_OUT + t * 16] = fmaf(o, ep.sc[t], ep.sh[t]);
            } else {
                out_lane[(2 * r) * S_OUT + t * 16] = fmaxf(even, 0.f);
                out_lane[(2 * r + 1) * S_OUT + t * 16] = fmaxf(odd, 0.f);
            }
        }
}

// k=3 convolution of the inception block: MT position tiles from tile m0 x NT channel tiles
// from tile t (pointers are for channel tile t; NT > 1 walks on in steps of 16 channels).
template <int SP, int MT, int NT, int S_IN, int S_OUT, bool POOLBN>
__device__ __forceinline__ void inception_k3(const float* in_region, const float* w_lds,
                                             float* out_region, int out_ch,
                                             const float* __restrict__ bias_lane,
                                             const float* __restrict__ scale_lane,
                                             const float* __restrict__ shift_lane, int t, int m0,
                                             int lane) {
    const int n = lane & 15, q = lane >> 4;
    EpiParams<NT, POOLBN> ep;
    ep.load(bias_lane, scale_lane, shift_lane);
    f4 acc[MT][NT];
    bias_acc(acc, ep);
    conv_tiles<3, SP, SP, MT, NT, 3, S_IN, 16>(in_region + (m0 * 16 + n) * S_IN + 2 * q,
                                               w_lds + t * 128 + lane * 2, acc);
    float* out_lane = out_region +
                      (1 + (POOLBN ? m0 * 8 + 2 * q : m0 * 16 + 4 * q)) * S_OUT + out_ch + n;
    epilogue<MT, NT, S_OUT, POOLBN, POOLBN, false>(acc, out_lane, ep);
}

// ---------------------------------------------------------------------------------------------
// make_sum_to_one + barcode call for one read held by a 32-lane group (lane c = class c):
// classify.py:387-393 in fp64 (what NumPy-1.x scalar promotion gave the reference) and
// classify.py:285-295 (ties to the lower class index: Python's stable sort with reverse=True).
// Shared by the stand-alone merge kernel and the forward kernel's fused single-step finish.
// ---------------------------------------------------------------------------------------------
// 64-bit / index moves inside a 16-lane row (DPP, no LDS round trip) for the reductions below.
template <int CTRL>
__device__ __forceinline__ int dpp_move_i32(int v) {
    return __builtin_amdgcn_update_dpp(0, v, CTRL, 0xF, 0xF, true);
}
template <int CTRL>
__device__ __forceinline__ double dpp_move_f64(double v) {
    const long long b = __builtin_bit_cast(long long, v);
    const unsigned lo = (unsigned)dpp_move_i32<CTRL>((int)b);
    const unsigned hi = (unsigned)dpp_move_i32<CTRL>((int)(b >> 32));
    return __builtin_bit_cast(double, (long long)(((unsigned long long)hi << 32) | lo));
}

// The 32 lanes c = 0..31 of one read (two 16-lane rows) finish it: make_sum_to_one in fp64
// (classify.py:387-393), then the top-two call rule (classify.py:285-295; ties to the lower
// index).  Each all-reduce is four DPP steps inside the rows plus one v_permlane16_swap across
// them - five ds_bpermute rounds of 64-bit values apiece made this the slowest 3k cycles of a
// window.
template <class T, class Combine>
__device__ __forceinline__ T reduce32(T v, const Combine& combine) {
    v = combine(v, T::template moved<0xB1>(v));     // quad_perm [1,0,3,2]
    v = combine(v, T::template moved<0x4E>(v));     // quad_perm [2,3,0,1]
    v = combine(v, T::template moved<0x141>(v));    // row_half_mirror
    v = combine(v, T::template moved<0x140>(v));    // row_mirror
    T row0, row1;                                   // both rows' results, seen from both rows
    T::rows(v, &row0, &row1);
    return combine(row0, row1);
}
// v_permlane16_swap_b32 (gfx950): (x, x) -> {the even row's x in both rows of a pair, the odd
// row's x in both rows} - the cross-row step of a 32-lane reduction without an LDS round trip.
__device__ __forceinline__ void rows_i32(int x, int* even, int* odd) {
    const auto r = __builtin_amdgcn_permlane16_swap((unsigned)x, (unsigned)x, false, false);
    *even = (int)r[0];
    *odd = (int)r[1];
}
__device__ __forceinline__ void rows_f64(double x, double* even, double* odd) {
    const long long b = __builtin_bit_cast(long long, x);
    int lo0, lo1, hi0, hi1;
    rows_i32((int)b, &lo0, &lo1);
    rows_i32((int)(b >> 32), &hi0, &hi1);
    *even = __builtin_bit_cast(double, (long long)(((unsigned long long)(unsigned)hi0 << 32) | (unsigned)lo0));
    *odd = __builtin_bit_cast(double, (long long)(((unsigned long long)(unsigned)hi1 << 32) | (unsigned)lo1));
}
struct RedF64 {
    double v;
    template <int CTRL>
    static __device__ __forceinline__ RedF64 moved(const RedF64& a) {
        return RedF64{dpp_move_f64<CTRL>(a.v)};
    }
    static __device__ __forceinline__ void rows(const RedF64& a, RedF64* even, RedF64* odd) {
        rows_f64(a.v, &even->v, &odd->v);
    }
};
struct RedBest {
    double v;
    int i;
    template <int CTRL>
    static __device__ __forceinline__ RedBest moved(const RedBest& a) {
        return RedBest{dpp_move_f64<CTRL>(a.v), dpp_move_i32<CTRL>(a.i)};
    }
    static __device__ __forceinline__ void rows(const RedBest& a, RedBest* even, RedBest* odd) {
        rows_f64(a.v, &even->v, &odd->v);
        rows_i32(a.i, &even->i, &odd->i);
    }
};

__device__ __forceinline__ void renormalise_and_call(float merged, int c, int n_classes,
                                                     double score_diff, float* probs_row,
                                                     int* call_out) {
    const bool valid = c < n_classes;
    double p = (double)merged;
    const double rest =
        reduce32(RedF64{(valid && c > 0) ? p : 0.0},
                 [](const RedF64& a, const RedF64& b) { return RedF64{a.v + b.v}; }).v;
    // (class 0 of this lane's half; the source lane made here: as a loop invariant its byte address
    // was kept in a register around the whole persistent loop - and spilled)
    int half_first;
    asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(half_first));
    half_first = (half_first & 32) << 2;
    const long long p_bits = __builtin_bit_cast(long long, p);
    const unsigned p0_lo = (unsigned)__builtin_amdgcn_ds_bpermute(half_first, (int)(unsigned)p_bits);
    const unsigned p0_hi = (unsigned)__builtin_amdgcn_ds_bpermute(half_first, (int)(p_bits >> 32));
    const double p0 = __builtin_bit_cast(double, ((long long)p0_hi << 32) | (long long)p0_lo);
    const double factor = (1.0 - p0) / rest;
    if (c > 0) p = p * factor;
    if (valid) probs_row[c] = (float)p;

    const RedBest best = reduce32(RedBest{valid ? p : -1.0, c}, [](const RedBest& a, const RedBest& b) {
        return (b.v > a.v || (b.v == a.v && b.i < a.i)) ? b : a;
    });
    const double second =
        reduce32(RedF64{(valid && c != best.i) ? p : -1.0},
                 [](const RedF64& a, const RedF64& b) { return RedF64{fmax(a.v, b.v)}; }).v;
    if (c == 0) *call_out = (best.i != 0 && (best.v - second) >= score_diff) ? best.i : 0;
}

// Window w of a launch = (read w / steps, scan step w % steps).  Window indices fit 32 bits
// (n_windows is an int) and steps == 1 - whole reads, the classify path - needs no division at
// all; a 64-bit division is ~150 scalar instructions that every wave would run per window.
__device__ __forceinline__ void split_window(unsigned win, int steps, unsigned* read, int* step) {
    if (steps == 1) {
        *read = win;
        *step = 0;
    } else {
        *read = win / (unsigned)steps;
        *step = (int)(win - *read * (unsigned)steps);
    }
}

// Window bounds of scan step `step` inside a read of `len` samples (classify.py:337-349).
__device__ __forceinline__ void window_bounds(long long len, int step, int side, long long* a,
                                              long long* b) {
    const long long sig_start = (long long)step * (kWindow / 2);
    const long long sig_end = sig_start + kWindow;
    if (side == 0) {
        *a = sig_start < len ? sig_start : len;
        *b = sig_end < len ? sig_end : len;
    } else {
        *a = len - sig_end > 0 ? len - sig_end : 0;
        *b = len - sig_start > 0 ? len - sig_start : 0;
    }
}

// z-normalisation constants from exact integer sums (trim_signal.py:61-69): x -> (x - mean) * inv
// with mean = sum(x)/n and inv = 1/std = n / sqrt(n*sum(x^2) - sum(x)^2), the radicand exact in
// int64; inv = 1 when std is 0 (the reference then only subtracts the mean).  Two fp64 divisions
// and one square root per window instead of a division per sample: fp64 division is ~20
// instructions at half rate, and stage A has nothing to hide them behind.  The product differs
// from the reference's quotient by at most one fp64 ulp before the cast to fp32 (the parity
// tests allow one fp32 ulp; both normalising kernels share this function, so they agree to the
// bit with each other).
__device__ __forceinline__ void mean_std(long long s1, long long s2, int cnt, double* mean,
                                         double* inv) {
    *mean = 0.0;
    *inv = 1.0;
    if (cnt > 0) {
        *mean = (double)s1 / (double)cnt;
        const long long num = (long long)cnt * s2 - s1 * s1;
        if (num > 0) *inv = (double)cnt / sqrt((double)num);
    }
}

// A pointer read from the kernel-argument segment is a generic ("flat") pointer to the compiler,
// and flat accesses are slower than global ones AND count against the LDS counter the hand-written
// fragment pipelines wait on.  glob() says what the pointer is: global memory.
template <class T>
__device__ __forceinline__ T* glob(T* p) {
    return (T*)(__attribute__((address_space(1))) T*)p;
}

// Arguments of the forward kernel (one by-value struct = the kernel-argument segment).
struct ForwardArgs {
    const float* packed;         // packed parameters (dbh_layout.h)
    const float* x;              // seam b1: [n_windows][1024] normalised windows, or null
    float* probs;                // [n_windows][n_classes]
    float* debug_out;
    const int16_t* samples;      // seam b2: int16 signals, or null
    const long long* offsets;    //          read r = samples[offsets[r] .. offsets[r+1])
    int* calls;                  //          barcode calls (one scan step per read), or null
    float* tail_scratch;         // [grid][kTailBatch][16][48]: conv17 outputs parked per workgroup
    int* win_counter;            // null: workgroup b walks groups b, b + grid, ...; else every
                                 // workgroup takes its next group of windows off this counter;
                                 // [1] counts the workgroups that have finished (both 0 between
                                 // launches: the last workgroup of a launch resets them)
    long long* clock_out;        // [grid][4 + kPhaseMarks * kPhaseGroups] or null: shader clock and 100
                                 // MHz clock at a workgroup's start and end (dbh_forward_clock_read),
                                 // then the phase stamps of its first groups (dbh_forward_phases_read)
    double score_diff;
    long long read0, len_hint, hint_cap;     // dbh_model_set_read_length_hint
    long long n_windows;
    int n_classes, debug_stage, steps, side;
    // windows not yet handed out below which a workgroup asks for groups of 2 / of 1 instead of
    // kGroup (the end of a launch: dbh_forward_kernel)
    int chunk4_min_left, chunk2_min_left;
    int phases;                  // clock probe on: also keep the phase stamps (dbh_forward_phases_enable)
};

// Window statistics, step 1: exact integer sums of this lane's two samples, sum(x) and sum(x^2)
// split in 16-bit halves so that every wave-wide partial stays below 2^31, reduced over the wave
// with DPP and left in the workgroup's kStatRed words.  Step 2 (after a barrier): mean and 1/std.
__device__ __forceinline__ void window_partial_sums(float* lds, int cnt, int v0, int v1, int tid,
                                                    int lane, int wave, int slot = 0) {
    const int biased0 = v0 + 32768, biased1 = v1 + 32768;       // 0 .. 65535
    const unsigned sq0 = (unsigned)(v0 * v0), sq1 = (unsigned)(v1 * v1);   // <= 2^30
    const int present = (tid < cnt ? 1 : 0) + (tid + kThreads < cnt ? 1 : 0);
    const int w_sum = wave_sum_i32((tid < cnt ? biased0 : 0) + (tid + kThreads < cnt ? biased1 : 0));
    const int w_cnt = wave_sum_i32(present);
    const int w_lo = wave_sum_i32((int)(sq0 & 0xFFFF) + (int)(sq1 & 0xFFFF));
    const int w_hi = wave_sum_i32((int)(sq0 >> 16) + (int)(sq1 >> 16));
    long long* red = reinterpret_cast<long long*>(lds + kStatRed + slot * 32);
    if (lane == 0) {
        red[wave] = (long long)w_sum - 32768LL * w_cnt;
        red[kWaves + wave] = ((long long)w_hi << 16) + (long long)w_lo;
    }
}
__device__ __forceinline__ void window_mean_inv(const float* lds, int cnt, double* mean,
                                                double* inv, int slot = 0) {
    const long long* red = reinterpret_cast<const long long*>(lds + kStatRed + slot * 32);
    long long s1 = 0, s2 = 0;
#pragma unroll
    for (int i = 0; i < kWaves; ++i) {
        s1 += red[i];
        s2 += red[kWaves + i];
    }
    mean_std(s1, s2, cnt, mean, inv);
}

// Seam-b2 input of one window, straight from the read's int16 samples: this lane's two samples
// for the window statistics (positions tid and tid + 512 of the slice) and the six samples its
// conv1d_1 products take as their B operand (positions 8j - 2 + 2t + q of the zero-padded window,
// t = 0..5: row t of quad j, tap q).  A position outside the slice (the padding on either side,
// the fourth "tap") is kOutside - no sample is: they are 16-bit.
constexpr int kOutside = 0x40000000;
__device__ __forceinline__ void fetch_window_at(const int16_t* __restrict__ src, int cnt,
                                                int pad_left, int tid, int j, int q, int& v0,
                                                int& v1, int (&raw)[6]) {
    // src = first sample of the window's slice (wave-uniform), cnt of them; every index below is
    // a 32-bit lane offset from it
    v0 = tid < cnt ? (int)src[(unsigned)tid] : 0;
    v1 = tid + 512 < cnt ? (int)src[(unsigned)(tid + 512)] : 0;
#pragma unroll
    for (int t = 0; t < 6; ++t) {
        const int k = 8 * j - 2 + 2 * t + q - pad_left;
        const bool inside = q < 3 && k >= 0 && k < cnt;
        raw[t] = inside ? (int)src[(unsigned)(inside ? k : 0)] : kOutside;
    }
}
__device__ __forceinline__ void fetch_window(const int16_t* __restrict__ samples, long long base,
                                             long long len, int step, int side, int tid, int j,
                                             int q, int& cnt, int& v0, int& v1, int (&raw)[6]) {
    long long wa, wb;
    window_bounds(len, step, side, &wa, &wb);
    cnt = (int)(wb - wa);
    fetch_window_at(samples + base + wa, cnt, (side == 0) ? 0 : kWindow - cnt, tid, j, q, v0, v1,
                    raw);
}

// =============================================================================================
// The kernel.  grid <= n_windows (persistent: a workgroup walks windows b, b + grid, ...), block = 512.
//   x      [n_windows][1024]   normalised windows (fp32)
//   probs  [n_windows][n_classes]
//   debug_stage in [0,7]: write the activations after stage 'A'+debug_stage to debug_out and
//   stop; 100+k: stop after stage k without writing (per-stage timing); -1: full forward.
// =============================================================================================
//   Fused seam-b2 mode (samples != nullptr): window w = (read w / steps, scan step w % steps) is
//   sliced and z-normalised from the int16 signal inside stage A, and when calls != nullptr
//   (steps == 1) the read is finished here too: renormalise + barcode call, no merge kernel.
__global__ __launch_bounds__(kThreads, 2) void dbh_forward_kernel(ForwardArgs by_value) {
    (void)by_value;
    // The arguments are read from the kernel-argument segment where they are used, through a
    // pointer made opaque each time: held in SGPRs across the persistent loop they (33 registers)
    // pushed the loop body into spilling scalars to vector lanes.
    typedef const __attribute__((address_space(4))) ForwardArgs* ArgsPtr;
    auto args = []() -> ArgsPtr {
        ArgsPtr p = (ArgsPtr)__builtin_amdgcn_kernarg_segment_ptr();
        asm volatile("" : "+s"(p));
        return p;
    };
    const float* __restrict__ packed_entry = glob(args()->packed);
    const int debug_stage = args()->debug_stage;
    // (window indices fit 32 bits: the host launches at most 2^20 reads at a time)
    const int n_windows = (int)args()->n_windows;
    // (the two pointers every window's first instructions branch on: read once - a scalar load at
    // the top of a window is ~200 cycles that every wave spends in front of the first barrier)
    const int16_t* const samples_entry = glob(args()->samples);
    int* const win_counter_entry = glob(args()->win_counter);
    __shared__ __attribute__((aligned(16))) float lds[kLdsFloats];

    const int tid_entry = threadIdx.x;
    // clock probe (off unless asked for): how fast the shader clock really runs under this load
    if (args()->clock_out != nullptr && tid_entry == 0) {
        long long* c = glob(args()->clock_out) + (size_t)blockIdx.x * (4 + kPhaseMarks * kPhaseGroups);
        c[0] = (long long)__builtin_readcyclecounter();
        c[1] = (long long)__builtin_amdgcn_s_memrealtime();
    }
    // PHASE STAMPS (clock probe on): the shader clock at five points of every group - its start, the
    // end of its stage A-C loop, of the stage D-E chain, of its stage F loop, of the batched tail -
    // taken by one lane right behind a barrier (nothing in flight that the counter's read could
    // hold up) and kept in LDS until the workgroup is done: the kernel measured is the kernel that
    // ships.
    const bool phases_on = args()->clock_out != nullptr && args()->phases != 0;
    int phase_group = 0;
    auto phase_stamp = [&](int mark) {
        if (phases_on && threadIdx.x == 0 && phase_group < kPhaseGroups)
            reinterpret_cast<unsigned*>(lds + kPhase)[phase_group * kPhaseMarks + mark] =
                (unsigned)__builtin_readcyclecounter();
    };
    // (an interval of stage F, added to the group's word `mark`; `since` moves on)
    auto phase_add = [&](int mark, unsigned& since) {
        if (phases_on && threadIdx.x == 0 && phase_group < kPhaseGroups) {
            const unsigned now = (unsigned)__builtin_readcyclecounter();
            reinterpret_cast<unsigned*>(lds + kPhase)[phase_group * kPhaseMarks + mark] += now - since;
            since = now;
        }
    };
    // debug_stage k: dump the activations after stage k and stop; 100+k: just stop (timing).
    const int stop_stage = debug_stage >= 100 ? debug_stage - 100 : debug_stage;

    // The LDS copy of stage B-D's epilogue parameters (biases of conv2..9, BN2..4) and the counter
    // of the split barrier: once per workgroup, published by the first window's first barrier;
    // nothing below writes lds[kParams..] again.
    for (int i = tid_entry; i < kParamFloats; i += kThreads) {
        const int src = i < kTabBias1 - kTabBias0 ? kTabBias0 + i
                                                  : kTabBn0 + (i - (kTabBias1 - kTabBias0));
        lds[kParams + i] = packed_entry[src];
    }
    if (tid_entry < kSyncWords) reinterpret_cast<unsigned*>(lds + kSync)[tid_entry] = 0u;
    if (tid_entry < 8) reinterpret_cast<unsigned*>(lds + kPairSync)[tid_entry] = 0u;
    // conv1d_1's weights as the A operand of its transposed MFMAs: lane (m, k) holds w[k][16g + m]
    // for the three channel groups g (tap k = lane >> 4; the fourth k is a zero column).  Its bias
    // and BN1 come from the LDS table.
    float bw_a[3];
    {
        const int n_e = tid_entry & 15, q_e = (tid_entry & 63) >> 4;
#pragma unroll
        for (int g = 0; g < 3; ++g)
            bw_a[g] = (q_e < 3) ? packed_entry[weight_offset(0) + q_e * 48 + g * 16 + n_e] : 0.f;
    }
    unsigned chain_windows = 0;   // windows this workgroup has taken through stage B (stage_b_chain)
    unsigned d_groups = 0;        // groups it has taken through stage D (stage_d_chain)
    int tail_slot = 0;            // windows of this workgroup waiting for the batched tail
    unsigned pair_rounds = 0;     // exchanges the wave pairs of conv7 have made so far

    // This workgroup's scratch in global memory (dbh_layout.h: kWgScratchFloats): conv17's outputs
    // waiting for the batched tail, and conv7's parks.
    float* const wg_scratch_entry = glob(args()->tail_scratch) + (size_t)blockIdx.x * kWgScratchFloats;


    // PERSISTENT GRID, GROUPS OF WINDOWS: the launch has at most one workgroup per CU (what 160 KiB
    // of LDS allows anyway).  A workgroup takes kGroup = 4 consecutive windows at a time: stages A,
    // B, C for each of them (conv7's output parked in global memory), stage D once for the four
    // together (stage_d_chain), then stages E and F for each, and every second group the batched
    // tail.  The first group of workgroup b is windows 4b ..; every further one comes off a counter
    // in global memory (win_counter != null, production launches; asked for by one lane in the
    // group's first stage A, known behind its stage B) - a workgroup that gets its CU late (another
    // kernel sat there: the inflate kernels of the streaming path) or runs slower (samples read
    // over PCIe) simply takes fewer.  Towards the end of a launch the groups shrink (two windows,
    // then one: chunk4_min_left / chunk2_min_left windows not yet handed out), so that the
    // workgroups finish within a window's time of each other, not within a group's.
    int group_start = (int)blockIdx.x * kGroup;
    int group_n = n_windows - group_start >= kGroup ? kGroup : n_windows - group_start;
    bool first_group = true;
    bool staged = false;          // this group's samples and statistics wait in the LDS staging
    bool thirds_ahead = false;    // slots 1 and 2 of conv2's weights requested under the group before
    // debug_stage >= 0 (tests, timeline): the batched tail runs behind every group
    const bool tail_every_group = debug_stage >= 0;

    // the first window of the next group, fetched under the group's last stage F: count, left padding
    // and this lane's two samples
    int carry_cnt = 0, carry_pad = 0, carry_v0 = 0, carry_v1 = 0;

    while (group_n > 0) {
    // (taken over and cleared at once: defined on every path round the loop, the four do not count as
    // live across stage B)
    const int first_cnt = carry_cnt, first_pad = carry_pad;
    int first_v0 = carry_v0, first_v1 = carry_v1;
    carry_cnt = 0;
    carry_pad = 0;
    carry_v0 = 0;
    carry_v1 = 0;
    // how many windows the NEXT group asks for: by what was left when this one was handed out
    const int left_now = n_windows - (group_start + group_n);
    const int chunk_next = win_counter_entry == nullptr             ? kGroup
                           : left_now >= args()->chunk4_min_left ? kGroup
                           : left_now >= args()->chunk2_min_left ? 2
                                                                 : 1;
    // (fixed shares - debug and timeline launches: groups b, b + grid, ...)
    int next_start = group_start + kGroup * (int)gridDim.x;
    int next_n = n_windows - next_start <= 0 ? 0
                 : n_windows - next_start < kGroup ? n_windows - next_start : kGroup;

    phase_stamp(0);
    if (phases_on && threadIdx.x == 0 && phase_group < kPhaseGroups)
        for (int i = 5; i < kPhaseMarks; ++i) reinterpret_cast<unsigned*>(lds + kPhase)[phase_group * kPhaseMarks + i] = 0u;
    // ================= stages A, B, C: one window of the group after the other ==================
    for (int k = 0; k < group_n; ++k) {
    // The thread index and the parameter pointer are made opaque once per round: otherwise the
    // loop-invariant-code pass hoists every lane address and constant of the (fully unrolled)
    // body out of the loop and keeps them alive across it - 245 spilled VGPRs instead of none.
    int tid = tid_entry;
    asm volatile("" : "+v"(tid));
    const __attribute__((address_space(1))) float* packed_opaque =
        (const __attribute__((address_space(1))) float*)packed_entry;
    __attribute__((address_space(1))) float* scratch_opaque =
        (__attribute__((address_space(1))) float*)wg_scratch_entry;
    asm volatile("" : "+s"(packed_opaque), "+s"(scratch_opaque));
    const float* __restrict__ packed = (const float*)packed_opaque;
    float* const wg_scratch = (float*)scratch_opaque;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int n = lane & 15, q = lane >> 4;
    const int win = group_start + k;
    // 300: timeline mode - lane 0 of every wave stamps the cycle counter at each phase boundary
    unsigned ts_acc = 0u;
    unsigned* ts = nullptr;
    long long* ts_out = nullptr;
    if (debug_stage >= 300) {                   // 301: the same in a persistent launch
        ts = &ts_acc;
        ts_out = reinterpret_cast<long long*>(glob(args()->debug_out)) + ((long)win * kWaves + wave) * 64;
    }
    mark(ts, 0);
    mark_realtime(ts, 62);
    unsigned ac_since = phases_on ? (unsigned)__builtin_readcyclecounter() : 0u;
    // (opaque once per round like the parameter pointer: see above)
    const __attribute__((address_space(1))) int* wc_opaque =
        (const __attribute__((address_space(1))) int*)win_counter_entry;
    const __attribute__((address_space(1))) int16_t* smp_opaque =
        (const __attribute__((address_space(1))) int16_t*)samples_entry;
    asm volatile("" : "+s"(wc_opaque), "+s"(smp_opaque));
    int* const win_counter = (int*)wc_opaque;
    int taken = 0;

    // ---------------- stage A: conv1d_1 (k3, stride 2, pad right) + ReLU + BN1 ---------------
    // ... has no phase of its own: the wave that owns a tile of conv1d_2's quads computes the
    // conv1d_1 rows they need inside conv1d_2's tile 0, in registers (w43a_tile0).  What is left
    // here: this lane's six samples of the window, normalised.
    //   Steady state (seam b2, second group of a workgroup onwards): the window's samples and
    // statistics wait in the LDS staging (fetched under the group before: stages E, F).  For the
    // group's first window the first third of conv1d_2's weights has been on its way to slot 0 since
    // that group's last stage F and ONE barrier publishes it and hands the LDS over; for the others
    // thirds 0 and 2 landed under conv7 of the window before (slots 0 and 2 are free behind its
    // mid-layer barrier; its closing barrier retired them) and there is NO barrier at all.  What is
    // missing of the thirds is requested between conv1d_1's MFMAs.
    //   Cold (a workgroup's first group; every window of seam b1): the samples are asked for here,
    // behind the barrier of the window statistics, and for the group's first window a second
    // barrier waits for the weights.
    ConvAIn in_a;
    bool weights_cold = false;
    int thirds_mode = 0;      // conv2's thirds tile 0 still has to request: 1 = third 1, 2 = thirds 1 and 2
    // the zero rows of stage B's halo arrays ('same' padding at the two ends of the window; the
    // place is taken by other stages' weights in between): published by the arrivals of stage B's
    // first tiles (every wave's, behind its LDS stores), long before conv3 reads them
    if (tid < 192) {
        const int a = tid / 48, c = tid - a * 48;
        lds[kHalo + (a >> 1) * 2 * kHaloRows + ((a & 1) ? kHaloRows + 48 : 8 * 96) + c] = 0.f;
    }
    {
        // quad of this lane's MFMA column: j = 16 wave + n (stage_b_chain: the neighbours of a quad
        // sit in the neighbouring lanes)
        const int j = wave * 16 + n;
        auto fetch_conv2_weights = [&] {      // all three thirds (slots 0..2 are adjacent)
            dma_weights<3 * kWinoHalf>(packed + weight_offset(1), lds + kSlot0, lane, wave);
        };
        const int16_t* __restrict__ samples = (const int16_t*)smp_opaque;
        // (debug_stage 0 / 1 leave stage B half-way, conv7 - under which a group's later windows
        // get their weights - does not run: every window starts cold)
        const bool stop_early = stop_stage == 0 || stop_stage == 1;
        if (samples == nullptr) {
            if (k == 0 || stop_early) {
                // a later group of this workgroup: the group before may still be read (stage H)
                if (!first_group || k > 0) full_barrier();
                fetch_conv2_weights();
                weights_cold = true;
            } else {
                thirds_mode = 1;
            }
            const float* xw = glob(args()->x) + (long)win * kWindow;
#pragma unroll
            for (int t = 0; t < 6; ++t) {
                const int idx = 8 * j - 2 + 2 * t + q;            // q = tap (3 = zero column)
                in_a.xs[t] = (q < 3 && idx >= 0 && idx < kWindow) ? xw[idx] * kActScale : 0.f;
            }
        } else {
            // fused slice + normalise (same arithmetic as dbh_normalise_kernel)
            double mean, inv;
            int in_raw[6];
            if (!staged) {
                int in_cnt = 0, in_v0 = 0, in_v1 = 0;
                ArgsPtr a = args();
                const int steps = a->steps, side = a->side;
                const long long read0 = a->read0, len_hint = a->len_hint, hint_cap = a->hint_cap;
                const long long* __restrict__ offsets = glob(a->offsets);
                unsigned read;
                int step;
                split_window((unsigned)win, steps, &read, &step);
                // Where a read starts is itself in memory (offsets[read]), and at the top of a
                // kernel a dependent load costs ~3k cycles.  If the caller says that all reads are
                // len_hint samples long, the samples are requested from where that puts them
                // TOGETHER with the offsets, and fetched again only if the offsets disagree.
                const long long guess = (read0 + read) * len_hint;
                const bool speculate = len_hint > 0 && guess + len_hint <= hint_cap;
                if (speculate)
                    fetch_window(samples, guess, len_hint, step, side, tid, j, q, in_cnt, in_v0,
                                 in_v1, in_raw);
                const long long base = offsets[read];
                const long long len = offsets[read + 1] - base;
                if (!speculate || base != guess || len != len_hint)
                    fetch_window(samples, base, len, step, side, tid, j, q, in_cnt, in_v0, in_v1,
                                 in_raw);
                window_partial_sums(lds, in_cnt, in_v0, in_v1, tid, lane, wave);
                full_barrier();
                if (k == 0 || stop_early) {
                    fetch_conv2_weights();
                    weights_cold = true;
                } else {
                    thirds_mode = 1;
                }
                window_mean_inv(lds, in_cnt, &mean, &inv);
            } else {
                int cnt, pad;
                if (k == 0 && DBH_STATS0_IN_F) {
                    // the group's first window was staged like the others (its statistics by wave 4
                    // under the group before's stage F); the barrier publishes slot 0's weights
                    // (asked for in that stage F) and keeps this group off the LDS that group's last
                    // reads still use
                    full_barrier();
                    mark(ts, 51);
                    thirds_mode = thirds_ahead ? 0 : 2;
                    const float* st = lds + kStageStats;
                    mean = reinterpret_cast<const double*>(st)[0];
                    inv = reinterpret_cast<const double*>(st)[1];
                    cnt = reinterpret_cast<const int*>(st)[4];
                    pad = reinterpret_cast<const int*>(st)[5];
                } else if (k == 0) {
                    // the group's first window came in registers (fetched under the last stage F of
                    // the group before): its samples go to the staging and its exact sums ride on
                    // the barrier that also publishes slot 0's weights (asked for in that stage F)
                    // and keeps this group off the LDS that group's last reads still use
                    short* put = reinterpret_cast<short*>(lds + kStage);
                    put[tid] = (short)first_v0;
                    put[tid + 512] = (short)first_v1;
                    window_partial_sums(lds, first_cnt, first_v0, first_v1, tid, lane, wave);
                    full_barrier();
                    mark(ts, 51);
                    window_mean_inv(lds, first_cnt, &mean, &inv);
                    cnt = first_cnt;
                    pad = first_pad;
                    thirds_mode = thirds_ahead ? 0 : 2;
                } else {
                    thirds_mode = 1;
                    const float* st = lds + kStageStats + k * 8;
                    mean = reinterpret_cast<const double*>(st)[0];
                    inv = reinterpret_cast<const double*>(st)[1];
                    cnt = reinterpret_cast<const int*>(st)[4];
                    pad = reinterpret_cast<const int*>(st)[5];
                }
                const short* smp = reinterpret_cast<const short*>(lds + kStage + k * kStageWin);
#pragma unroll
                for (int t = 0; t < 6; ++t) {
                    const int i = 8 * j - 2 + 2 * t + q - pad;
                    const bool inside = q < 3 && i >= 0 && i < cnt;
                    in_raw[t] = inside ? (int)smp[inside ? i : 0] : kOutside;
                }
            }
            inv *= (double)kActScale;      // (exact; see kActScale)
#pragma unroll
            for (int t = 0; t < 6; ++t)
                in_a.xs[t] = in_raw[t] != kOutside ? (float)(((double)in_raw[t] - mean) * inv) : 0.f;
            mark(ts, 54);
        }
        // (the next group's first window: asked for behind the barriers above - they wait for every
        // outstanding request - and needed behind stage B's closing barrier)
        if (win_counter != nullptr && tid == 0 && k == 0) taken = atomicAdd(win_counter, chunk_next);
        if (weights_cold) full_barrier();     // conv1d_2's weights have landed
#pragma unroll
        for (int g = 0; g < 3; ++g) in_a.w[g] = bw_a[g];
        in_a.p4_edge = j == 0 ? f2{0.f, 0.f} : f2{4.f, 4.f};
        in_a.one_edge = j == 127 ? f2{0.f, 0.f} : f2{1.f, 1.f};
        in_a.dump_on = debug_stage == 0;
        in_a.dump = nullptr;
        if (in_a.dump_on)
            in_a.dump = glob(args()->debug_out) + (long)win * kStageFloats[0] + 4 * j * 48 + 4 * q;
        in_a.stop = stop_stage == 0;
        in_a.wave_hi = wave >= 4;
        in_a.dump_b = debug_stage == 1;
        mark(ts, 1);
    }

    // (a group's later windows start without a barrier of their own: the first third of conv2's
    // weights was retired by conv7's mid-layer barrier, the last third - asked for behind it - by
    // stage B's first arrivals, which tile 1 waits for; the park's stores likewise)
    phase_add(9, ac_since);
    // ---------------- stage B: conv2,3,4 (L=512) + MaxPool + BN2, conv5, conv6 ----------------
    // One chain in registers (stage_b_chain): nine Winograd F(4,3) tiles back to back, no
    // workgroup barrier, no activation image between the layers; conv5 and conv6 on its end.
    stage_b_chain(
        lds, packed, tid, lane, wave, ts, chain_windows, in_a, thirds_mode != 0,
        [&](int i) {
            // a request costs ~100 cycles of issue: one piece behind each of conv1's first MFMAs
            if (thirds_mode == 2 && i < (2 * kWinoHalf / 256 + kWaves - 1) / kWaves)
                dma_weights_one<2 * kWinoHalf>(packed + weight_offset(1) + kWinoHalf, lds + kSlot1,
                                               lane, wave, i);
            if (thirds_mode == 1 && i < (kWinoHalf / 256 + kWaves - 1) / kWaves)
                dma_weights_one<kWinoHalf>(packed + weight_offset(1) + kWinoHalf, lds + kSlot1, lane,
                                           wave, i);
        },
        [&] {
            // (this wave's arrival has waited for the atomic's answer)
            if (win_counter != nullptr && tid == 0 && k == 0)
                reinterpret_cast<int*>(lds + kNextWin)[0] = taken;
        },
        [&] { return glob(args()->debug_out) + (long)win * kStageFloats[1]; },
        // (a group's later windows: the first third waits where conv7 of the window before put it)
        lds + ((k > 0 && !weights_cold) ? kChainP : kSlot0));
    if (stop_stage == 0 || stop_stage == 1) {
        // (debug_stage 0: tile 0 has written the dump itself; 1: dumped from registers, conv5 did
        // not run.  The chain was left half-way: its counters start over for the next window.)
        full_barrier();
        if (tid < kSyncWords) reinterpret_cast<unsigned*>(lds + kSync)[tid] = 0u;
        chain_windows = 0;
        full_barrier();
        continue;
    }
    // where this workgroup's NEXT GROUP starts: written by thread 0 early in stage B of the group's
    // first window, published by that stage's closing barrier
    if (k == 0 && win_counter != nullptr) {
        next_start = kGroup * (int)gridDim.x + reinterpret_cast<const int*>(lds + kNextWin)[0];
        const int left = n_windows - next_start;
        next_n = left <= 0 ? 0 : left < chunk_next ? left : chunk_next;
    }

    // ---------------- stage C: conv7 (L=256, F(4,3)) + MaxPool + BN3 -> this window's park -----
    // Meanwhile the NEXT window's conv2 weights: the first third -> kChainP (the idle part of the
    // activation buffer: slot 0 holds conv7's own N tile 0) in the layer's first steps, the last
    // third -> slot 2 behind the mid-layer barrier (conv7's N tile 2 lay there); five pieces per
    // requesting wave and third, two per step.  The layer ends with an LDS-only barrier: the park's
    // stores and these requests are retired by the full barrier in front of the next window's stage B.
    phase_add(10, ac_since);
    // (the group's last window: below, outside the loop)
    if (k == group_n - 1) {
        flush_marks(ts, ts_out, lane);
        break;
    }
    {
        auto third_step = [&](const float* src, float* dst, int step) {
            constexpr int NW = DBH_DMA_WAVES;
            constexpr int per_wave = (kWinoHalf / 256 + NW - 1) / NW;
            if (2 * step < per_wave) dma_weights_one<kWinoHalf, NW>(src, dst, lane, wave, 2 * step);
            if (2 * step + 1 < per_wave) dma_weights_one<kWinoHalf, NW>(src, dst, lane, wave, 2 * step + 1);
        };
        w43_nsplit_pooled_layer<6, 2, false>(
            lds, packed, wg_scratch + kWgPark7Off + k * kPark7Floats, tid, lane, wave, ts, 22, pair_rounds,
            [&](auto tag) {
                constexpr int G = decltype(tag)::value;
                if constexpr (G < 3) third_step(packed + weight_offset(1), lds + kChainP, G);
                if constexpr (G >= 6) third_step(packed + weight_offset(1) + 2 * kWinoHalf, lds + kSlot2, G - 6);
            });
    }
    phase_add(11, ac_since);
    if (stop_stage == 2) {
        full_barrier();      // (the park's stores are out)
        if (debug_stage < 100) {
            // (from the park, in the order stage D reads it)
            float* out = glob(args()->debug_out) + (long)win * kStageFloats[2];
            const float* pk = wg_scratch + kWgPark7Off + k * kPark7Floats;
            for (int idx = tid; idx < 128 * 48; idx += kThreads) {
                const int p = idx / 48, c = idx - p * 48;
                const int hf = p >> 6, nn = (p >> 2) & 15, i = p & 3, g = c >> 4, qq = (c >> 2) & 3, r = c & 3;
                out[idx] = pk[hf * 3072 + (i * 3 + g) * 256 + (qq * 16 + nn) * 4 + r] * kActUnscale;
            }
        }
        continue;
    }
    flush_marks(ts, ts_out, lane);
    }   // stages A, B, C of the group's windows

    if (stop_stage != 0 && stop_stage != 1) {
    // (opaque once per phase: see the top of the loop above)
    int tid = tid_entry;
    asm volatile("" : "+v"(tid));
    const __attribute__((address_space(1))) float* packed_opaque =
        (const __attribute__((address_space(1))) float*)packed_entry;
    __attribute__((address_space(1))) float* scratch_opaque =
        (__attribute__((address_space(1))) float*)wg_scratch_entry;
    asm volatile("" : "+s"(packed_opaque), "+s"(scratch_opaque));
    const float* __restrict__ packed = (const float*)packed_opaque;
    float* const wg_scratch = (float*)scratch_opaque;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    unsigned ts_acc = 0u;
    unsigned* ts = nullptr;
    long long* ts_out = nullptr;
    if (debug_stage >= 300) {
        ts = &ts_acc;
        ts_out = reinterpret_cast<long long*>(glob(args()->debug_out)) +
                 ((long)(group_start + (wave >> 1 < group_n ? wave >> 1 : 0)) * kWaves + wave) * 64;
    }
    const bool seam_b2 = samples_entry != nullptr;

    // ---------------- stage C of the group's LAST window, with stage D's first requests ---------
    // Stage D's operands of the group's EARLIER windows start their trip from the parks behind this
    // layer's last MFMAs (waves 2k, 2k + 1 own window k there; twelve 16-byte loads per lane into
    // registers conv7 has finished with; in front of the layer's exchange, last epilogue and
    // closing barrier, which does not wait for them) - the parks are MALL / HBM resident (120 KB per
    // workgroup: more than the L2 holds), a round trip of thousands of cycles.  The last window's own output goes to
    // LDS (w43_nsplit_half<LAST>) and is read back behind the layer's closing barrier.  conv8's
    // thirds 0 and 1 -> stage D's slots 0 and 1 (the idle part of the activation buffer) in the
    // layer's first half: nothing is requested in its second half, whose closing barrier waits for
    // what was (measured with them there: 3 k cycles per group longer than the other windows' conv7).
    // (the waves of windows the group does not have run along on whatever their parks hold: the
    // weight slots and their counters are shared by all eight, and nothing of theirs is looked at)
    const int k_last = group_n - 1;
    const bool own_last = (wave >> 1) == k_last;
    f2 Y[3][2][4];
    unsigned c7_since = phases_on ? (unsigned)__builtin_readcyclecounter() : 0u;
    {
        auto third_step = [&](const float* src, float* dst, int step) {
            constexpr int NW = DBH_DMA_WAVES;
            constexpr int per_wave = (kWinoHalf / 256 + NW - 1) / NW;
            if (2 * step < per_wave) dma_weights_one<kWinoHalf, NW>(src, dst, lane, wave, 2 * step);
            if (2 * step + 1 < per_wave) dma_weights_one<kWinoHalf, NW>(src, dst, lane, wave, 2 * step + 1);
        };
        w43_nsplit_pooled_layer<6, 2, true>(
            lds, packed, nullptr, tid, lane, wave, ts, 59, pair_rounds, [&](auto tag) {
                constexpr int G = decltype(tag)::value;
                if constexpr (G < 3) third_step(packed + weight_offset(7), lds + kDS0, G);
                else if constexpr (G < 6) third_step(packed + weight_offset(7) + kWinoHalf, lds + kDS1, G - 3);
#if DBH_Y_EARLY
                if constexpr (G == 6)
                    d_load_y(Y, (const float*)(wg_scratch + kWgPark7Off + (wave >> 1) * kPark7Floats +
                                               (wave & 1) * 3072 + lane * 4));
#endif
            },
            [&]() -> bool {
#if DBH_Y_EARLY
                return true;
#endif
                // (every wave: the two that own the last window load what their park holds - stale -
                // and overwrite it below.  Loaded under a condition the registers would count as
                // live around the whole persistent loop - on the path that takes neither branch -
                // and be spilled in stage B.)
                d_load_y(Y, (const float*)(wg_scratch + kWgPark7Off + (wave >> 1) * kPark7Floats +
                                           (wave & 1) * 3072 + lane * 4));
                return true;
            });
    }
    phase_add(12, c7_since);
    if (stop_stage == 2) {
        if (debug_stage < 100) {
            float* out = glob(args()->debug_out) + (long)(group_start + k_last) * kStageFloats[2];
            const float* pk = lds + kPark7Lds;
            for (int idx = tid; idx < 128 * 48; idx += kThreads) {
                const int p = idx / 48, c = idx - p * 48;
                const int hf = p >> 6, nn = (p >> 2) & 15, i = p & 3, g = c >> 4, qq = (c >> 2) & 3, r = c & 3;
                out[idx] = pk[hf * 3072 + (i * 3 + g) * 256 + (qq * 16 + nn) * 4 + r] * kActUnscale;
            }
        }
        full_barrier();
    } else {
    // (kernel arguments of the sample prefetch: scalar loads, long back when they are needed)
    const long long* __restrict__ offsets_arg = glob(args()->offsets);
    const int steps_arg = args()->steps;
    const int side_arg = args()->side;
    if (own_last)
        d_load_y(Y, (const float*)(lds + kPark7Lds + (wave & 1) * 3072 + lane * 4));

    phase_add(13, c7_since);
    phase_stamp(1);
    // ================= stages D and E: conv8, conv9 (+ MaxPool + BN4) and the inception block (+
    // MaxPool + BN5), the group together, one chain in registers ==================================
    // What stage F needs of global memory is asked for INSIDE the chain, in front of conv1d_16's
    // MFMAs (the chain's closing barrier retires it): conv1d_17's fragments - this wave's EIGHTH of
    // the contraction, three groups of eight channels x three taps x three N tiles = 27 fragment
    // pairs, buffer loads the compiler does not see (scalar base and fragment offset, the lane's
    // eight bytes) - and the two offsets of each of the next group's windows (VECTOR loads - the
    // address made per-lane on purpose: as scalar loads they would count against lgkmcnt, which the
    // compiler's LDS waits watch).  The 110 KB of fragments cross the CU's vector memory path once
    // per group (64 B a cycle: 1.7 k cycles, under conv1d_16) instead of twice in front of stage F's
    // MFMAs, with every matrix pipe idle (waves 0-3 and 4-7 each fetched all of it for two windows).
    f2 w17[27];
    long long noff0[kGroup], noff1[kGroup];
    int nstep[kGroup];
    // bias and BN6 of the output quadruples this lane finishes (tid and, for tid < 256, 512 + tid of
    // the group's 768: quadruple e = window e / 192, N tile (e % 192) / 64, lane (e % 64))
    EpiParams<1, true> ep17[2];
    auto fetch_f = [&]() {
#pragma unroll
        for (int r = 0; r < 2; ++r) {
            const int e = (r * kThreads + tid) % 192, ch = (e >> 6) * 16 + (e & 15);
            ep17[r].load(packed + bias_offset(16) + ch, packed + bn_scale_offset(5) + ch, packed + bn_shift_offset(5) + ch);
        }
        const __amdgpu_buffer_rsrc_t view = buffer_view(packed + weight_offset(16));
        const unsigned lane_bytes = (unsigned)lane * 8u;
#pragma unroll
        for (int tap = 0; tap < 3; ++tap)
#pragma unroll
            for (int sp = 0; sp < 3; ++sp)
#pragma unroll
                for (int t = 0; t < 3; ++t) {
                    const unsigned soff = (unsigned)(((tap * 24 + wave * 3 + sp) * 3 + t) * 128) * 4u;
                    asm volatile("buffer_load_dwordx2 %0, %1, %2, %3 offen"
                                 : "=v"(w17[(tap * 3 + sp) * 3 + t])
                                 : "v"(lane_bytes), "s"(view), "s"(soff)
                                 : "memory");
                }
#pragma unroll
        for (int w = 0; w < kGroup; ++w) {
            noff0[w] = 0;
            noff1[w] = 0;
            nstep[w] = 0;
            if (seam_b2 && w < next_n) {
                unsigned next_read;
                split_window((unsigned)(next_start + w), steps_arg, &next_read, &nstep[w]);
                unsigned lane_zero = 0;
                asm volatile("" : "+v"(lane_zero));
                noff0[w] = offsets_arg[next_read + lane_zero];
                noff1[w] = offsets_arg[next_read + lane_zero + 1];
            }
        }
    };
    {
        float* dump3 = nullptr;
        if (debug_stage == 3 && (wave >> 1) < group_n)
            dump3 = glob(args()->debug_out) + (long)(group_start + (wave >> 1)) * kStageFloats[3];
        stage_d_chain(lds, packed, wg_scratch, lane, wave, ts, d_groups, Y, dump3, stop_stage == 3,
                      (wave >> 1) < group_n ? cat_offset(wave >> 1, group_n) : -1, fetch_f);
    }
    if (stop_stage == 4) {
        if (debug_stage < 100)
            for (int kk = 0; kk < group_n; ++kk)
                dump_stage(lds + cat_offset(kk, group_n), kS192, 32, 192,
                           glob(args()->debug_out) + (long)(group_start + kk) * kStageFloats[4], tid);
        full_barrier();
    } else if (stop_stage != 3) {
    // ================= stage F: conv17 (192->48, k3, stride 2) + ReLU + BN6 -> 16 x 48, the group
    // together ====================================================================================
    // The four concat images wait in LDS; conv17's weights (110 KB, each used once per window) skip
    // LDS: every wave takes an EIGHTH of the contraction (three groups of eight channels x three taps
    // x three N tiles = 27 fragment pairs in registers, fetched from L2 once per group - inside the
    // chain, above) for all four windows: 216 MFMAs per wave in one run, twelve independent
    // accumulator chains.  Then ONE reduction for the group - 96 partial tiles through LDS, every
    // lane finishing one or two of the 768 output quadruples - where the window-by-window form paid
    // two barriers, a pipeline fill and a three-wave reduction per window (8.3 k cycles per window
    // for 3.5 k of matrix work: profiles/r06_*).  conv17's output (3 KB per window) goes to the
    // workgroup's slots in global memory for the batched tail below.
    //   The NEXT group's samples (seam b2) are fetched meanwhile: the places of its windows in the
    // sample buffer (offsets) came with the fragments; the samples themselves are asked for in front
    // of the MFMAs; their exact sums ride on the last barrier, and waves 5-7 turn those of windows
    // 1-3 into mean and 1/std while the samples go to the LDS staging; window 0 stays in registers
    // for that group's stage A (whose first barrier carries its sums).
    phase_stamp(2);
    const bool batch_ends = tail_every_group || tail_slot + group_n + next_n > kTailBatch || next_n == 0;
    const bool slot0_next = seam_b2 && next_n > 0;       // the next group's first window: conv2's weights
    const bool thirds_now = slot0_next && !batch_ends;
    const bool run_tail = batch_ends;
    const bool thirds_next = thirds_now;
    const int n = lane & 15, q = lane >> 4;
    if (tid < group_n) reinterpret_cast<int*>(lds + kTailWins)[tail_slot + tid] = group_start + tid;
    flush_marks(ts, ts_out, lane);
    unsigned f_since = phases_on ? (unsigned)__builtin_readcyclecounter() : 0u;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // the fragments and the offsets are in
#pragma unroll
    for (int k17 = 0; k17 < 27; ++k17) asm volatile("" : "+v"(w17[k17]));
#pragma unroll
    for (int w = 0; w < kGroup; ++w) asm volatile("" : "+v"(noff0[w]), "+v"(noff1[w]));
    // (3) the next group's samples: this lane's two of each window - asked for here, behind the wait
    // that has just retired the offsets, in front of the layer's 216 MFMAs per wave, which cover their
    // trip from HBM
    int ncnt[kGroup], npad[kGroup], nv0[kGroup], nv1[kGroup];
#pragma unroll
    for (int w = 0; w < kGroup; ++w) {
        ncnt[w] = 0;
        npad[w] = 0;
        nv0[w] = 0;
        nv1[w] = 0;
        if (seam_b2 && w < next_n) {
            long long wa, wb;
            const long long next_base =
                ((long long)__builtin_amdgcn_readfirstlane((int)(noff0[w] >> 32)) << 32) |
                (unsigned)__builtin_amdgcn_readfirstlane((int)noff0[w]);
            const long long next_end =
                ((long long)__builtin_amdgcn_readfirstlane((int)(noff1[w] >> 32)) << 32) |
                (unsigned)__builtin_amdgcn_readfirstlane((int)noff1[w]);
            window_bounds(next_end - next_base, nstep[w], side_arg, &wa, &wb);
            ncnt[w] = (int)(wb - wa);
            npad[w] = (side_arg == 0) ? 0 : kWindow - ncnt[w];
            const int16_t* src = (const int16_t*)samples_entry + next_base + wa;
            nv0[w] = tid < ncnt[w] ? (int)src[(unsigned)tid] : 0;
            nv1[w] = tid + 512 < ncnt[w] ? (int)src[(unsigned)(tid + 512)] : 0;
        }
    }
    {
        // stride-2 'same' pads on the right only: output position n reads rows 2n, 2n + 1, 2n + 2 of
        // the image's rows 1 .. 32 (+ the zero row 33)
        unsigned a_addr[kGroup];
#pragma unroll
        for (int w4 = 0; w4 < kGroup; ++w4)
            a_addr[w4] = lds_addr(lds + cat_offset(w4 < group_n ? w4 : 0, group_n) + (1 + 2 * n) * kS192 + 2 * q + wave * 24);
        f4 acc[kGroup][3];
#pragma unroll
        for (int w4 = 0; w4 < kGroup; ++w4)
#pragma unroll
            for (int t = 0; t < 3; ++t) acc[w4][t] = f4{0.f, 0.f, 0.f, 0.f};
        // the twelve operand pairs of a tap (four windows x three channel groups of eight) are asked
        // for ONE TAP AHEAD from inline asm and waited for by count (as plain loads hipcc put sixteen
        // reads with an lgkmcnt(0) each in front of their first MFMAs)
        f2 a[2][kGroup][3];
        auto ask_tap = [&](auto tap_tag) {
            constexpr int TAP = decltype(tap_tag)::value;
#pragma unroll
            for (int w4 = 0; w4 < kGroup; ++w4) {
                a[TAP & 1][w4][0] = ds_read_f2<(TAP * kS192 + 0) * 4>(a_addr[w4]);
                a[TAP & 1][w4][1] = ds_read_f2<(TAP * kS192 + 8) * 4>(a_addr[w4]);
                a[TAP & 1][w4][2] = ds_read_f2<(TAP * kS192 + 16) * 4>(a_addr[w4]);
            }
        };
        auto run_tap = [&](auto tap_tag) {
            constexpr int TAP = decltype(tap_tag)::value;
            if constexpr (TAP + 1 < 3) {
                ask_tap(IntC<TAP + 1>{});
                asm volatile("s_waitcnt lgkmcnt(12)" ::: "memory");
            } else {
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            }
            f2(&at)[kGroup][3] = a[TAP & 1];
#pragma unroll
            for (int w4 = 0; w4 < kGroup; ++w4)
#pragma unroll
                for (int sp = 0; sp < 3; ++sp) asm volatile("" : "+v"(at[w4][sp]));
#pragma unroll
            for (int sp = 0; sp < 3; ++sp)
#pragma unroll
                for (int e = 0; e < 2; ++e)
#pragma unroll
                    for (int w4 = 0; w4 < kGroup; ++w4)
#pragma unroll
                        for (int t = 0; t < 3; ++t)
                            acc[w4][t] = mfma4(at[w4][sp][e], w17[(TAP * 3 + sp) * 3 + t][e], acc[w4][t]);
        };
        ask_tap(IntC<0>{});
        run_tap(IntC<0>{});
        run_tap(IntC<1>{});
        run_tap(IntC<2>{});
        phase_add(5, f_since);
        mark(ts, 41);
        lds_barrier();         // every wave has read the images: the partial tiles go over them
        phase_add(6, f_since);
        // the next window's first conv2 weights: slot 0 (tile 0 multiplies right behind that window's
        // first barrier, which retires these requests); the images that lay there are dead
        if (slot0_next) dma_weights<kWinoHalf, kWaves>(packed + weight_offset(1), lds + kSlot0, lane, wave);
        // tile (wave, window, N tile): 32 to a run (dbh_layout.h: red_tile_offset)
#pragma unroll
        for (int w4 = 0; w4 < kGroup; ++w4)
#pragma unroll
            for (int t = 0; t < 3; ++t) {
                const int tile = wave * 12 + w4 * 3 + t;
                const int off = (tile < 32 ? kRed : tile < 64 ? kRedB - kRedRun : kRedC - 2 * kRedRun) + tile * 256;
                *reinterpret_cast<f4*>(lds + off + lane * 4) = acc[w4][t];
            }
    }
    mark(ts, 42);
    lds_barrier();             // the partial tiles are out
    // the reduction: quadruple e of the group's 768 = the sum of its window's four partial tiles,
    // + bias, ReLU, BN6 -> the window's slot (dense [16][48], row 0 = position 0)
#pragma unroll
    for (int r = 0; r < 2; ++r) {
        const int e = r * kThreads + tid;
        const int win_k = e / 192, rest = e - win_k * 192, t = rest >> 6, l2 = rest & 63;
        if (e < kGroup * 192 && win_k < group_n) {
            f4 part[8];
#pragma unroll
            for (int w8 = 0; w8 < 8; ++w8) {
                const int tile = w8 * 12 + win_k * 3 + t;
                const int off = (tile < 32 ? kRed : tile < 64 ? kRedB - kRedRun : kRedC - 2 * kRedRun) + tile * 256;
                part[w8] = *reinterpret_cast<const f4*>(lds + off + l2 * 4);
            }
            f4 sum[1][1];
            sum[0][0] = ((part[0] + part[1]) + (part[2] + part[3])) + ((part[4] + part[5]) + (part[6] + part[7]));
            float* slot = wg_scratch + kWgTailOff + (tail_slot + win_k) * kTailSlotFloats;
            epilogue<1, 1, 48, false, true>(sum, slot + 4 * (l2 >> 4) * 48 + t * 16 + (l2 & 15), ep17[r]);
        }
    }
    phase_add(7, f_since);
    mark(ts, 43);
    // the next group's windows 1 .. 3: samples to the staging, exact sums for the barrier below
#pragma unroll
    for (int w = DBH_STATS0_IN_F ? 0 : 1; w < kGroup; ++w) {
        asm volatile("" : "+v"(nv0[w]), "+v"(nv1[w]));
        if (seam_b2 && w < next_n) {
            window_partial_sums(lds, ncnt[w], nv0[w], nv1[w], tid, lane, wave, w);
            short* smp = reinterpret_cast<short*>(lds + kStage + w * kStageWin);
            smp[tid] = (short)nv0[w];
            smp[tid + 512] = (short)nv1[w];
        }
    }
    // the batch's weights (conv18, conv19, conv20): requested now - the partial tiles they land on
    // are read - and used behind two barriers
    tail_slot += group_n;
    lds_barrier();
    // conv2's other two thirds for the next window (slots 1 and 2: the partial tiles that lay there
    // are summed) - unless the batched tail, whose buffers lie there, runs in between
    if (thirds_now) dma_weights<2 * kWinoHalf, kWaves>(packed + weight_offset(1) + kWinoHalf, lds + kSlot1, lane, wave);
    if (batch_ends) {
        dma_weights<conv_weight_floats(17)>(packed + weight_offset(17), lds + kTW18, lane, wave);
        dma_weights<conv_weight_floats(18)>(packed + weight_offset(18), lds + kTW19, lane, wave);
        dma_weights<conv_weight_floats(19)>(packed + weight_offset(19), lds + kTW20, lane, wave);
    }
    if (wave >= (DBH_STATS0_IN_F ? 4 : 5)) {
        const int w = wave - 4;
        if (seam_b2 && w < next_n) {
            double mean, inv;
            if (DBH_STATS0_IN_F) {
                const int c = w == 0 ? ncnt[0] : w == 1 ? ncnt[1] : w == 2 ? ncnt[2] : ncnt[3];
                window_mean_inv(lds, c, &mean, &inv, w);
            } else {
                window_mean_inv(lds, ncnt[1], &mean, &inv, 1);
                if (w == 2) window_mean_inv(lds, ncnt[2], &mean, &inv, 2);
                if (w == 3) window_mean_inv(lds, ncnt[3], &mean, &inv, 3);
            }
            float* st = lds + kStageStats + w * 8;
            if (lane == 0) {
                reinterpret_cast<double*>(st)[0] = mean;
                reinterpret_cast<double*>(st)[1] = inv;
                reinterpret_cast<int*>(st)[4] = w == 0 ? ncnt[0] : w == 1 ? ncnt[1] : w == 2 ? ncnt[2] : ncnt[3];
                reinterpret_cast<int*>(st)[5] = w == 0 ? npad[0] : w == 1 ? npad[1] : w == 2 ? npad[2] : npad[3];
            }
        }
    }
    phase_add(8, f_since);
    mark_realtime(ts, 63);
    phase_stamp(3);
    if (!DBH_STATS0_IN_F) {
        carry_cnt = ncnt[0];
        carry_pad = npad[0];
        carry_v0 = nv0[0];
        carry_v1 = nv1[0];
    }

    // ---------------- stages G + H for the batch: conv18, conv19 (+ MaxPool + BN7), conv20 (1x1 ->
    // classes) + ReLU + GlobalAveragePool + Softmax (+ renormalise + call), one wave per window ---
    if (run_tail && tail_slot > 0) {
        const int n = lane & 15, q = lane >> 4;
        const int n_batch = tail_slot;
        tail_slot = 0;
        ArgsPtr a = args();
        const int n_classes = a->n_classes;
        const bool mine = wave < n_batch;
        long my_win = 0;         // (read behind the barrier below)
        float* X = lds + tail_x_offset(wave);
        float* Y = X + kTailBuf;
        // epilogue parameters of the three layers, asked for before anything waits
        EpiParams<3, false> ep18;
        EpiParams<3, true> ep19;
        ep18.load(packed + bias_offset(17) + n, nullptr, nullptr);
        ep19.load(packed + bias_offset(18) + n, packed + bn_scale_offset(6) + n,
                  packed + bn_shift_offset(6) + n);
        const float bias20a = packed[bias_offset(19) + n];
        const float bias20b = packed[bias_offset(19) + 16 + n];
        full_barrier();        // conv17's stores of the last window are out; its partial tiles / the concat buffer free
        mark(ts, 45);
        my_win = (long)reinterpret_cast<const int*>(lds + kTailWins)[mine ? wave : 0];
        if (mine) {
            const float* src = wg_scratch + kWgTailOff + wave * kTailSlotFloats;
            f4 v[3];
#pragma unroll
            for (int i = 0; i < 3; ++i) v[i] = *reinterpret_cast<const f4*>(src + i * 256 + lane * 4);
            if (lane < kS48) {       // zero rows before position 0 and after position 15, X and Y
                X[lane] = 0.f;
                X[17 * kS48 + lane] = 0.f;
                Y[lane] = 0.f;
                Y[17 * kS48 + lane] = 0.f;
            }
#pragma unroll
            for (int i = 0; i < 3; ++i) {
                const int idx = i * 256 + lane * 4, row = idx / 48, ch = idx - row * 48;
                float* dst = X + (1 + row) * kS48 + ch;        // 8-byte aligned
                *reinterpret_cast<f2*>(dst) = f2{v[i].x, v[i].y};
                *reinterpret_cast<f2*>(dst + 2) = f2{v[i].z, v[i].w};
            }
        }
        full_barrier();        // the batch's weights have landed (vmcnt(0) rides on the barrier)
        mark(ts, 46);
        bool stop_here = false;
        if (debug_stage >= 0 && stop_stage == 5) {
            if (debug_stage < 100 && mine)      // (each wave its own window's 16 x 48)
                for (int idx = lane; idx < 16 * 48; idx += 64)
                    glob(a->debug_out)[my_win * kStageFloats[5] + idx] =
                        X[(idx / 48 + 1) * kS48 + idx % 48] * kActUnscale;
            stop_here = true;
        }
        if (mine && !stop_here) {
            // conv18: 16 positions x 48 -> 48, k = 3, bias + ReLU -> Y
            {
                f4 acc[1][3];
                bias_acc(acc, ep18);
                conv_tiles<3, 6, 6, 1, 3, 3, kS48, 16>(X + n * kS48 + 2 * q,
                                                       lds + kTW18 + lane * 2, acc);
                epilogue<1, 3, kS48, false, false, false>(acc, Y + (1 + 4 * q) * kS48 + n, ep18);
            }
            mark(ts, 47);
            // conv19 + MaxPool + BN7 -> 8 positions, into X (rows 9.. keep conv17's values: the
            // positions 8..15 conv20's tile computes from them are never looked at)
            {
                f4 acc[1][3];
                bias_acc(acc, ep19);
                conv_tiles<3, 6, 6, 1, 3, 3, kS48, 16>(Y + n * kS48 + 2 * q,
                                                       lds + kTW19 + lane * 2, acc);
                epilogue<1, 3, kS48, true, true, false>(acc, X + (1 + 2 * q) * kS48 + n, ep19);
            }
            mark(ts, 49);
        }
        if (debug_stage >= 0 && stop_stage == 6) {
            if (debug_stage < 100 && mine)
                for (int idx = lane; idx < 8 * 48; idx += 64)
                    glob(a->debug_out)[my_win * kStageFloats[6] + idx] =
                        X[(idx / 48 + 1) * kS48 + idx % 48] * kActUnscale;
            stop_here = true;
        }
        if (mine && !stop_here) {
            // conv20 (1x1 -> classes, one or two N tiles) + ReLU + global average: the logit of
            // class 16t + n ends up in the q = 0 lanes
            auto conv20_logit = [&](int t, float bias) -> float {
                const float* a_lane = X + (n + 1) * kS48 + 2 * q;
                const float* b_lane = lds + kTW20 + t * 128 + lane * 2;
                f4 acc0 = f4{0.f, 0.f, 0.f, 0.f}, acc1 = f4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int sp = 0; sp < 6; ++sp) {
                    const f2 av = *reinterpret_cast<const f2*>(a_lane + sp * 8);
                    const f2 bv = *reinterpret_cast<const f2*>(b_lane + sp * 2 * 128);
                    acc0 = mfma4(av.x, bv.x, acc0);
                    acc1 = mfma4(av.y, bv.y, acc1);
                }
                const f4 acc = acc0 + acc1;
                float sum = 0.f;
                if (q < 2)     // rows 4q..4q+3 of the tile; only positions 0..7 exist
                    sum = fmaxf(acc.x + bias, 0.f) + fmaxf(acc.y + bias, 0.f) +
                          fmaxf(acc.z + bias, 0.f) + fmaxf(acc.w + bias, 0.f);
                int even, odd;     // rows q = 0 and q = 1 hold the two halves of the position sum
                rows_i32(__builtin_bit_cast(int, sum), &even, &odd);
                // the mean over the 8 positions, and out of the kernel's activation scale
                return (__builtin_bit_cast(float, even) + __builtin_bit_cast(float, odd)) *
                       (0.125f * kActUnscale);
            };
            float logit = conv20_logit(0, bias20a);        // classes 0..15 in lanes 0..15
            if (n_classes > 16) {
                // classes 16..31: through the wave's own 32 LDS words into lanes 16..31
                const float upper = conv20_logit(1, bias20b);
                float* wl = lds + kTLog + wave * 32;
                if (q == 0) wl[16 + n] = upper;
                const float moved = wl[lane & 31];
                if (lane >= 16) logit = moved;
            }
            const bool valid = lane < n_classes;
            const float v = valid ? logit : -INFINITY;
            // classes live in lanes 0..31 = two 16-lane rows
            const float rmx = row16_max(v);
            const float mx = fmaxf(lane_value(rmx, 0), lane_value(rmx, 16));
            const float e = valid ? expf(v - mx) : 0.f;
            const float rsum = row16_sum(e);
            const float sum = lane_value(rsum, 0) + lane_value(rsum, 16);
            if (debug_stage == 7) {
                if (lane < 32) glob(a->debug_out)[my_win * kStageFloats[7] + lane] = valid ? v : 0.f;
            } else {
                const float p = e / sum;
                float* __restrict__ probs = glob(a->probs);
                int* __restrict__ calls = glob(a->calls);
                if (calls != nullptr) {
                    // single scan step: this window IS the read (classify.py:368-382, one range)
                    if (lane < 32)
                        renormalise_and_call(valid ? p : 0.f, lane, n_classes, a->score_diff,
                                             probs + my_win * n_classes, calls + my_win);
                } else if (valid) {
                    probs[my_win * n_classes + lane] = p;
                }
            }
        }
        mark(ts, 53);
        full_barrier();        // the next group's stage A writes over all of this
        mark(ts, 55);
    }
    flush_marks(ts, ts_out, lane);
    phase_stamp(4);
    thirds_ahead = thirds_next;
    }   // (not stop_stage 3)
    }   // (not stop_stage 2)
    }   // (not stop_stage 0, 1)

    first_group = false;
    ++phase_group;
    staged = samples_entry != nullptr && next_n > 0 && (stop_stage < 0 || stop_stage > 4);
    if (!staged) thirds_ahead = false;
    group_start = next_start;
    group_n = next_n;
    }   // persistent loop over this workgroup's groups of windows
    // the launch's last workgroup puts the counter (and the count of finished workgroups beside
    // it) back to zero for the next launch on this stream
    if (win_counter_entry != nullptr && threadIdx.x == 0) {
        __threadfence();
        if (atomicAdd(win_counter_entry + 1, 1) == (int)gridDim.x - 1) {
            win_counter_entry[0] = 0;
            win_counter_entry[1] = 0;
        }
    }
    if (args()->clock_out != nullptr) {
        long long* c = glob(args()->clock_out) + (size_t)blockIdx.x * (4 + kPhaseMarks * kPhaseGroups);
        if (threadIdx.x == 0) {
            c[2] = (long long)__builtin_readcyclecounter();
            c[3] = (long long)__builtin_amdgcn_s_memrealtime();
        }
        __syncthreads();
        for (int i = threadIdx.x; i < kPhaseMarks * kPhaseGroups; i += kThreads)
            c[4 + i] = i < phase_group * kPhaseMarks ? (long long)reinterpret_cast<const unsigned*>(lds + kPhase)[i] : -1;
    }
}

// =============================================================================================
// Seam b2 helpers.
// =============================================================================================

// One block per (read, step): slice the window (classify.py:337-349), z-normalise it in fp64
// (trim_signal.py:61-69; the sums are exact integers), zero-pad right ('start') or left ('end')
// (classify.py:352-357) and emit fp32, which is what Keras casts the float64 input to.
__global__ __launch_bounds__(256) void dbh_normalise_kernel(
    const int16_t* __restrict__ samples, const long long* __restrict__ offsets, int steps,
    int side, float* __restrict__ windows) {
    __shared__ long long red[2][4];
    const long long read = blockIdx.x / steps;
    const int step = blockIdx.x - (int)(read * steps);
    const long long base = offsets[read];
    const long long len = offsets[read + 1] - base;
    long long a, b;
    window_bounds(len, step, side, &a, &b);
    const int cnt = (int)(b - a);
    const int tid = threadIdx.x;
    const int16_t* src = samples + base + a;

    int v[4];
    long long s1 = 0, s2 = 0;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int k = tid + i * 256;
        v[i] = (k < cnt) ? (int)src[k] : 0;
        s1 += v[i];
        s2 += (long long)v[i] * v[i];
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
        s1 += __shfl_xor(s1, off);
        s2 += __shfl_xor(s2, off);
    }
    if ((tid & 63) == 0) {
        red[0][tid >> 6] = s1;
        red[1][tid >> 6] = s2;
    }
    __syncthreads();
    s1 = red[0][0] + red[0][1] + red[0][2] + red[0][3];
    s2 = red[1][0] + red[1][1] + red[1][2] + red[1][3];

    float* out = windows + (long long)blockIdx.x * kWindow;
    double mean, inv;
    mean_std(s1, s2, cnt, &mean, &inv);
    const int pad_left = (side == 0) ? 0 : kWindow - cnt;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int k = tid + i * 256;          // position in the source slice
        if (k < cnt) out[pad_left + k] = (float)(((double)v[i] - mean) * inv);
    }
    // zero padding: [cnt, 1024) for 'start', [0, 1024-cnt) for 'end'
    const int pad_begin = (side == 0) ? cnt : 0;
    const int pad_count = kWindow - cnt;
    for (int k = tid; k < pad_count; k += 256) out[pad_begin + k] = 0.f;
}

// 32 lanes per read: merge the per-step softmax vectors (classify.py:368-374: min for class 0,
// max for the barcodes), rescale the barcodes so the vector sums to one (classify.py:387-393,
// in fp64 like NumPy-1.x scalar promotion did) and make the call (classify.py:285-295).
__global__ __launch_bounds__(256) void dbh_merge_kernel(const float* __restrict__ wprobs,
                                                        long long n_reads, int steps,
                                                        int n_classes, double score_diff,
                                                        float* __restrict__ probs,
                                                        int* __restrict__ calls) {
    const long long read = (long long)blockIdx.x * 8 + (threadIdx.x >> 5);
    const int c = threadIdx.x & 31;
    if (read >= n_reads) return;   // whole 32-lane groups exit together
    const bool valid = c < n_classes;
    float merged = 0.f;
    if (valid) {
        const float* src = wprobs + read * steps * n_classes + c;
        merged = src[0];
        for (int s = 1; s < steps; ++s) {
            const float v = src[(long long)s * n_classes];
            merged = (c == 0) ? fminf(merged, v) : fmaxf(merged, v);
        }
    }
    renormalise_and_call(merged, c, n_classes, score_diff, probs + read * n_classes,
                         calls + read);
}

}  // namespace DBH_FORWARD_NS
