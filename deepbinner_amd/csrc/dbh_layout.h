// dbh_layout.h — compile-time description of the Deepbinner network
// (reference deepbinner/network_architecture.py:18-95) as the forward kernel sees it:
// which convolutions run on the MFMA path, where their pre-swizzled "fragment order" weights
// live in the packed HBM buffer, and how the LDS arena is carved per stage.
//
// Shared by the host-side packer (dbh_api.hip) and the device code (dbh_forward.hip).
#pragma once

// A/B switch (tools/ab_variants.sh): BN2 folded into conv1d_5's weights and bias by the packer
#ifndef DBH_FOLD_BN2
#define DBH_FOLD_BN2 1
#endif

namespace dbh {

constexpr int kWindow = 1024;          // model input size (classify.py:96)
constexpr int kMaxClasses = 32;        // conv1d_20 is padded to two 16-wide N tiles
constexpr int kNumConvs = 20;
constexpr int kNumBn = 7;

// ---------------------------------------------------------------------------------------------
// Convolution table (index = Keras layer number - 1).  cout_pad is C_out rounded up to 16.
// ---------------------------------------------------------------------------------------------
// wino = 2: the layer runs as Winograd F(2,3) — two outputs per position pair from FOUR
//   element-wise-transformed products instead of six (1.5x fewer MFMAs); weights are stored as
//   V0 = g0, V1 = (g0+g1+g2)/2, V2 = (g0-g1+g2)/2, V3 = g2.
// wino = 4: Winograd F(4,3) — four outputs per position quad from SIX products instead of
//   twelve (2x fewer MFMAs); weights are stored as V0 = g0/4, V1 = -(g0+g1+g2)/6,
//   V2 = -(g0-g1+g2)/6, V3 = g0/24+g1/12+g2/6, V4 = g0/24-g1/12+g2/6, V5 = g2.
// wino = 0: direct convolution.
struct ConvSpec { int taps, cin, cout_pad, stride; int wino; };
constexpr ConvSpec kConv[kNumConvs] = {
    {3, 1, 48, 2, 0},     // conv1d_1   (K = 3 padded to one MFMA k-step)
    {3, 48, 48, 1, 4},        // conv1d_2
    {3, 48, 48, 1, 4},        // conv1d_3
    {3, 48, 48, 1, 4},        // conv1d_4
    {1, 48, 16, 1, 0},    // conv1d_5
    {3, 16, 48, 1, 2},        // conv1d_6
    {3, 48, 48, 1, 4},        // conv1d_7
    {3, 48, 48, 1, 4},        // conv1d_8   (round 6: F(4,3), four windows at a time - stage_d_chain)
    {3, 48, 48, 1, 4},        // conv1d_9
    {1, 48, 48, 1, 0},    // conv1d_10
    {1, 48, 48, 1, 0},    // conv1d_11
    {1, 48, 16, 1, 0},    // conv1d_12
    {3, 16, 48, 1, 2},        // conv1d_13
    {1, 48, 16, 1, 0},    // conv1d_14
    {3, 16, 48, 1, 2},        // conv1d_15
    {3, 48, 48, 1, 2},    // conv1d_16  (round 6: F(2,3) on the chain's position pairs)
    {3, 192, 48, 2, 0},   // conv1d_17
    {3, 48, 48, 1, 0},    // conv1d_18
    {3, 48, 48, 1, 0},    // conv1d_19
    {1, 48, 32, 1, 0},    // conv1d_20  (n_classes <= 32, zero padded)
};
constexpr int kBnChannels[kNumBn] = {48, 48, 48, 48, 192, 48, 48};
// F(2,3) layers whose weights are stored by N tile ([t][sp][matrix pair][lane][matrix][e]) for the
// N-tile-outer loops of dbh_forward.hip (conv1d_6, conv1d_13, conv1d_15, conv1d_16).
constexpr bool wino2_by_tile(int i) { return i == 5 || i == 12 || i == 14 || i == 15; }
// positions each convolution produces (after its stride, before any pooling)
constexpr int kConvLout[kNumConvs] = {512, 512, 512, 512, 256, 256, 256, 128, 128, 64,
                                      64,  64,  64,  64,  64,  64,  16,  16,  16,  8};

// v_mfma_f32_16x16x4_f32 instructions the forward kernel ISSUES for convolution i of one window
// (2,048 FLOP each): M tiles of 16 positions (pairs / quads for the Winograd layers) x N tiles of
// 16 channels x k-steps of 4 input channels per tap (transformed matrix).  The direct count of
// the same layer is what SURVEY.md's 33,629,952 FLOP per window adds up; the Winograd layers
// issue 4/6 and 6/12 of it.
constexpr int conv_mfmas(int i, int n_classes) {
    // conv1d_1 is computed inside conv1d_2's first tile, transposed (channels x positions) and with
    // the halo rows of every quad tile recomputed: per wave 6 input rows x 3 channel groups
    if (i == 0) return 8 * 6 * 3;
    const int units = kConvLout[i] / (kConv[i].wino ? kConv[i].wino : 1);
    const int m_tiles = (units + 15) / 16;
    const int n_tiles = i == kNumConvs - 1 ? (n_classes <= 16 ? 1 : 2) : kConv[i].cout_pad / 16;
    const int mats = kConv[i].wino == 4 ? 6 : kConv[i].wino == 2 ? 4 : kConv[i].taps;
    const int k_steps = kConv[i].cin == 1 ? 1 : mats * kConv[i].cin / 4;
    return m_tiles * n_tiles * k_steps;
}
constexpr int forward_mfmas(int n_classes) {
    int n = 0;
    for (int i = 0; i < kNumConvs; ++i) n += conv_mfmas(i, n_classes);
    return n;
}
static_assert(forward_mfmas(13) == 9156, "MFMA count per window (SQ_INSTS_MFMA; 9,588 until conv1d_8/9 became F(4,3) and conv1d_16 F(2,3))");

// Number of floats of fragment-ordered weights of conv layer i (0-based); conv1 keeps [3][48].
constexpr int conv_weight_floats(int i) {
    return (kConv[i].wino == 4 ? 6 : kConv[i].wino == 2 ? 4 : kConv[i].taps) * kConv[i].cin *
           kConv[i].cout_pad;
}

// ---------------------------------------------------------------------------------------------
// THE ACTIVATION SCALE.  Inside the forward kernel every activation is held as its value times
// kActScale = 2^-60: the input is scaled on the way in, biases and BN shifts are packed scaled,
// the logits are scaled back before the softmax.  A power of two commutes with every fp32
// rounding (nothing here comes near the subnormal range or overflow), so all values are exactly
// 2^-60 times what they would be - and ReLU becomes the `clamp` output modifier ([0, 1]: no
// activation of this network comes near 2^60 - the largest, before BN2, are ~1e8) of whatever instruction produces the value: the
// output transforms of the Winograd layers end in a packed add or fma, and there is no packed
// fp32 max in the ISA, so this takes a third of their epilogues' vector instructions away.
// ---------------------------------------------------------------------------------------------
constexpr float kActScale = 1.f / 1152921504606846976.f;       // 2^-60
constexpr float kActUnscale = 1152921504606846976.f;

// ---------------------------------------------------------------------------------------------
// Packed parameter buffer (floats).  [ weights of conv 1..20 | bias of conv 1..20 (cout_pad
// each) | BN scale,shift of bn 1..7 ].  BN is pre-folded on the host in fp64:
//   scale = gamma / sqrt(var + 1e-3),  shift = beta - mean * scale.
// Biases and shifts are stored times kActScale (see above).
// ---------------------------------------------------------------------------------------------
constexpr int weight_offset(int i) {
    int off = 0;
    for (int j = 0; j < i; ++j) off += conv_weight_floats(j);
    return off;
}
constexpr int kWeightFloats = weight_offset(kNumConvs);
constexpr int bias_offset(int i) {
    int off = kWeightFloats;
    for (int j = 0; j < i; ++j) off += kConv[j].cout_pad;
    return off;
}
constexpr int kBiasEnd = bias_offset(kNumConvs);
constexpr int bn_scale_offset(int i) {
    int off = kBiasEnd;
    for (int j = 0; j < i; ++j) off += 2 * kBnChannels[j];
    return off;
}
constexpr int bn_shift_offset(int i) { return bn_scale_offset(i) + kBnChannels[i]; }
constexpr int kPackedFloats = bn_scale_offset(kNumBn);

// Fragment order of an MFMA-path layer (v_mfma_f32_16x16x4_f32, B operand = weights):
//   index = (((tap * SP + sp) * NT + t) * 64 + lane) * 2 + e
//   value = W[tap][8*sp + 2*(lane>>4) + e][16*t + (lane&15)]
// with SP = C_in/8 and NT = cout_pad/16 (for a Winograd layer "tap" runs over the four
// transformed matrices V0..V3).  One ds_read_b64 / global_load_dwordx2 per lane yields
// the B fragments of two consecutive k-steps; the A side (activations) pairs channels the same
// way, so the contraction order is a fixed permutation of (tap, c_in).

// conv1d_2 is the exception: its input never exists as an LDS image.  conv1d_1 runs as the
// TRANSPOSED product (M = 16 output channels, N = 16 positions) inside conv1d_2's first tile, and
// an MFMA leaves lane l with rows 4*(l>>4) .. +3 of its result = output channels 16g + 4*(l>>4) + r
// (g = channel group of the MFMA, r = result register).  Those registers, ReLU'd, batch-normalised
// and Winograd-transformed in place, ARE conv1d_2's A fragments if k-step s = 4g + r of conv1d_2
// contracts over channels {16g + 4q + r : q = 0..3} - a permutation of the input channels that
// only the packer needs to know about (dbh_api.hip: pack_weights).
// conv1d_3 and conv1d_4 take the same order (round 5): all of stage B runs in that TRANSPOSED
// orientation (M = 16 output channels, N = 16 quads), so that a layer's accumulators - lane
// (quad, q), registers = channels 16t + 4q + r - are, after the output transform, ReLU and the next
// input transform (all in-lane but for one halo position each side), the next layer's B operand.
// (conv1d_5, a 1x1 convolution on conv1d_4's pooled output, is fed from the same registers)
// ... and conv1d_6 from conv1d_5's accumulators
// conv1d_8 and conv1d_9 (round 6) run the same way for FOUR windows at a time (stage_d_chain): a
// wave owns half a window (16 quads of its 128 positions) from conv1d_7's parked output to BN4.
// ... and the inception block behind them (conv1d_10 .. conv1d_16) on the same lanes' position pairs.
constexpr bool chained(int conv) { return (conv >= 1 && conv <= 5) || (conv >= 7 && conv <= 15); }
constexpr int frag_cin(int conv, int sp, int q, int e) {
    return chained(conv) ? 16 * ((2 * sp + e) >> 2) + 4 * q + ((2 * sp + e) & 3) : 8 * sp + 2 * q + e;
}

// ---------------------------------------------------------------------------------------------
// LDS arena (floats).  Activations are [position][channel] with a padded row stride and one zero
// row before and after the data ("physical row = logical position + 1") to serve 'same' padding.
// The stride decides the bank pattern of the 16 rows an A-fragment ds_read_b64 touches, which
// are 1, 2 or 4 rows apart (direct / Winograd F(2,3) / F(4,3)): for the 48-channel buffers 50
// floats gives distinct bank groups at row steps 1 and 2 and a 2-way conflict at step 4 (52 is
// clean at step 1 but 4-way at step 4, which made LDS the limiter of the F(4,3) layers).
// ---------------------------------------------------------------------------------------------
constexpr int kS48 = 50;
constexpr int kS16 = 20;
constexpr int kS192 = 196;

// stages A-D: one in-place activation buffer + a 54 KiB weight area filled by LDS-DMA
// (global_load_lds_dwordx4) ahead of use.  Direct layers see it as two 6,912-float buffers
// (kW0, kW1); the Winograd layers of stage B as three 4,608-float slots (two transformed
// matrices each) rotated so that the half a layer needs next is always already in flight.
constexpr int kActOff = 0;
constexpr int kActFloats = (512 + 2) * kS48;               // 25,700
constexpr int kWFloats = 3 * 48 * 48;                      // 6,912
constexpr int kW0 = kActOff + kActFloats;
constexpr int kW1 = kW0 + kWFloats;
constexpr int kLdsFloatsAD = kW1 + kWFloats;               // 39,524
constexpr int kWinoHalf = 2 * 48 * 48;                     // 4,608: two transformed matrices
// An F(4,3) layer's image is three thirds of kWinoHalf floats, third t = everything N tile t
// (output channels 16t..16t+15) needs: [sp 0..5][matrix pair 0..2][lane][matrix of the pair][e].
constexpr int kSlot0 = kW0;
constexpr int kSlot1 = kW0 + kWinoHalf;
constexpr int kSlot2 = kW0 + 2 * kWinoHalf;
static_assert(kSlot2 + kWinoHalf == kLdsFloatsAD, "three Winograd slots fill the weight area");
// stage C onwards the activations need at most 258 rows, so the upper half of the activation
// buffer doubles as one more weight buffer
constexpr int kUpper = 258 * kS48;                          // 12,900
static_assert(kUpper + 4 * 48 * 48 <= kW0, "upper weight buffer must stay below the weight area");
// ... and, before that, as the home of conv5's 16-channel output (258 rows x kS16)
constexpr int kMid16 = kUpper;
static_assert(kMid16 + 258 * kS16 <= kW0, "");
// conv7's six F(4,3) matrices take the whole weight area (thirds by N tile in slots 0..2, like a
// stage-B layer's: requested as conv4 leaves each slot); conv5's and conv6's weights live in the
// upper buffer behind conv5's output (rows conv4 has finished reading by its mid-layer barrier)
constexpr int kW5 = kMid16 + 258 * kS16;     // 18,060
constexpr int kW6 = kW5 + 1 * 48 * 16;
static_assert((kW5 * 4) % 16 == 0 && (kW6 * 4) % 16 == 0 && kW6 + 4 * 16 * 48 <= kW0, "");
// Stage B chained in registers (dbh_forward.hip: stage_b_chain): the activation buffer is idle from
// the top of a window until conv1d_4's pooled outputs arrive, so conv1d_3's three thirds wait there
// (a ring of six slots with the weight area: conv2 in slots 0-2, conv3 here, conv4 in slots 0-2
// again as conv2 leaves them) ...
constexpr int kChainW3 = kActOff;
static_assert((kChainW3 * 4) % 16 == 0 && kChainW3 + 3 * kWinoHalf <= kW5, "");
// ... and the halo rows the waves hand each other between two layers live above conv6's weights:
// per layer output (conv2's, conv3's) two arrays [9 rows][2 sides][48 channels] - position 0 of a
// wave's first quad and position 3 of its last, written by its lanes n = 0 (side 0) and n = 15
// (side 1); A0 row w = wave w's, A3 row w + 1 = wave w's.  Wave w reads its right halo from
// A0[w + 1][0] and its left halo from A3[w][1]; A0[8][0] and A3[0][1] stay zero = 'same' padding.
constexpr int kHaloRows = 9 * 2 * 48;                        // 864 floats per array
constexpr int kHalo = kW6 + 4 * 16 * 48;                     // 21,900
// 1 KB of scratch behind them: where the lanes that have nothing to store or to post point their
// share of an unmasked LDS instruction (floats: [0, 176) stores, [176, 240) posts, [240, 312) arrivals)
constexpr int kChainDummy = kHalo + 4 * kHaloRows;           // 25,356
static_assert((kHalo * 4) % 16 == 0 && kChainDummy % 2 == 0 && kChainDummy + 312 <= kW0, "");
// conv7's pair exchange: two f4 per lane and wave, in activation rows its pooled output leaves free
constexpr int kX7 = 130 * kS48;
static_assert((kX7 * 4) % 16 == 0 && kX7 + 8 * 512 <= 258 * kS48, "");
// ---------------------------------------------------------------------------------------------
// GROUPS OF FOUR WINDOWS (round 6).  A workgroup takes kGroup windows at a time:
//   [stage A, B, C] x kGroup   conv1d_7's pooled + BN3 output (128 x 48) is PARKED in global memory
//   stage D once               conv1d_8, conv1d_9 as F(4,3) for the four windows together: 32 quads
//                              per window = two tiles of 16 = one per wave of a wave pair, eight
//                              tiles for eight waves, a chain in registers like stage B
//                              ... and stage E (the inception block) on its end, on registers
//   [stage F] x kGroup         the block's outputs (32 x 192 + zero rows each) wait in LDS
// Per-workgroup scratch in global memory (floats): conv1d_17's outputs for the batched tail, then
// the two parks.
// ---------------------------------------------------------------------------------------------
constexpr int kGroup = 4;
constexpr int kPark7Floats = 128 * 48;                      // [half][row of quad][channel group][lane][4]
// the group's LAST window skips the park: stage D follows its conv1d_7 at once, so its output goes
// to LDS in the same layout - over rows of conv1d_7's input image (all read by its mid-layer barrier),
// below its pair exchange; the waves that own the window in stage D read it back behind the layer's
// closing barrier (stage D's weights for stage E land there two tiles later)
constexpr int kPark7Lds = 0;
static_assert(kPark7Lds + kPark7Floats <= kX7, "");

// Samples of the NEXT group's windows (seam b2), staged while this group runs stages E and F:
// 1,024 int16 per window + {mean, 1/std (doubles), count, left padding}.  Always live, so it sits
// in a hole every stage's plan leaves free: above conv1d_3's ring of weights in stages A-C.
constexpr int kStage = 3 * kWinoHalf;                       // 13,824
constexpr int kStageWin = kWindow / 2;                      // floats per staged window
constexpr int kStageStats = kStage + kGroup * kStageWin;    // 8 floats per window
constexpr int kStageEnd = kStageStats + kGroup * 8;         // 15,904
static_assert(kChainW3 + 3 * kWinoHalf <= kStage && kStageEnd <= kW5 && kStageStats % 2 == 0, "");
static_assert(258 * kS48 <= kStage, "conv1d_6's image (conv1d_7's input) must stay below the staging");

// stage D (stage_d_chain): weights in four slots of one third each (conv1d_8's thirds in slots
// 0-2, conv1d_9's in slot 3 and - as conv1d_8 leaves them - slots 0 and 1), halo rows, scratch.
// Slots 0 and 1 are requested while the group's last conv1d_7 runs, before its mid-layer barrier:
// both lie in the idle part of the activation buffer, between the staging and conv1d_7's weights.
constexpr int kDS0 = kStageEnd;                              // 15,904
constexpr int kDS1 = kDS0 + kWinoHalf, kDS2 = kDS1 + kWinoHalf, kDS3 = kDS2 + kWinoHalf;
// [layer output 0..1][wave][side: 0 = the wave's left halo row, 1 = its right one][48]
constexpr int kDHaloLayer = 8 * 2 * 48;               // 768
constexpr int kDHalo = kDS3 + kWinoHalf;                    // 37,408
constexpr int kDDummy = kDHalo + 2 * kDHaloLayer;           // 38,944 (+ 312: see kChainDummy)
constexpr int kLdsFloatsD = kDDummy + 312;
static_assert((kDS0 * 4) % 16 == 0 && kDS0 >= kStageEnd && kDDummy % 2 == 0, "");
static_assert(kDS1 + kWinoHalf <= kW0, "slots 0 and 1 must be clear of conv1d_7's weights");
// The same place serves stage B of a group's LATER windows: the first third of conv1d_2's weights
// (what its tile 0 multiplies by) is requested there under conv1d_7 of the window before - in that
// layer's first half, when slot 0 itself still holds conv1d_7's own weights.  (Stage B's own use of
// the region - conv1d_5's and conv1d_6's weights, the halo rows - begins after tile 0.)
constexpr int kChainP = kStageEnd;
static_assert(kChainP + kWinoHalf <= kHalo && kChainP >= kStageEnd, "");

// stage E (inception block) runs on the END of stage D's chain, on the same lanes' registers
// (dbh_forward.hip: stage_d_chain): BN4's output never leaves them.  Its weights - conv10 .. conv15,
// one contiguous run of the packed image - land at the front of the arena while stage D runs (from
// its tile 2 on: the group's last window has its conv7 output there until its two waves have read
// it), conv16's four F(2,3) matrices in stage D's slots 2 and 3 as it leaves them.
constexpr int kEW10 = 0;
constexpr int kEW11 = kEW10 + conv_weight_floats(9);
constexpr int kEW12 = kEW11 + conv_weight_floats(10);
constexpr int kEW13 = kEW12 + conv_weight_floats(11);
constexpr int kEW14 = kEW13 + conv_weight_floats(12);
constexpr int kEW15 = kEW14 + conv_weight_floats(13);
constexpr int kEWEnd = kEW15 + conv_weight_floats(14);     // 12,288
constexpr int kEW16 = kDS2;                                // 9,216 floats = slots 2 and 3
static_assert(conv_weight_floats(15) == 2 * kWinoHalf && kEWEnd == weight_offset(15) - weight_offset(9), "");
// halo rows of the two 16-channel bottlenecks (conv12's and conv14's outputs: [wave][side][Z3 16 | Z4 16]);
// the 48-channel ones (conv10's products, conv15's output) reuse stage D's two arrays
constexpr int kEHaloZ = kEWEnd;                            // 512 floats
// BN5's scale and shift (2 x 192 floats + what follows them in the packed image: 512 by DMA)
constexpr int kEBn5 = kEHaloZ + 8 * 2 * 32;                // 12,800
static_assert(kEBn5 + 512 <= kStage && (kEBn5 * 4) % 16 == 0, "");
// The block's output - pooled + BN5 concat, 34 rows x 196 with the two zero rows conv1d_17's
// 'same' padding reads - is what stage F works on, one window after the other.  All four images
// wait in LDS: the chain keeps a lane's 48 outputs in registers to its end and writes them behind
// a barrier, over its own dead weights.  The group's LAST window's image lies below the staging
// (while its conv1d_17 runs, the next window's conv1d_2 weights land in slots 0..2), the others'
// above it.
constexpr int kCatFloats = 34 * kS192;                     // 6,664
constexpr int kCatLast = kStage - kCatFloats - 248;        // 6,912
constexpr int kCatHigh = kStageEnd;                        // 15,904: + k * kCatFloats for window k
constexpr int cat_offset(int k, int group_n) { return k == group_n - 1 ? kCatLast : kCatHigh + k * kCatFloats; }
// conv1d_17's split-K partial tiles: 96 of 256 floats (eight waves' eighths of the contraction x
// four windows x three N tiles), over the dead images in three runs of 32 that keep clear of the
// staging of the next group's samples and of slot 0 (the next window's first weights, on their way
// in while the tiles are summed): tile i lies at red_tile_offset(i)
constexpr int kRed = 0;
constexpr int kRedFloats = 24 * 256;        // (where the batched tail's first weights begin)
constexpr int kRedRun = 32 * 256;
constexpr int kRedB = kStageEnd, kRedC = kW0 + kWinoHalf;
constexpr int red_tile_offset(int i) {
    return (i < 32 ? kRed : i < 64 ? kRedB - kRedRun : kRedC - 2 * kRedRun) + i * 256;
}
static_assert(kRed + kRedRun <= kStage && kRedB + kRedRun <= kW0 && kRedC + kRedRun <= kLdsFloatsAD, "");
static_assert(kCatLast + kCatFloats <= kStage && (kCatLast * 4) % 16 == 0, "");
// (the images are written behind a barrier at the chain's very end: they may lie on anything of its)
static_assert(kCatHigh + (kGroup - 1) * kCatFloats <= kLdsFloatsAD && (kCatFloats * 4) % 16 == 0, "");
constexpr int kLdsFloatsE = kDDummy + 312;

// stages G-H (conv18, conv19, conv20, softmax, call) run for up to kTailBatch windows at a time,
// ONE WAVE PER WINDOW (dbh_forward.hip: batched tail), at the end of a group: the three layers'
// weights in fragment order (LDS-DMA'd behind the barrier of the batch's last conv17, when its
// concat image is dead: clear of its partial tiles at kRed, of slot 0 and of the staging), then two
// 18 x 50 activation buffers per wave (X: conv17 out, later conv19 out; Y: conv18 out).
constexpr int kTailBatch = 8;
constexpr int kTW18 = kRed + kRedFloats;                   // 6,144
constexpr int kTW19 = kStageEnd;                           // 15,904
constexpr int kTW20 = kTW19 + conv_weight_floats(18);      // 22,816
constexpr int kTWEnd = kTW20 + conv_weight_floats(19);     // 24,352
static_assert(kTW18 + conv_weight_floats(17) <= kStage && (kTW19 * 4) % 16 == 0, "");
constexpr int kTailBuf = 18 * kS48;                        // 900 floats
// The X/Y pairs of waves 0-2 lie where conv17's partial tiles were (dead behind the tail's first
// barrier), those of waves 3-7 from kSlot1 upwards (the concat buffer, dead too): between them
// slot 0 stays untouched - the NEXT window's first third of conv1d_2's weights is on its way there
// since the top of stage F (dbh_forward.hip: the fused stage A needs it right behind its first
// barrier).
constexpr int kTXLowWaves = 3;
constexpr int kTXLow = 0;                                  // wave w < 3: X at kTXLow + w * 2 * kTailBuf
constexpr int kTLog = kTXLow + kTXLowWaves * 2 * kTailBuf; // 32 logits per wave
constexpr int kTXHigh = kSlot1;                            // wave w >= 3: X at kTXHigh + (w - 3) * 2 * kTailBuf
static_assert(kTLog + kTailBatch * 32 <= kTW18, "the tail's low buffers would land on its weights");
static_assert(kTWEnd <= kSlot0, "the tail's weights would land on slot 0");
static_assert(kTXHigh + (kTailBatch - kTXLowWaves) * 2 * kTailBuf <= kLdsFloatsAD, "batched tail overflows the arena");
constexpr int tail_x_offset(int wave) {
    return wave < kTXLowWaves ? kTXLow + wave * 2 * kTailBuf
                              : kTXHigh + (wave - kTXLowWaves) * 2 * kTailBuf;
}
constexpr int kTailSlotFloats = 16 * 48;                   // conv17 output of one window
// per-workgroup scratch in global memory: [kTailBatch x conv17 output][kGroup x conv7 park]
constexpr int kWgTailOff = 0;
constexpr int kWgPark7Off = kTailBatch * kTailSlotFloats;
constexpr int kWgScratchFloats = kWgPark7Off + kGroup * kPark7Floats;
static_assert((kWgScratchFloats * 4) % 16 == 0 && (kWgPark7Off * 4) % 16 == 0, "");
constexpr int kArenaFloats =
    (kLdsFloatsE > kLdsFloatsAD ? kLdsFloatsE : kLdsFloatsAD) > kLdsFloatsD
        ? (kLdsFloatsE > kLdsFloatsAD ? kLdsFloatsE : kLdsFloatsAD)
        : kLdsFloatsD;
// Above the arena, for the whole kernel: a copy of the biases of conv1..conv16 and of BN1..BN4's
// scale/shift (two contiguous runs of the packed image), so that the epilogue parameters of
// stages A-E come from LDS (~100 cycles) instead of L2 (~700 cycles, exposed
// at the top of every layer); everything else is fetched from global memory well ahead of use.
// Then one word: the arrival counter of the split barrier (dbh_forward.hip: lds_arrive/lds_wait).
constexpr int kParams = kArenaFloats;
constexpr int kTabBias0 = bias_offset(0), kTabBias1 = bias_offset(16);
constexpr int kTabBn0 = bn_scale_offset(0), kTabBn1 = bn_scale_offset(4);
constexpr int kParamFloats = (kTabBias1 - kTabBias0) + (kTabBn1 - kTabBn0);     // 672 + 384
constexpr int kSync = kParams + kParamFloats;
// 20 words: [0..2] arrivals at the end of tile t of a stage-B layer (dbh_forward.hip: chain_arrive),
// [4..19] per wave the posts of its two neighbours (halo_post); the same for stage D's chain: six
// tile words (one per tile of its two layers), then the posts
constexpr int kSyncTiles = kSync, kSyncHalo = kSync + 4;
constexpr int kSyncDTiles = kSync + 20, kSyncDHalo = kSyncDTiles + 6, kSyncWords = 20 + 6 + 16;
// window statistics: 16 int64 partial sums (two per wave) and the resulting {mean, 1/std} doubles
constexpr int kStatRed = kSync + kSyncWords;
constexpr int kStatOut = kStatRed + kGroup * 32;      // (a slot of 16 int64 per window staged at once)
static_assert(kStatRed % 2 == 0, "64-bit words");
// one counter word per half of each of conv7's four wave pairs (dbh_forward.hip: pair_signal)
constexpr int kPairSync = kStatOut + 4;
// which window this workgroup takes next (windows handed out by a counter: dbh_forward.hip) and
// which windows wait in the slots of the batched tail
constexpr int kNextWin = kPairSync + 8;
constexpr int kTailWins = kNextWin + 1;
// phase stamps of the clock probe (dbh_forward.hip: phase_stamp): kPhaseMarks per group for a
// workgroup's first kPhaseGroups groups, dumped behind the launch's work
// (marks 5..8: stage F's inner intervals as wave 0 sees them, summed over the group's windows: to the
// end of conv17's MFMAs, to behind its barrier, to the end of the reduction, to behind the window's
// last barrier)
// (marks 9..11: stages A-C's inner intervals, summed over the group's windows: a window's top to the
// start of stage B's chain, the chain, conv1d_7)
constexpr int kPhaseMarks = 14, kPhaseGroups = 12;      // (12, 13: the group's last conv1d_7; from it to the chain)
constexpr int kPhase = kTailWins + kTailBatch;
constexpr int kLdsFloats = kPhase + kPhaseMarks * kPhaseGroups;
static_assert(kLdsFloats * 4 <= 160 * 1024, "LDS arena exceeds 160 KiB");

// floats per window of the debug dump after each stage (dense [L][C])
constexpr int kStageFloats[8] = {512 * 48, 256 * 48, 128 * 48, 64 * 48, 32 * 192, 16 * 48,
                                 8 * 48, 32};

}  // namespace dbh
