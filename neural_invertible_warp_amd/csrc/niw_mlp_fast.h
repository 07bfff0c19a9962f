// Geometry of the split-bf16 ("fast precision") image of the field MLP, shared by niw_mlp_fast.hip and its callers.
//
// Precision modes of the field-MLP entry points (include/niw.h NIW_PREC_*):
//   NIW_PREC_FP32    v_mfma_f32_32x32x2_f32, exact fp32 (the default and the only mode the headline numbers use)
//   NIW_PREC_BF16X3  every fp32 operand value x is carried as TWO bf16 planes, hi = bf16(x), mid = bf16(x - hi) (16 significand bits
//                    together), and a product a*b is formed as a_hi*b_hi + a_hi*b_mid + a_mid*b_hi on v_mfma_f32_32x32x16_bf16 with
//                    fp32 accumulation: three matrix instructions of 32 cycles per 16-deep k-slice instead of eight of 64 (96 vs 512
//                    matrix-pipe cycles).  Dropped terms are below 2^-16 of a product.
//   NIW_PREC_BF16    the hi planes only: one instruction per 16-deep k-slice (SURVEY 8(c)'s bf16 tolerance class).
// Why not the 6- / 9-term (three-plane) formulation that reaches fp32-level error: the register-chained design keeps a layer's input AND
// output for 32 samples in registers; three planes of both are 384 VGPRs before accumulators, weight fragments and encodings (512 per lane).
//
// Image layout.  A "chunk" is the A operand of one (32-row block, 16-deep k-step): [plane 0..1][lane 0..63][8 bf16] = 2 KiB;
// lane (i = lane & 31, h = lane >> 5), element j holds W[row i of the block][k = 16 q + perm(h, j)], perm(h, j) = 8 (j >> 2) + 4 h + (j & 3):
// the k order in which the 32x32 fp32 accumulator of the PREVIOUS layer, converted pairwise to bf16, is the B operand of this one
// (accumulator register 8 s + j of lane half h <-> row 16 s + perm(h, j); cdna_hip_programming.md section 3).
// The chunks of a kernel lie in the ORDER THE KERNEL CONSUMES THEM -- one linear stream per kernel -- so that the four waves of a
// workgroup, which all walk the same stream, can share every chunk through LDS (niw_mlp_fast.hip WeightStream):
//   forward  stream: layers 0..9, row block nb, k-step q          -> W_l[out_row(l, 32 nb + i)][fwd_slot_col(l, 16 q + perm)]
//   backward stream: the dX chain's order (segments below), slot block ob, reduction step q
//                                                                -> W_l[out_row(l, 16 q + perm)][fwd_slot_col(l, 32 ob + i)]
//   bias     section: fp32 [layer][row block][lane half][16], the accumulator a forward block starts from.
// Every layer (segment) starts on a multiple of kStageChunks chunks (the backward stream pads layer 9's four chunks to eight).
// Both heads are ordinary row blocks here (the fp32 kernel forms them on the vector ALU to save 2.3 % of its matrix work; at a
// sixteenth of the cost per instruction they are not worth a special case): layer 7 has a ninth row block whose row 0 is the density
// row, layer 9 is one row block with three valid rows.
#pragma once
#include "niw_common.h"

namespace niw {

constexpr int kChunkBytes = 2048;
constexpr int kStageChunks = 8;           // chunks the four waves of a workgroup fetch together (one LDS stage = 16 KiB)

__host__ __device__ constexpr int fast_nb(int l) { return l == 7 ? 9 : l == 8 ? 4 : l == 9 ? 1 : 8; }          // forward row blocks
__host__ __device__ constexpr int fast_ks(int l) { return l == 0 ? 4 : l == 4 ? 20 : l == 8 ? 18 : l == 9 ? 8 : 16; }   // forward k-steps of 16
__host__ __device__ constexpr int fast_rs(int l) { return l == 7 ? 17 : l == 8 ? 8 : l == 9 ? 1 : 16; }          // backward reduction steps

// ---- forward stream: chunk index of (layer l, row block 0, k-step 0)
__host__ __device__ constexpr int fast_fwd_chunk(int l) {
    int o = 0;
    for (int i = 0; i < l; ++i) o += fast_nb(i) * fast_ks(i);
    return o;
}
constexpr int kFastFwdChunks = fast_fwd_chunk(kLayers);

// ---- backward stream: segments in the order the dX chain runs them
struct FastBwdSeg { int layer, ob0, nob, pad; };      // slot blocks ob0 .. ob0 + nob - 1 of `layer`, then `pad` zero chunks
constexpr int kFastBwdSegs = 12;
__host__ __device__ constexpr FastBwdSeg fast_bwd_seg(int s) {
    constexpr FastBwdSeg t[kFastBwdSegs] = {{9, 0, 4, 4}, {8, 8, 1, 0}, {8, 0, 8, 0}, {7, 0, 8, 0}, {6, 0, 8, 0}, {5, 0, 8, 0},
                                            {4, 8, 2, 0}, {4, 0, 8, 0}, {3, 0, 8, 0}, {2, 0, 8, 0}, {1, 0, 8, 0}, {0, 0, 2, 0}};
    return t[s];
}
__host__ __device__ constexpr int fast_bwd_chunk(int s) {      // first chunk of segment s (counted from the start of the backward stream)
    int o = 0;
    for (int i = 0; i < s; ++i) o += fast_bwd_seg(i).nob * fast_rs(fast_bwd_seg(i).layer) + fast_bwd_seg(i).pad;
    return o;
}
constexpr int kFastBwdChunks = fast_bwd_chunk(kFastBwdSegs);

constexpr bool fast_streams_are_stage_aligned() {
    for (int l = 0; l <= kLayers; ++l)
        if (fast_fwd_chunk(l) % kStageChunks) return false;
    for (int s = 0; s <= kFastBwdSegs; ++s)
        if (fast_bwd_chunk(s) % kStageChunks) return false;
    return true;
}
static_assert(fast_streams_are_stage_aligned(), "every layer / segment must start on a stage boundary");

constexpr int kFastBwdOffBytes = kFastFwdChunks * kChunkBytes;                     // the backward stream follows the forward one
__host__ __device__ constexpr int fast_bias_off(int l) {                           // bytes; [row block][half][16] floats per layer
    int o = kFastBwdOffBytes + kFastBwdChunks * kChunkBytes;
    for (int i = 0; i < l; ++i) o += fast_nb(i) * 128;
    return o;
}
constexpr int kFastImageBytes = fast_bias_off(kLayers);
static_assert(kFastImageBytes % 16 == 0, "image is a whole number of 16-byte words");

__host__ __device__ constexpr int fast_perm(int h, int j) { return 8 * (j >> 2) + 4 * h + (j & 3); }

}  // namespace niw
