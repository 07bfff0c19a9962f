// Field MLP backward, pass 2: weight and bias gradients.
//     dW[n][k] = sum_m dY[n][m] * X[k][m]
// dY and X are the feature-major tensors written by pass 1 / the forward ([rows][Mpad], sample
// index contiguous), i.e. both operands are "row = output index, contiguous = reduction
// index": an NT GEMM with a tiny output (<= 256 x 256) and a huge reduction (all samples).
// Decomposition: split-M.  Every workgroup owns one contiguous range of samples, stages
// 32-sample slices of both operands through LDS (double-buffered, rows padded by 4 floats so
// that ds_read_b128 fragment reads are conflict free), accumulates the whole output tile in
// registers with v_mfma_f32_32x32x2_f32, and writes ONE partial tile at the end.  A second
// kernel sums the partial tiles in a fixed order (deterministic; no float atomics) and
// scatters the result into the state-dict layout of d_params.
#include "niw_common.h"
#include <mutex>
#include "niw_mlp_device.h"
#include "niw_bf16.h"
#include <type_traits>
#include <stdlib.h>
#include <atomic>
#include "niw_trace.h"

using namespace niw;
NIW_TRACE_SETTER(niw_trace_set_dw)

int niw_launch_mlp_bwd_dx(const float* packed, const float* center, const float* ray, const float* depth,
                          int64_t n_rays, int n_samples, int density_activ, const float* rgb, const float* d_rgb,
                          const float* d_sigma, const float* save, float* gradws, float* d_center, float* d_ray,
                          hipStream_t stream);

int niw_launch_mlp_bwd_dx_fast(int precision, const void* image, const float* center, const float* ray, const float* depth, int64_t n_rays,
                               int n_samples, int density_activ, const float* rgb, const float* d_rgb, const float* d_sigma, const float* save,
                               float* gradws, float* d_center, float* d_ray, hipStream_t stream);

// C[n][k] = sum_m A[n][m] B[k][m] per batch; tile 256x256 (wide) or 256x64; partial tiles
// [batch][nsplit][TN*TK + 256] (the trailing 256 floats are row sums of A (bias_side 1) or B (2)).
// Operand addressing (NiwGemmOperand, niw_common.h): 32-sample slice `s` of row r sits at p + batch*batch_stride + r*row_stride + s*32.
int niw_launch_nt_gemm(int wide, NiwGemmOperand A, NiwGemmOperand B, long long mpad, int batches, float* partial,
                       int bias_side, int* nsplit_out, hipStream_t st);

namespace {

constexpr int kLdsStride = 36;   // 32 samples + 4 pad floats per row
#ifndef NIW_DW_LOAD_AUX
// cache policy of the operand loads of the exact-fp32 NT GEMM: 2 = non-temporal.  Every slice of `save` / `gradws` is read once per product,
// 512 contiguous bytes per half wave (whole lines), and the skinny launch is HBM-bound: dW group 6070 -> 6047 us at 785 k samples, 2106 ->
// 2088 at 260 k, level at 32 k (round 5; 0 = default policy)
#define NIW_DW_LOAD_AUX 2
#endif
#ifndef NIW_DW_SUMS_KB
#define NIW_DW_SUMS_KB 0
#endif
#ifndef NIW_DW_PF_SKINNY
#define NIW_DW_PF_SKINNY 2
#endif
#ifndef NIW_DW_PF_COLOUR
#define NIW_DW_PF_COLOUR 2
#endif
#ifndef NIW_DW_COLOUR_288
constexpr int kColourTK = 320;   // colour layer as 4 x 2 waves of 1 x 5 blocks (the 10th column block holds no valid row and is skipped)
#else
constexpr int kColourTK = 288;
#endif

// Up to kMaxBatch independent products per launch (blockIdx.y): the seven 256 x 256 weight gradients of a network, or its four
// skinny pieces -- each with its own operand pair and bias side.
constexpr int kMaxBatch = 28;      // 7 layers x 4 quadrant tiles in the small-batch form of the wide launch (niw_mlp_bwd_dw)
struct GemmBatch {
    NiwGemmOperand A[kMaxBatch], B[kMaxBatch];
    int bias_side[kMaxBatch];
};

// WN x WK waves; each wave owns NBW x KBW blocks of 32x32 outputs.
// Operand slices reach LDS through registers: per 32-sample step every thread issues LOADS 16-byte buffer loads -- ONE 32-bit
// lane offset per operand, the row group and the step folded into the scalar offset, rows beyond the operand's valid rows cut
// off by the descriptor's range (they read as zero) -- and writes them to the other LDS buffer half a step later.  (Round 2:
// with a 64-bit address and an exec-mask branch per load the loads alone cost the wide kernel 11 %: both waves of a SIMD reach
// them together, right after the barrier, so their ~100 address / branch instructions ran with the matrix pipe idle.)
// SKIP: 32 x 32 output blocks that lie entirely beyond the valid rows of either operand (the 3-row colour head in a 64-column
// tile, the 128-row operand in a 256-row tile, the 288 of 320 columns of the colour layer) issue no MFMAs -- the skinny
// launches are bound by their padded matrix work, not by HBM.
// QUAD: the operands are quad-row images [row / 4][samples][4] (the field MLP's workspaces, niw_mlp_device.h): a 16-byte load is
// four rows of one sample and goes to four LDS rows as ds_write_b32 (32 lanes = 32 consecutive samples of a row: conflict free).
// Plain [row][samples] operands (the warp's factor rows) take the b128 path.
// Block -> (product, sample range).  Plain launches: grid (nsplit, products).  XCD-grouped launches (map.groups_pad > 0; the small-batch
// wide launch, whose products come in sets of `map.set` tiles that read the same operand rows): a 1-D grid of set x groups_pad blocks,
// block = member * groups_pad + group, group = set_index * nsplit + split.  groups_pad is a multiple of 8 and blocks are dealt round-robin
// over the 8 XCDs, so the members of a set -- same group, ids groups_pad apart -- share an XCD's L2 (placement is speed only).
struct BlockMap {
    int groups_pad, groups, nsplit, set;
};

template <int WN, int WK, int NBW, int KBW, bool SKIP, int PF, bool QUAD>
__global__ __launch_bounds__(64 * WN * WK) void dw_gemm_kernel(GemmBatch batch, int steps_total, int steps_per_wg,
                                                               float* __restrict__ partial, BlockMap map) {
    constexpr int kTraceKind = WN * 1000 + WK * 100 + NBW * 10 + KBW;       // 4224: wide tile, 8112: skinny, 4215: colour, 4212: quadrant tile
    NIW_STAMP_KIND(0, kTraceKind);
    int bx = blockIdx.x, by = blockIdx.y, nx = gridDim.x;
    if (map.groups_pad > 0) {
        const int member = (int)blockIdx.x / map.groups_pad, group = (int)blockIdx.x % map.groups_pad;
        if (group >= map.groups) return;                      // (the whole workgroup leaves before its first barrier)
        bx = group % map.nsplit;
        by = (group / map.nsplit) * map.set + member;
        nx = map.nsplit;
    }
    const NiwGemmOperand opA = batch.A[by], opB = batch.B[by];
    const int bias_side = batch.bias_side[by];
    constexpr int TN = WN * NBW * 32, TK = WK * KBW * 32, NT = 64 * WN * WK;
    constexpr int ROWS = TN + TK;
    constexpr int LOADS = ROWS * 8 / NT;                  // float4 loads per thread per 32-sample step
    constexpr int GROUP = NT / 8;                         // rows covered by one load of the whole workgroup
    static_assert(ROWS * 8 % NT == 0 && TN % GROUP == 0, "tile must divide over the threads, each load entirely A or B");
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wn = wave / WK, wk = wave % WK;
    const int i = lane & 31, h = lane >> 5;
    const int step0 = bx * steps_per_wg;
    bool live[NBW][KBW];
#pragma unroll
    for (int x = 0; x < NBW; ++x)
#pragma unroll
        for (int y = 0; y < KBW; ++y) live[x][y] = !SKIP || ((wn * NBW + x) * 32 < opA.rows && (wk * KBW + y) * 32 < opB.rows);
    const int nsteps = min(steps_per_wg, steps_total - step0);
    // descriptors end after the last valid row: loads of rows beyond return zero (host: rows * row_stride * 4 < 2^31)
    const int strideA4 = (int)opA.row_stride * 4, strideB4 = (int)opB.row_stride * 4;
    const int cutA = QUAD ? (min(opA.rows, TN) + 3) / 4 * 4 : min(opA.rows, TN), cutB = QUAD ? (min(opB.rows, TK) + 3) / 4 * 4 : min(opB.rows, TK);
    const rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(opA.p), 0, cutA * strideA4, 0x00020000);
    const rsrc_t rsB = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(opB.p), 0, cutB * strideB4, 0x00020000);
    // plain: thread -> (row tid / 8 of the group, 16-byte column tid % 8);  quad: thread -> (quad tid / 32 of the group, sample tid % 32)
    const int voffA = QUAD ? (tid >> 5) * 4 * strideA4 + (tid & 31) * 16 : (tid >> 3) * strideA4 + (tid & 7) * 16;
    const int voffB = QUAD ? (tid >> 5) * 4 * strideB4 + (tid & 31) * 16 : (tid >> 3) * strideB4 + (tid & 7) * 16;
    constexpr int STEP_BYTES = QUAD ? 512 : 128;          // one 32-sample step along a row (plain) / along a quad (quad-row image)

    f32x16 acc[NBW][KBW];
#pragma unroll
    for (int x = 0; x < NBW; ++x)
#pragma unroll
        for (int y = 0; y < KBW; ++y)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[x][y][r] = 0.f;
    float bsum = 0.f;

    // Register prefetch: the loads of step s + PF are issued at the start of step s and reach LDS half a step before they are
    // needed (LDS stays double-buffered).  PF = 1 for the wide tile (a step is ~7 us of MFMA work: ample cover for an HBM read,
    // and its 8 loads per thread leave no registers for a second stage); PF = 2 for the skinny and colour tiles, whose steps
    // are 4-8x shorter than an HBM round trip under load -- with PF = 1 they were bound by that latency, not by MFMA or bytes.
    f32x4 stage[PF][LOADS];
    auto gload = [&](int step, int slot) {
#pragma unroll
        for (int k = 0; k < LOADS; ++k) {
            const bool isA = k * GROUP < TN;
            const int g = isA ? k * GROUP : k * GROUP - TN;                 // first row of this load's row group
            stage[slot][k] = isA ? buf_load4_aux<NIW_DW_LOAD_AUX>(rsA, voffA, step * STEP_BYTES + g * strideA4)
                                 : buf_load4_aux<NIW_DW_LOAD_AUX>(rsB, voffB, step * STEP_BYTES + g * strideB4);
        }
    };
    auto lstore = [&](int buf, int slot) {
#pragma unroll
        for (int k = 0; k < LOADS; ++k) {
            if (QUAD) {
                const int row0 = k * GROUP + (tid >> 5) * 4, m = tid & 31;
#pragma unroll
                for (int j = 0; j < 4; ++j) lds[(buf * ROWS + row0 + j) * kLdsStride + m] = stage[slot][k][j];
            } else {
                const int idx = tid + k * NT, row = idx >> 3, c4 = idx & 7;
                *reinterpret_cast<f32x4*>(lds + (buf * ROWS + row) * kLdsStride + c4 * 4) = stage[slot][k];
            }
        }
    };
    // one 32-sample step; `slot` = s % PF at compile time (the register stage that receives step s + PF, after step s + 1 has
    // left stage (s + 1) % PF for LDS)
    auto step_body = [&](int s, auto slot_c) {
        constexpr int slot = decltype(slot_c)::value;
        const int buf = s & 1;
        if (s + PF < nsteps) gload(step0 + s + PF, slot);
        const float* As = lds + buf * ROWS * kLdsStride;
        const float* Bs = As + TN * kLdsStride;
        // fragment reads of k-block kb+1 are issued before the MFMAs of k-block kb (the two waves of a SIMD are
        // barrier-synchronised, so without this both sit in the LDS latency at the same time)
        f32x4 af[2][NBW], bf[2][KBW];
        auto lread = [&](int kb, int fs) {
#pragma unroll
            for (int x = 0; x < NBW; ++x)
                af[fs][x] = *reinterpret_cast<const f32x4*>(As + ((wn * NBW + x) * 32 + i) * kLdsStride + kb * 8 + 4 * h);
#pragma unroll
            for (int y = 0; y < KBW; ++y)
                bf[fs][y] = *reinterpret_cast<const f32x4*>(Bs + ((wk * KBW + y) * 32 + i) * kLdsStride + kb * 8 + 4 * h);
        };
        auto row_sums = [&]() {
            if (bias_side) {
                // row sums of the dY operand: thread t < rows sums its row of the staged slice
                const int nrows = bias_side == 1 ? TN : TK;
                if (tid < nrows) {
                    // two packed adds per 16-byte read (the halves of a ds_read_b128 are aligned register pairs: v_pk_add_f32 without
                    // moves) and two scalar adds per step: 18 vector instructions where the scalar form compiled to ~45 -- and vector
                    // instructions are paid in matrix-pipe time (tools/mfma_valu_contention.hip)
                    typedef float f32x2 __attribute__((ext_vector_type(2)));
                    const float* rowp = (bias_side == 1 ? As : Bs) + tid * kLdsStride;
                    f32x2 s2 = {0.f, 0.f};
#pragma unroll
                    for (int c = 0; c < 8; ++c) {
                        const f32x4 v = *reinterpret_cast<const f32x4*>(rowp + c * 4);
                        s2 += v.lo;
                        s2 += v.hi;
                    }
                    bsum += s2[0] + s2[1];
                }
            }
        };
        lread(0, 0);
#pragma unroll
        for (int kb = 0; kb < 4; ++kb) {
            if (kb + 1 < 4) lread(kb + 1, (kb + 1) & 1);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int x = 0; x < NBW; ++x)
#pragma unroll
                for (int y = 0; y < KBW; ++y)
                    if (live[x][y]) {
#pragma unroll
                        for (int t = 0; t < 4; ++t) acc[x][y] = mfma32(af[kb & 1][x][t], bf[kb & 1][y][t], acc[x][y]);
                    }
            __builtin_amdgcn_sched_barrier(0);
            // the next slice goes to the OTHER LDS buffer (free since the barrier that ended the previous step): write it
            // under the MFMAs of the last k-block instead of between the last MFMA and the barrier, where all eight waves
            // would queue 64 KB of ds_write with the matrix pipe idle
            if (kb == 2 && s + 1 < nsteps) lstore(buf ^ 1, (slot + 1) % PF);
            // bias sums of waves 0..3 go between two k-blocks, where their SIMD partners (waves 4..7) keep the matrix pipe busy
            // (measured equal to placing them after the last k-block)
            if (kb == NIW_DW_SUMS_KB) row_sums();
        }
        if (NIW_DW_SUMS_KB > 3) row_sums();
        __syncthreads();
    };

    if (nsteps > 0) {
        gload(step0, 0);
        lstore(0, 0);
    }
    if (PF >= 2 && nsteps > 1) gload(step0 + 1, 1 % PF);
    if (PF >= 3 && nsteps > 2) gload(step0 + 2, 2 % PF);
    __syncthreads();
    NIW_STAMP_KIND(1, kTraceKind);                   // first slice in LDS: the first matrix instruction follows
    for (int s = 0; s < nsteps; s += PF) {
        step_body(s, std::integral_constant<int, 0>{});
        if (PF >= 2 && s + 1 < nsteps) step_body(s + 1, std::integral_constant<int, 1 % PF>{});
        if (PF >= 3 && s + 2 < nsteps) step_body(s + 2, std::integral_constant<int, 2 % PF>{});
    }
    NIW_STAMP_KIND(2, kTraceKind);                   // reduction done: the partial tile follows
    // partial tile [TN][TK] (+ TN-or-TK bias sums) of this workgroup
    float* out = partial + ((long long)by * nx + bx) * (TN * TK + 256);
#pragma unroll
    for (int x = 0; x < NBW; ++x)
#pragma unroll
        for (int y = 0; y < KBW; ++y)
#pragma unroll
            for (int r = 0; r < 16; ++r)
                out[((wn * NBW + x) * 32 + acc_row(r, h)) * TK + (wk * KBW + y) * 32 + i] = acc[x][y][r];
    if (tid < 256) out[TN * TK + tid] = bsum;
    NIW_STAMP_KIND_LAST(3, kTraceKind);
}

// The same product in the fast-precision modes (include/niw.h NIW_PREC_BF16X3 / NIW_PREC_BF16): the fp32 operands (quad-row images
// only) are split into bf16 planes (hi, mid) on their way into LDS and multiplied on v_mfma_f32_32x32x16_bf16 with fp32 accumulation --
// TERMS = 3: hi*hi + hi*mid + mid*hi, TERMS = 1: hi*hi.  Same split-M decomposition, same partial-tile format, same reduction kernel.
//   * a 16-byte load is four rows of ONE sample; a bf16 dword needs TWO samples of one row, so neighbouring lanes (samples m, m ^ 1)
//     swap two values each: the even lane then owns rows 0, 1 of the quad for both samples, the odd lane rows 2, 3;
//   * LDS image of a 32-sample slice: [plane][row][64 B]; the four 16-byte chunks of a row (8 samples each = one MFMA operand
//     fragment of a lane) are XOR-swizzled with (row >> 2) & 3, which makes the ds_read_b128 of 16 consecutive rows conflict free
//     without padding (2 planes x 512 rows x 64 B x 2 buffers = 128 KiB for the wide tile);
//   * bias sums are accumulated in fp32 from the staged registers (exact, like the fp32 path), not from the planes;
//   * with a sixteenth / three sixteenths of the matrix time the kernel is bound by its HBM reads (the fp32 workspaces).
template <int WN, int WK, int NBW, int KBW, bool SKIP, int PF, int TERMS>
__global__ __launch_bounds__(64 * WN * WK) void dw_gemm_fast_kernel(GemmBatch batch, int steps_total, int steps_per_wg,
                                                                    float* __restrict__ partial) {
    const NiwGemmOperand opA = batch.A[blockIdx.y], opB = batch.B[blockIdx.y];
    const int bias_side = batch.bias_side[blockIdx.y];
    constexpr int TN = WN * NBW * 32, TK = WK * KBW * 32, NT = 64 * WN * WK;
    constexpr int ROWS = TN + TK;
    constexpr int LOADS = ROWS * 8 / NT, GROUP = NT / 8;
    constexpr int PL = TERMS >= 3 ? 2 : 1;                  // planes kept in LDS
    constexpr int PLANE_BYTES = ROWS * 64, BUF_BYTES = PL * PLANE_BYTES;
    static_assert(ROWS * 8 % NT == 0 && TN % GROUP == 0, "tile must divide over the threads, each load entirely A or B");
    extern __shared__ __attribute__((aligned(16))) float lds[];
    char* const ldsb = reinterpret_cast<char*>(lds);
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wn = wave / WK, wk = wave % WK;
    const int i = lane & 31, h = lane >> 5;
    const int step0 = blockIdx.x * steps_per_wg;
    bool live[NBW][KBW];
#pragma unroll
    for (int x = 0; x < NBW; ++x)
#pragma unroll
        for (int y = 0; y < KBW; ++y) live[x][y] = !SKIP || ((wn * NBW + x) * 32 < opA.rows && (wk * KBW + y) * 32 < opB.rows);
    const int nsteps = min(steps_per_wg, steps_total - step0);
    const int strideA4 = (int)opA.row_stride * 4, strideB4 = (int)opB.row_stride * 4;
    const int cutA = (min(opA.rows, TN) + 3) / 4 * 4, cutB = (min(opB.rows, TK) + 3) / 4 * 4;
    const rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(opA.p), 0, cutA * strideA4, 0x00020000);
    const rsrc_t rsB = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(opB.p), 0, cutB * strideB4, 0x00020000);
    const int voffA = (tid >> 5) * 4 * strideA4 + (tid & 31) * 16, voffB = (tid >> 5) * 4 * strideB4 + (tid & 31) * 16;
    constexpr int STEP_BYTES = 512;

    f32x16 acc[NBW][KBW];
#pragma unroll
    for (int x = 0; x < NBW; ++x)
#pragma unroll
        for (int y = 0; y < KBW; ++y)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[x][y][r] = 0.f;
    constexpr int SIDE_LOADS = LOADS / 2 > 0 ? LOADS : LOADS;   // (bias sums are kept per load slot; only one operand side is ever used)
    float bsum[SIDE_LOADS][4];
#pragma unroll
    for (int k = 0; k < SIDE_LOADS; ++k)
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) bsum[k][jj] = 0.f;

    f32x4 stage[PF][LOADS];
    auto gload = [&](int step, int slot) {
#pragma unroll
        for (int k = 0; k < LOADS; ++k) {
            const bool isA = k * GROUP < TN;
            const int g = isA ? k * GROUP : k * GROUP - TN;
            stage[slot][k] = isA ? buf_load4(rsA, voffA, step * STEP_BYTES + g * strideA4) : buf_load4(rsB, voffB, step * STEP_BYTES + g * strideB4);
        }
    };
    // byte offset of dword w (two samples) of `row` inside a plane
    auto lds_off = [&](int row, int w) { return row * 64 + ((((w >> 2) ^ (row >> 2)) & 3) << 4) + ((w & 3) << 2); };
    auto lstore = [&](int buf, int slot) {
        const int m = tid & 31, odd = m & 1, w = m >> 1;
#pragma unroll
        for (int k = 0; k < LOADS; ++k) {
            const f32x4 v = stage[slot][k];
            const bool isA = k * GROUP < TN;
            if ((bias_side == 1 && isA) || (bias_side == 2 && !isA)) {
#pragma unroll
                for (int jj = 0; jj < 4; ++jj) bsum[k][jj] += v[jj];
            }
            // the even lane keeps rows 0, 1 and receives them for sample m + 1; the odd lane keeps rows 2, 3 and receives them for m - 1
            const float s0 = odd ? v[0] : v[2], s1 = odd ? v[1] : v[3];
            const float r0 = __shfl_xor(s0, 1), r1 = __shfl_xor(s1, 1);
            const float a0 = odd ? r0 : v[0], b0 = odd ? v[2] : r0;      // row (0 | 2): samples (m & ~1, m | 1)
            const float a1 = odd ? r1 : v[1], b1 = odd ? v[3] : r1;      // row (1 | 3)
            unsigned hi0, mid0, hi1, mid1;
            split_pair(a0, b0, hi0, mid0);
            split_pair(a1, b1, hi1, mid1);
            const int row = k * GROUP + (tid >> 5) * 4 + 2 * odd;
            char* base = ldsb + buf * BUF_BYTES;
            *reinterpret_cast<unsigned*>(base + lds_off(row, w)) = hi0;
            *reinterpret_cast<unsigned*>(base + lds_off(row + 1, w)) = hi1;
            if (PL == 2) {
                *reinterpret_cast<unsigned*>(base + PLANE_BYTES + lds_off(row, w)) = mid0;
                *reinterpret_cast<unsigned*>(base + PLANE_BYTES + lds_off(row + 1, w)) = mid1;
            }
        }
    };
    auto step_body = [&](int s, auto slot_c) {
        constexpr int slot = decltype(slot_c)::value;
        const int buf = s & 1;
        if (s + PF < nsteps) gload(step0 + s + PF, slot);
        const char* base = ldsb + buf * BUF_BYTES;
        // fragment of a lane: 8 consecutive samples (16 bytes) of its row: chunk 2 ks + h of the row
        auto frag = [&](int plane, int row, int ks) {
            return *reinterpret_cast<const u32x4_t*>(base + plane * PLANE_BYTES + row * 64 + ((((2 * ks + h) ^ (row >> 2)) & 3) << 4));
        };
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            u32x4_t af[PL][NBW];
#pragma unroll
            for (int pl = 0; pl < PL; ++pl)
#pragma unroll
                for (int x = 0; x < NBW; ++x) af[pl][x] = frag(pl, (wn * NBW + x) * 32 + i, ks);
#pragma unroll
            for (int y = 0; y < KBW; ++y) {
                u32x4_t bf[PL];
#pragma unroll
                for (int pl = 0; pl < PL; ++pl) bf[pl] = frag(pl, TN + (wk * KBW + y) * 32 + i, ks);
#pragma unroll
                for (int x = 0; x < NBW; ++x)
                    if (live[x][y]) {
                        if (TERMS >= 3) {
                            acc[x][y] = mfma_bf16(af[0][x], bf[PL - 1], acc[x][y]);
                            acc[x][y] = mfma_bf16(af[PL - 1][x], bf[0], acc[x][y]);
                        }
                        acc[x][y] = mfma_bf16(af[0][x], bf[0], acc[x][y]);
                    }
            }
            if (ks == 0 && s + 1 < nsteps) lstore(buf ^ 1, (slot + 1) % PF);
        }
        __syncthreads();
    };

    if (nsteps > 0) {
        gload(step0, 0);
        lstore(0, 0);
    }
    if (PF >= 2 && nsteps > 1) gload(step0 + 1, 1 % PF);
    if (PF >= 3 && nsteps > 2) gload(step0 + 2, 2 % PF);
    __syncthreads();
    for (int s = 0; s < nsteps; s += PF) {
        step_body(s, std::integral_constant<int, 0>{});
        if (PF >= 2 && s + 1 < nsteps) step_body(s + 1, std::integral_constant<int, 1 % PF>{});
        if (PF >= 3 && s + 2 < nsteps) step_body(s + 2, std::integral_constant<int, 2 % PF>{});
    }
    float* out = partial + ((long long)blockIdx.y * gridDim.x + blockIdx.x) * (TN * TK + 256);
#pragma unroll
    for (int x = 0; x < NBW; ++x)
#pragma unroll
        for (int y = 0; y < KBW; ++y)
#pragma unroll
            for (int r = 0; r < 16; ++r)
                out[((wn * NBW + x) * 32 + acc_row(r, h)) * TK + (wk * KBW + y) * 32 + i] = acc[x][y][r];
    // bias sums: reduce the 32 samples of a wave half (fixed xor tree), one write per row
    if (tid < 256) out[TN * TK + tid] = 0.f;
    __syncthreads();
    if (bias_side) {
#pragma unroll
        for (int k = 0; k < LOADS; ++k) {
            const bool isA = k * GROUP < TN;
            if ((bias_side == 1 && isA) || (bias_side == 2 && !isA)) {
#pragma unroll
                for (int jj = 0; jj < 4; ++jj) {
                    float v = bsum[k][jj];
#pragma unroll
                    for (int o = 16; o > 0; o >>= 1) v += __shfl_xor(v, o);
                    const int row = (isA ? k * GROUP : k * GROUP - TN) + (tid >> 5) * 4 + jj;
                    if ((tid & 31) == 0 && row < 256) out[TN * TK + row] = v;
                }
            }
        }
    }
}

// The one-term (bf16) product on bf16 WORKSPACES (niw_mlp_fast.hip kHalfWorkspace): the operands are quad-row images of bf16 -- four rows
// of one sample in 8 bytes, quad q of sample m at (q * Mpad + m) * 8 -- i.e. already the plane this mode multiplies.  A 16-byte load
// is one quad of TWO neighbouring samples: transposed in registers (four v_perm_b32) it is the four LDS dwords (row, sample pair) of
// the same [row][64 B] swizzled image the fp32-workspace loader builds; no split, no lane exchange, half the bytes from HBM -- which is
// what this kernel is bound by.  A workgroup load covers NT / 16 quads (128 rows for 512 threads); operands whose tile is not a
// multiple of that (64 or 320 rows) leave part of a load idle.  Bias sums: fp32 sums of the bf16 values (the fp32-workspace form sums
// the unrounded values).
template <int WN, int WK, int NBW, int KBW, bool SKIP, int PF>
__global__ __launch_bounds__(64 * WN * WK) void dw_gemm_half_kernel(GemmBatch batch, int steps_total, int steps_per_wg,
                                                                    float* __restrict__ partial) {
    const NiwGemmOperand opA = batch.A[blockIdx.y], opB = batch.B[blockIdx.y];
    const int bias_side = batch.bias_side[blockIdx.y];
    constexpr int TN = WN * NBW * 32, TK = WK * KBW * 32, NT = 64 * WN * WK;
    constexpr int ROWS = TN + TK;
    constexpr int QPL = NT / 16;                            // quads per workgroup load
    constexpr int LA = (TN / 4 + QPL - 1) / QPL, LB = (TK / 4 + QPL - 1) / QPL, LOADS = LA + LB;
    constexpr int BUF_BYTES = ROWS * 64;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    char* const ldsb = reinterpret_cast<char*>(lds);
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wn = wave / WK, wk = wave % WK;
    const int i = lane & 31, h = lane >> 5;
    const int step0 = blockIdx.x * steps_per_wg;
    bool live[NBW][KBW];
#pragma unroll
    for (int x = 0; x < NBW; ++x)
#pragma unroll
        for (int y = 0; y < KBW; ++y) live[x][y] = !SKIP || ((wn * NBW + x) * 32 < opA.rows && (wk * KBW + y) * 32 < opB.rows);
    const int nsteps = min(steps_per_wg, steps_total - step0);
    // bytes between consecutive quads of an operand: row_stride samples of 8 bytes; descriptors end after the last valid quad
    const int qsA = (int)opA.row_stride * 8, qsB = (int)opB.row_stride * 8;
    const int cutA = (min(opA.rows, TN) + 3) / 4, cutB = (min(opB.rows, TK) + 3) / 4;
    const rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(opA.p), 0, cutA * qsA, 0x00020000);
    const rsrc_t rsB = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(opB.p), 0, cutB * qsB, 0x00020000);
    const int quad = tid >> 4, w = tid & 15;                 // this thread's quad within a load, its pair of samples within the step
    const int voffA = quad * qsA + w * 16, voffB = quad * qsB + w * 16;
    constexpr int STEP_BYTES = 256;                          // 32 samples of 8 bytes

    f32x16 acc[NBW][KBW];
#pragma unroll
    for (int x = 0; x < NBW; ++x)
#pragma unroll
        for (int y = 0; y < KBW; ++y)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[x][y][r] = 0.f;
    float bsum[LOADS][4];
#pragma unroll
    for (int k = 0; k < LOADS; ++k)
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) bsum[k][jj] = 0.f;

    u32x4_t stage[PF][LOADS];
    auto gload = [&](int step, int slot) {
#pragma unroll
        for (int k = 0; k < LOADS; ++k) {
            const bool isA = k < LA;
            const int q0 = (isA ? k : k - LA) * QPL;         // first quad of this load
            stage[slot][k] = isA ? __builtin_bit_cast(u32x4_t, __builtin_amdgcn_raw_buffer_load_b128(rsA, voffA, step * STEP_BYTES + q0 * qsA, 0))
                                 : __builtin_bit_cast(u32x4_t, __builtin_amdgcn_raw_buffer_load_b128(rsB, voffB, step * STEP_BYTES + q0 * qsB, 0));
        }
    };
    auto lds_off = [&](int row, int ww) { return row * 64 + ((((ww >> 2) ^ (row >> 2)) & 3) << 4) + ((ww & 3) << 2); };
    auto lstore = [&](int buf, int slot) {
        char* base = ldsb + buf * BUF_BYTES;
#pragma unroll
        for (int k = 0; k < LOADS; ++k) {
            const bool isA = k < LA;
            const int row0 = ((isA ? k : k - LA) * QPL + quad) * 4;                  // row inside the operand's tile
            if (row0 >= (isA ? TN : TK)) continue;                                    // the idle part of a load that overhangs the tile
            const u32x4_t v = stage[slot][k];                 // {rows 0,1 | rows 2,3} of sample 2w, then of sample 2w + 1
            unsigned d[4];
            d[0] = __builtin_amdgcn_perm(v[2], v[0], 0x05040100u);
            d[1] = __builtin_amdgcn_perm(v[2], v[0], 0x07060302u);
            d[2] = __builtin_amdgcn_perm(v[3], v[1], 0x05040100u);
            d[3] = __builtin_amdgcn_perm(v[3], v[1], 0x07060302u);
            const int row = (isA ? 0 : TN) + row0;
#pragma unroll
            for (int jj = 0; jj < 4; ++jj) *reinterpret_cast<unsigned*>(base + lds_off(row + jj, w)) = d[jj];
            if ((bias_side == 1 && isA) || (bias_side == 2 && !isA)) {
#pragma unroll
                for (int jj = 0; jj < 4; ++jj) bsum[k][jj] += __builtin_bit_cast(float, d[jj] << 16) + __builtin_bit_cast(float, d[jj] & 0xffff0000u);
            }
        }
    };
    auto step_body = [&](int s, auto slot_c) {
        constexpr int slot = decltype(slot_c)::value;
        const int buf = s & 1;
        if (s + PF < nsteps) gload(step0 + s + PF, slot);
        const char* base = ldsb + buf * BUF_BYTES;
        auto frag = [&](int row, int ks) {
            return *reinterpret_cast<const u32x4_t*>(base + row * 64 + ((((2 * ks + h) ^ (row >> 2)) & 3) << 4));
        };
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            u32x4_t af[NBW];
#pragma unroll
            for (int x = 0; x < NBW; ++x) af[x] = frag((wn * NBW + x) * 32 + i, ks);
#pragma unroll
            for (int y = 0; y < KBW; ++y) {
                const u32x4_t bf = frag(TN + (wk * KBW + y) * 32 + i, ks);
#pragma unroll
                for (int x = 0; x < NBW; ++x)
                    if (live[x][y]) acc[x][y] = mfma_bf16(af[x], bf, acc[x][y]);
            }
            if (ks == 0 && s + 1 < nsteps) lstore(buf ^ 1, (slot + 1) % PF);
        }
        __syncthreads();
    };

    if (nsteps > 0) {
        gload(step0, 0);
        lstore(0, 0);
    }
    if (PF >= 2 && nsteps > 1) gload(step0 + 1, 1 % PF);
    if (PF >= 3 && nsteps > 2) gload(step0 + 2, 2 % PF);
    __syncthreads();
    for (int s = 0; s < nsteps; s += PF) {
        step_body(s, std::integral_constant<int, 0>{});
        if (PF >= 2 && s + 1 < nsteps) step_body(s + 1, std::integral_constant<int, 1 % PF>{});
        if (PF >= 3 && s + 2 < nsteps) step_body(s + 2, std::integral_constant<int, 2 % PF>{});
    }
    float* out = partial + ((long long)blockIdx.y * gridDim.x + blockIdx.x) * (TN * TK + 256);
#pragma unroll
    for (int x = 0; x < NBW; ++x)
#pragma unroll
        for (int y = 0; y < KBW; ++y)
#pragma unroll
            for (int r = 0; r < 16; ++r)
                out[((wn * NBW + x) * 32 + acc_row(r, h)) * TK + (wk * KBW + y) * 32 + i] = acc[x][y][r];
    // bias sums: fold the 16 sample pairs of a quad's lanes (fixed xor tree), one write per row
    if (tid < 256) out[TN * TK + tid] = 0.f;
    __syncthreads();
    if (bias_side) {
#pragma unroll
        for (int k = 0; k < LOADS; ++k) {
            const bool isA = k < LA;
            if ((bias_side == 1 && isA) || (bias_side == 2 && !isA)) {
#pragma unroll
                for (int jj = 0; jj < 4; ++jj) {
                    float v = bsum[k][jj];
#pragma unroll
                    for (int o = 8; o > 0; o >>= 1) v += __shfl_xor(v, o);
                    const int row = ((isA ? k : k - LA) * QPL + quad) * 4 + jj;
                    if (w == 0 && row < 256 && row < (isA ? TN : TK)) out[TN * TK + row] = v;
                }
            }
        }
    }
}

// Deterministic reduction of the partial tiles + scatter into the flat parameter gradient: ONE launch for all pieces of a
// network (blockIdx.y = piece).
struct ReduceArgs {
    long long partial_off;   // floats from the start of the partial workspace to this piece's [nsplit][TN*TK + 256] tiles
    int nsplit, TN, TK;
    int layer;        // 0..9
    int n_off;        // kernel-row offset of tile row 0 (dY side)
    int k_off;        // slot offset of tile column 0 (X side)
    int transposed;   // tile is [slot][row] instead of [row][slot]
    int bias;         // 0: none, 1: bias sums indexed by the dY row
};
constexpr int kMaxPieces = 40;
struct ReduceBatch {
    ReduceArgs r[kMaxPieces];
};

// fixed-order sum over the split-M partial tiles with 8 independent accumulators (8 loads in flight
// per thread instead of a 256-long dependent chain); the order does not depend on timing.
__device__ __forceinline__ float sum_partials(const float* __restrict__ p, long long stride, int n) {
    float s[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    int w = 0;
    // sixteen loads in flight per trip (the nine partial tiles of a short reduction: one round trip, not two); same order of additions
    // as the eight-at-a-time loop of rounds 1-4: whole groups of eight to the eight accumulators, a remainder to s[0]
    for (; w + 16 <= n; w += 16) {
        float x[16];
#pragma unroll
        for (int k = 0; k < 16; ++k) x[k] = p[(long long)(w + k) * stride];
#pragma unroll
        for (int k = 0; k < 16; ++k) s[k & 7] += x[k];
    }
    if (w < n) {
        float x[16];
#pragma unroll
        for (int k = 0; k < 16; ++k) x[k] = w + k < n ? p[(long long)(w + k) * stride] : 0.f;
        const int full = (n - w) & ~7;
#pragma unroll
        for (int k = 0; k < 16; ++k)
            if (w + k < n) { if (k < full) s[k & 7] += x[k]; else s[0] += x[k]; }
    }
    return ((s[0] + s[1]) + (s[2] + s[3])) + ((s[4] + s[5]) + (s[6] + s[7]));
}

__global__ void dw_reduce_kernel(ReduceBatch batch, const float* __restrict__ partial_base, float* __restrict__ d_params) {
    const ReduceArgs a = batch.r[blockIdx.y];
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    const int tile = a.TN * a.TK;
    const int stride = tile + 256;
    const float* partial = partial_base + a.partial_off;
    if (idx < tile) {
        const int tr = idx / a.TK, tc = idx % a.TK;
        const int n = a.n_off + (a.transposed ? tc : tr), s = a.k_off + (a.transposed ? tr : tc);
        const int row = out_row(a.layer, n), col = fwd_slot_col(a.layer, s);
        if (row < 0 || col < 0) return;
        d_params[weight_off(a.layer) + row * layer_k(a.layer) + col] = sum_partials(partial + idx, stride, a.nsplit);
    } else if (a.bias && idx < tile + 256) {
        const int b = idx - tile;
        const int row = out_row(a.layer, a.n_off + b);
        if (row < 0 || b >= (a.transposed ? a.TK : a.TN)) return;
        d_params[bias_off(a.layer) + row] = sum_partials(partial + idx, stride, a.nsplit);
    }
}

// ---- The two GEMV-shaped pieces of a network's weight gradient -- the density row (dsigma . h7^T: 1 x 256) and the colour rows
// (d rgb_raw . hr^T: 3 x 128) -- on the vector ALU.  As pieces of the skinny MFMA launch they cost a 256 x 64 tile of matrix work each
// for 1 / 3 useful columns and 0.35 ms of a cfg2 step (HISTORY.md): they are pure HBM time (1.5 KB per sample) with next to no
// arithmetic, so they run on a second stream BESIDE the wide launch, which is matrix-bound and leaves 70 % of the HBM rate unused.
// One wave: Q quads (4 Q rows) of X over one chunk of samples, 64 lanes = 64 consecutive samples of a quad (1 KB per load
// instruction), R dY rows; partial tiles in the reducer's format ([chunk][tile + 256]: transposed tile [slot][row], then the bias sums).
struct HeadsArgs {
    const float* X[2];     // first quad of h7 / hr in the forward's workspace image
    const float* D[2];     // the quad that holds dsigma (element 0) / the three d rgb_raw rows (elements 0..2)
    float* partial[2];     // [chunk][256 * 1 + 256] / [chunk][128 * 3 + 256]
    long long mpad;
    int per;               // samples per chunk
};

typedef float f32x2 __attribute__((ext_vector_type(2)));

// acc += x * d (both halves by the SAME scalar d): the scalar is element SEL of the register pair `dpair` -- selected by op_sel, not
// copied into a pair first.  The kernel shares its SIMDs with the matrix waves of the wide launch (fp32 MFMA and vector instructions of
// one SIMD do not overlap, tools/mfma_valu_contention.hip) and has 64 registers, so its loop is written to the minimum: one
// v_pk_fma_f32 per two products, buffer loads with scalar offsets (no address arithmetic), no moves.
template <int SEL>
__device__ __forceinline__ void pk_fma_bcast(f32x2& acc, f32x2 x, f32x2 dpair) {
    if constexpr (SEL == 0) asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[1,0,1]" : "+v"(acc) : "v"(x), "v"(dpair));
    else asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[0,1,0] op_sel_hi:[1,1,1]" : "+v"(acc) : "v"(x), "v"(dpair));
}

// Sum each of N per-lane values over the 64 lanes in ~3 N vector instructions instead of 6 N: at step s the lanes whose bit s is set keep
// the upper half of the values and hand the lower half to their partner (and vice versa), so the count halves with every exchange.  On
// return lane l < N holds the total of value bitrev(l) (bit reversal over log2 N bits).
template <int N, int MASK = 1, int CNT = N>
__device__ __forceinline__ float reduce_pack(float (&v)[N]) {
    if constexpr (MASK >= 64) return v[0];
    else if constexpr (CNT > 1) {
        constexpr int H = CNT / 2;
        const bool upper = (threadIdx.x & MASK) != 0;
#pragma unroll
        for (int i = 0; i < H; ++i) {
            const float give = upper ? v[i] : v[i + H], keep = upper ? v[i + H] : v[i];
            v[i] = keep + __shfl_xor(give, MASK, 64);
        }
        return reduce_pack<N, MASK * 2, H>(v);
    } else {
        v[0] += __shfl_xor(v[0], MASK, 64);
        return reduce_pack<N, MASK * 2, 1>(v);
    }
}
template <int BITS>
__device__ __forceinline__ int bitrev(int x) {
    int r = 0;
#pragma unroll
    for (int b = 0; b < BITS; ++b) r |= ((x >> b) & 1) << (BITS - 1 - b);
    return r;
}

// Q quads of X (from quad `quad0`) times R rows of the dY quad over the samples [m0, m1) (a multiple of 128 long), one wave
template <int Q, int R, bool BIAS>
__device__ __forceinline__ void heads_body(rsrc_t rsX, rsrc_t rsD, long long mpad, long long m0, long long m1, int quad0, float* __restrict__ out,
                                           int tile) {
    const int lane = threadIdx.x & 63;
    f32x2 acc[Q][R][2];
    f32x2 bs[2] = {f32x2{0.f, 0.f}, f32x2{0.f, 0.f}};
#pragma unroll
    for (int q = 0; q < Q; ++q)
#pragma unroll
        for (int r = 0; r < R; ++r) acc[q][r][0] = acc[q][r][1] = f32x2{0.f, 0.f};
    const int voff = lane * 16;
    // 32-bit scalar byte offsets (the host keeps every operand below 2 GiB)
    const int qstride = (int)(mpad * 16), trips = (int)((m1 - m0) >> 7);
    int soff = (int)(m0 * 16);
    for (int trip = 0; trip < trips; ++trip, soff += 2048) {
        // two 64-sample groups per trip: 2 (Q + 1) loads of 16 bytes in flight per lane
        f32x4 x[2][Q], d[2];
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            d[t] = buf_load4(rsD, voff, soff + 1024 * t);
#pragma unroll
            for (int q = 0; q < Q; ++q) x[t][q] = buf_load4(rsX, voff, (quad0 + q) * qstride + soff + 1024 * t);
        }
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            if (BIAS) {
                bs[0] += d[t].lo;
                bs[1] += d[t].hi;
            }
#pragma unroll
            for (int q = 0; q < Q; ++q) {
                pk_fma_bcast<0>(acc[q][0][0], x[t][q].lo, d[t].lo);
                pk_fma_bcast<0>(acc[q][0][1], x[t][q].hi, d[t].lo);
                if constexpr (R == 3) {
                    pk_fma_bcast<1>(acc[q][1][0], x[t][q].lo, d[t].lo);
                    pk_fma_bcast<1>(acc[q][1][1], x[t][q].hi, d[t].lo);
                    pk_fma_bcast<0>(acc[q][2][0], x[t][q].lo, d[t].hi);
                    pk_fma_bcast<0>(acc[q][2][1], x[t][q].hi, d[t].hi);
                }
            }
        }
    }
    // the tile is [slot][row]: value index (4 q + j) R + r
    constexpr int N = Q * 4 * R;
    float v[N];
#pragma unroll
    for (int q = 0; q < Q; ++q)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int r = 0; r < R; ++r) v[(q * 4 + j) * R + r] = acc[q][r][j >> 1][j & 1];
    float* dst = out + quad0 * 4 * R;
    if constexpr (N == 16) {
        const float tot = reduce_pack<16>(v);
        if (lane < 16) dst[bitrev<4>(lane)] = tot;
    } else {
        static_assert(N == 24, "4 quads x 1 row or 2 quads x 3 rows");
        float lo[16], hi[8];
#pragma unroll
        for (int k = 0; k < 16; ++k) lo[k] = v[k];
#pragma unroll
        for (int k = 0; k < 8; ++k) hi[k] = v[16 + k];
        const float t0 = reduce_pack<16>(lo), t1 = reduce_pack<8>(hi);
        if (lane < 16) dst[bitrev<4>(lane)] = t0;
        if (lane < 8) dst[16 + bitrev<3>(lane)] = t1;
    }
    if constexpr (BIAS) {
        float b[4] = {bs[0][0], bs[0][1], bs[1][0], bs[1][1]};
        const float tot = reduce_pack<4>(b);
        if (lane < R) out[tile + bitrev<2>(lane)] = tot;
    }
}

// grid (chunks), 4 waves: wave w walks quad groups w, w + 4, w + 8, w + 12 of h7 (4 quads each) and of hr (2 quads each) over the chunk's
// samples; with <= 256 chunks that is at most one wave per SIMD of the chip.  64 REGISTERS: the wide launch keeps two waves of 222 (224)
// registers on every SIMD, which leaves exactly 64 -- a heads wave that needs more cannot join them; its workgroups then take CUs of their
// own between the wide launch's rounds and make that launch's workgroups wait for them (measured with 114 registers: the wide launch
// +290..390 us, the heads kernel stretched to 2.9 ms, i.e. all of the gain lost).
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(8, 8))) void dw_heads_kernel(HeadsArgs a) {
    const int wave = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);
    const long long m0 = (long long)blockIdx.x * a.per, m1 = m0 + a.per < a.mpad ? m0 + a.per : a.mpad;
    float* out0 = a.partial[0] + (long long)blockIdx.x * (256 + 256);
    float* out1 = a.partial[1] + (long long)blockIdx.x * (384 + 256);
    // descriptors end with the operand: X = 64 / 32 quads of mpad samples; the dY quad = mpad samples (host: < 2 GiB each)
    const rsrc_t rsX0 = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.X[0]), 0, (int)(64 * a.mpad * 16), 0x00020000);
    const rsrc_t rsX1 = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.X[1]), 0, (int)(32 * a.mpad * 16), 0x00020000);
    const rsrc_t rsD0 = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.D[0]), 0, (int)(a.mpad * 16), 0x00020000);
    const rsrc_t rsD1 = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.D[1]), 0, (int)(a.mpad * 16), 0x00020000);
    if (wave == 0) heads_body<4, 1, true>(rsX0, rsD0, a.mpad, m0, m1, 0, out0, 256);
    else heads_body<4, 1, false>(rsX0, rsD0, a.mpad, m0, m1, wave * 4, out0, 256);
    for (int g = 1; g < 4; ++g) heads_body<4, 1, false>(rsX0, rsD0, a.mpad, m0, m1, (wave + 4 * g) * 4, out0, 256);
    if (wave == 0) heads_body<2, 3, true>(rsX1, rsD1, a.mpad, m0, m1, 0, out1, 384);
    else heads_body<2, 3, false>(rsX1, rsD1, a.mpad, m0, m1, wave * 2, out1, 384);
    for (int g = 1; g < 4; ++g) heads_body<2, 3, false>(rsX1, rsD1, a.mpad, m0, m1, (wave + 4 * g) * 2, out1, 384);
}

// the stream the heads kernel runs on, and the two events that fork it from / join it to the caller's stream (all capturable; created on
// first use, which therefore must not happen under stream capture: niw_train_step_prepare() and any launched warm-up iteration do it)
struct HeadsLane {
    hipStream_t s = nullptr;
    hipEvent_t fork = nullptr, join = nullptr;
    std::mutex mu;      // one caller at a time between fork and join: the events are shared by every caller of the device
};

// Between fork and join the lane belongs to ONE call (two host threads, or two trainers on different streams, would otherwise re-record
// each other's events); and a call that fails after the fork still joins the lane -- an unjoined stream invalidates a capture.
struct HeadsLaneHold {
    HeadsLane* lane = nullptr;
    hipStream_t st = nullptr;
    bool forked = false;
    void take(HeadsLane* l, hipStream_t s) { lane = l; st = s; lane->mu.lock(); }
    hipError_t join() {
        if (!lane || !forked) return hipSuccess;
        forked = false;
        return hipStreamWaitEvent(st, lane->join, 0);
    }
    ~HeadsLaneHold() {
        if (!lane) return;
        if (forked) {                          // error path: whatever the lane holds joins the caller's stream
            (void)hipEventRecord(lane->join, lane->s);
            (void)hipStreamWaitEvent(st, lane->join, 0);
        }
        lane->mu.unlock();
    }
};

HeadsLane* heads_lane() {
    static HeadsLane lanes[64];
    static std::atomic<unsigned long long> ready{0ull};
    static std::atomic_flag busy = ATOMIC_FLAG_INIT;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return nullptr;
    const unsigned long long bit = 1ull << dev;
    if (ready.load(std::memory_order_acquire) & bit) return &lanes[dev];
    while (busy.test_and_set(std::memory_order_acquire)) {}
    bool ok = true;
    if (!(ready.load(std::memory_order_acquire) & bit)) {
        HeadsLane& L = lanes[dev];
        ok = hipStreamCreateWithFlags(&L.s, hipStreamNonBlocking) == hipSuccess;       // normal priority (niw_step.hip: SideLane)
        ok = ok && hipEventCreateWithFlags(&L.fork, hipEventDisableTiming) == hipSuccess;
        ok = ok && hipEventCreateWithFlags(&L.join, hipEventDisableTiming) == hipSuccess;
        if (ok) ready.fetch_or(bit, std::memory_order_release);
    }
    busy.clear(std::memory_order_release);
    return ok ? &lanes[dev] : nullptr;
}

// every group's partial tiles live side by side until the single reduction: <= 511 tiles of 256 x 256 (7 x 73 splits),
// <= 508 of 256 x 64 (4 x 127), <= 256 of 128 x 320
// + the vector-ALU head pieces: <= 256 chunks of (256 x 1 + 256) and (128 x 3 + 256)
constexpr int kHeadChunks = 256;
constexpr long long kPartialTileFloats = 511ll * (256 * 256 + 256) + 512ll * (256 * 64 + 256) + 256ll * (128 * 320 + 256) + kHeadChunks * (512ll + 640ll);

struct Piece {
    int layer;
    int a_row, a_rows;   // dY-side operand: first row in gradws / valid rows   (X side when transposed)
    int b_row, b_rows;   // X-side operand: first row in save / valid rows
    int n_off, k_off, transposed, bias, wide;   // wide: 256x256 tile, else 256x64
};

// TERMS: 0 exact fp32; 3 bf16x3 (fp32 workspaces); 1 bf16 on bf16 workspaces (dw_gemm_half_kernel); -1 bf16 on fp32 workspaces (diagnostic)
template <int WN, int WK, int NBW, int KBW, bool SKIP, int PF, bool QUAD, int TERMS>
constexpr auto fast_or_exact() {
    if constexpr (TERMS == 0) return dw_gemm_kernel<WN, WK, NBW, KBW, SKIP, PF, QUAD>;
    else if constexpr (TERMS == 1) return dw_gemm_half_kernel<WN, WK, NBW, KBW, SKIP, PF>;
    else return dw_gemm_fast_kernel<WN, WK, NBW, KBW, SKIP, PF, (TERMS < 0 ? 1 : TERMS)>;
}

template <int WN, int WK, int NBW, int KBW, bool SKIP, int PF, bool QUAD, int TERMS = 0>
int launch_gemm(const GemmBatch& batch, long long mpad, int batches, float* partial, int* nsplit_out, hipStream_t st) {
    constexpr int TN = WN * NBW * 32, TK = WK * KBW * 32;
    const int steps_total = (int)(mpad / 32);
    // at least 8 slices per workgroup; one workgroup per CU, or -- for a batch of equal pieces -- just under two
    // full rounds of the 256 CUs in total (fewer, longer workgroups: less partial-tile traffic for the reducer)
    // (two rounds balance better at >= 130 k samples; below, one round halves the partial-tile traffic that then dominates:
    // measured 216 / 350 / 601 us against 252 / 374 / 629 us for the whole dW group at 16 k / 32 k / 65 k samples)
    const int rounds = steps_total <= 4096 ? 1 : 2;
    // (tried in round 4: 128 workgroups of 8 slices instead of 255 of 4 for the colour layer of a 1/8 share -- half the partial tiles, but the
    // launch itself took 47 us instead of 36 with half the CUs idle, more than the reducer saved)
    const int cap = batches >= 2 ? (rounds * 256 - 1) / batches : 256;
    // short reductions (the warp's few thousand points, a 1/8 ray shard): fewer slices per workgroup, down to 2, until the
    // launch has a workgroup for every CU -- the partial tile each workgroup writes (<= 256 KB) is the price of a split
    int min_slices = 8;
    while (min_slices > 2 && (long long)((steps_total + min_slices - 1) / min_slices) * batches < 256) min_slices >>= 1;
    int nsplit = (steps_total + min_slices - 1) / min_slices;
    nsplit = nsplit < 1 ? 1 : (nsplit > cap ? cap : nsplit);
    const int per = (steps_total + nsplit - 1) / nsplit;
    nsplit = (steps_total + per - 1) / per;
    static_assert(TERMS == 0 || QUAD, "the fast-precision loader reads quad-row images");
    const size_t lds = TERMS == 0 ? 2 * (size_t)(TN + TK) * kLdsStride * sizeof(float) : 2 * (size_t)(TERMS >= 3 ? 2 : 1) * (TN + TK) * 64;
    auto kern = fast_or_exact<WN, WK, NBW, KBW, SKIP, PF, QUAD, TERMS>();
    static std::atomic<unsigned long long> attr_set{0ull};
    if (int rc = niw_ensure_dynamic_lds(reinterpret_cast<const void*>(kern), lds, attr_set, "NT GEMM")) return rc;
    for (int b = 0; b < batches; ++b) {
        // 32-bit byte offsets inside an operand (buffer addressing)
        const long long ea = (long long)(batch.A[b].rows < TN ? batch.A[b].rows : TN) * batch.A[b].row_stride * 4;
        const long long eb = (long long)(batch.B[b].rows < TK ? batch.B[b].rows : TK) * batch.B[b].row_stride * 4;
        if (ea >= (1ll << 31) || eb >= (1ll << 31) || mpad > batch.A[b].row_stride || mpad > batch.B[b].row_stride) {
            niw_set_error("NT GEMM: operand of %lld rows x %lld samples exceeds the 2 GiB reach of a buffer descriptor (or rows shorter than the reduction)",
                          (long long)(ea >= eb ? batch.A[b].rows : batch.B[b].rows), mpad);
            return NIW_ERR_INVALID_ARG;
        }
    }
    if constexpr (TERMS == 0) kern<<<dim3(nsplit, batches), 64 * WN * WK, lds, st>>>(batch, steps_total, per, partial, BlockMap{0, 0, 0, 0});
    else kern<<<dim3(nsplit, batches), 64 * WN * WK, lds, st>>>(batch, steps_total, per, partial);
    NIW_LAUNCH_CHECK("NT GEMM");
    *nsplit_out = nsplit;
    return NIW_OK;
}

template <bool QUAD, int TERMS = 0>
int launch_shape(int wide, const GemmBatch& batch, long long mpad, int batches, float* partial, int* nsplit_out, hipStream_t st) {
    // tile shapes: 0 = 256 x 64, 1 = 256 x 256, 2 = 128 x 288 (the colour layer: 128 outputs x [256 features + 32 view slots])
#ifndef NIW_DW_COLOUR_288
    if (wide == 2) return launch_gemm<4, 2, 1, 5, true, NIW_DW_PF_COLOUR, QUAD, TERMS>(batch, mpad, batches, partial, nsplit_out, st);
#else
    if (wide == 2) return launch_gemm<4, 1, 1, 9, false, NIW_DW_PF_COLOUR, QUAD, TERMS>(batch, mpad, batches, partial, nsplit_out, st);
#endif
    return wide ? launch_gemm<4, 2, 2, 4, false, 1, QUAD, TERMS>(batch, mpad, batches, partial, nsplit_out, st)
                : launch_gemm<8, 1, 1, 2, true, NIW_DW_PF_SKINNY, QUAD, TERMS>(batch, mpad, batches, partial, nsplit_out, st);
}

// The seven 256 x 256 products of a network over a SHORT reduction (<= 32 k samples: a rank's 1/8 share), as 7 x 4 quadrant tiles of
// 128 x 128 in ONE round of workgroups: 9 sample ranges per tile instead of 36 per product, i.e. a quarter of the partial-tile bytes
// (16.5 MB instead of 66 MB at 32 k samples, which the reducer then re-reads), while the four tiles of a product and sample range sit on
// one XCD (BlockMap) and share the operand rows they all read through its L2.
int launch_wide_quadrants(const GemmBatch& batch, int n_sets, long long mpad, float* partial, int* nsplit_out, hipStream_t st) {
    constexpr int WN = 4, WK = 2, NBW = 1, KBW = 2, TN = 128, TK = 128;
    const int steps_total = (int)(mpad / 32);
    // (NIW_DW_QUAD_GROUPS: diagnostic -- groups of four tiles per launch; 64 = one workgroup per CU, 128 = two co-resident ones)
    static const int quad_groups = [] { const char* e = getenv("NIW_DW_QUAD_GROUPS"); const int v = e ? atoi(e) : 64; return v < 8 ? 8 : v; }();
    int nsplit = quad_groups / n_sets;                         // 7 products: 9 ranges -> 63 groups, padded to 64
    if (nsplit > (steps_total + 1) / 2) nsplit = (steps_total + 1) / 2;
    if (nsplit < 1) nsplit = 1;
    const int per = (steps_total + nsplit - 1) / nsplit;
    nsplit = (steps_total + per - 1) / per;
    const int groups = n_sets * nsplit, groups_pad = (groups + 7) / 8 * 8;
    const size_t lds = 2 * (size_t)(TN + TK) * kLdsStride * sizeof(float);
    auto kern = dw_gemm_kernel<WN, WK, NBW, KBW, false, 1, true>;
    static std::atomic<unsigned long long> attr_set{0ull};
    if (int rc = niw_ensure_dynamic_lds(reinterpret_cast<const void*>(kern), lds, attr_set, "NT GEMM (quadrants)")) return rc;
    for (int b = 0; b < 4 * n_sets; ++b) {
        const long long ea = (long long)batch.A[b].rows * batch.A[b].row_stride * 4, eb = (long long)batch.B[b].rows * batch.B[b].row_stride * 4;
        if (ea >= (1ll << 31) || eb >= (1ll << 31) || mpad > batch.A[b].row_stride || mpad > batch.B[b].row_stride) {
            niw_set_error("NT GEMM (quadrants): operand exceeds the 2 GiB reach of a buffer descriptor");
            return NIW_ERR_INVALID_ARG;
        }
    }
    kern<<<dim3(4 * groups_pad, 1), 64 * WN * WK, lds, st>>>(batch, steps_total, per, partial, BlockMap{groups_pad, groups, nsplit, 4});
    NIW_LAUNCH_CHECK("NT GEMM (quadrants)");
    *nsplit_out = nsplit;
    return NIW_OK;
}

}  // namespace

int niw_launch_nt_gemm(int wide, NiwGemmOperand A, NiwGemmOperand B, long long mpad, int batches, float* partial,
                       int bias_side, int* nsplit_out, hipStream_t st) {
    // `batches` products whose operands are batch_stride floats apart
    if (batches < 1 || batches > kMaxBatch) {
        niw_set_error("NT GEMM: %d batches (1..%d)", batches, kMaxBatch);
        return NIW_ERR_INVALID_ARG;
    }
    GemmBatch gb{};
    for (int b = 0; b < batches; ++b) {
        gb.A[b] = A; gb.A[b].p = A.p + (long long)b * A.batch_stride;
        gb.B[b] = B; gb.B[b].p = B.p + (long long)b * B.batch_stride;
        gb.bias_side[b] = bias_side;
    }
    return launch_shape<false>(wide, gb, mpad, batches, partial, nsplit_out, st);
}

// `n` products with operand pairs of their own in ONE launch of the 256 x 256 tile with SKIP (32 x 32 blocks beyond an operand's valid rows
// issue no MFMAs): the warp backward's two GEMM families over a short reduction (niw_warp.hip) -- [G x E] and [H x head gradients], the
// second with 4 valid columns -- which as two launches of one round each were 2 x 33 us of a second-stream chain that closes the iteration
int niw_launch_nt_gemm_pairs(int n, const NiwGemmOperand* A, const NiwGemmOperand* B, const int* bias_side, long long mpad, float* partial,
                             int* nsplit_out, hipStream_t st) {
    if (n < 1 || n > kMaxBatch) {
        niw_set_error("NT GEMM: %d products (1..%d)", n, kMaxBatch);
        return NIW_ERR_INVALID_ARG;
    }
    GemmBatch gb{};
    for (int b = 0; b < n; ++b) { gb.A[b] = A[b]; gb.B[b] = B[b]; gb.bias_side[b] = bias_side[b]; }
    return launch_gemm<4, 2, 2, 4, true, 1, false>(gb, mpad, n, partial, nsplit_out, st);
}

int niw_dw_heads_prepare() { return heads_lane() ? NIW_OK : NIW_ERR_LAUNCH; }

extern "C" int64_t niw_mlp_bwd_workspace_floats(int64_t n_rays, int n_samples) {
    (void)n_rays; (void)n_samples;
    return kPartialTileFloats;
}

extern "C" int niw_mlp_bwd_dx(const float* packed, const float* center, const float* ray, const float* depth,
                              int64_t n_rays, int n_samples, int density_activ, int precision,
                              const float* rgb, const float* d_rgb, const float* d_sigma,
                              const float* save, float* gradws, float* d_center, float* d_ray, niw_stream_t stream) {
    NIW_REQUIRE(packed && center && ray && depth && rgb && d_rgb && d_sigma && save && gradws, "niw_mlp_bwd_dx: null pointer");
    NIW_REQUIRE((d_center == nullptr) == (d_ray == nullptr), "niw_mlp_bwd_dx: d_center and d_ray must both be given or both be NULL");
    NIW_REQUIRE(n_rays > 0 && n_samples > 0, "niw_mlp_bwd_dx: empty input");
    NIW_REQUIRE(density_activ == NIW_ACT_RELU || density_activ == NIW_ACT_SOFTPLUS, "niw_mlp_bwd_dx: unknown density activation %d", density_activ);
    NIW_REQUIRE(niw_mlp_padded_rows(n_rays, n_samples) < (1ll << 24), "niw_mlp_bwd_dx: too many samples per call");
    NIW_REQUIRE(precision == NIW_PREC_FP32 || precision == NIW_PREC_BF16X3 || precision == NIW_PREC_BF16, "niw_mlp_bwd_dx: unknown precision %d", precision);
    if (precision != NIW_PREC_FP32)
        return niw_launch_mlp_bwd_dx_fast(precision, packed, center, ray, depth, n_rays, n_samples, density_activ, rgb, d_rgb, d_sigma, save, gradws,
                                          d_center, d_ray, (hipStream_t)stream);
    return niw_launch_mlp_bwd_dx(packed, center, ray, depth, n_rays, n_samples, density_activ, rgb, d_rgb, d_sigma, save, gradws,
                                 d_center, d_ray, (hipStream_t)stream);
}

extern "C" int niw_mlp_bwd_dw(const float* save, const float* gradws, int64_t n_rays, int n_samples, int precision, float* partial,
                              float* d_params, niw_stream_t stream) {
    NIW_REQUIRE(save && gradws && partial && d_params, "niw_mlp_bwd_dw: null pointer");
    NIW_REQUIRE(precision == NIW_PREC_FP32 || precision == NIW_PREC_BF16X3 || precision == NIW_PREC_BF16, "niw_mlp_bwd_dw: unknown precision %d", precision);
    NIW_REQUIRE(n_rays > 0 && n_samples > 0, "niw_mlp_bwd_dw: empty input");
    const long long mpad = niw_mlp_padded_rows(n_rays, n_samples);
    hipStream_t st = (hipStream_t)stream;
    // dW pieces.  Non-transposed: tile rows = dY rows (gradws), tile columns = X slots (save).
    // Transposed (skinny dY: the density row, the 3 colour rows): tile rows = X slots, columns = dY rows.
    // Launches per network: [layers 1..7: seven 256 x 256 products] [256 x 64 pieces: the encoding columns of layers 0 and 4 -- in the
    // fast-precision modes and below 96 k samples also the density row and the colour rows] [the colour layer, 128 x 288 in a 128 x 320
    // tile] [one reduction of all partial-tile sets]; from 96 k samples in exact mode the density row and the colour rows are
    // dw_heads_kernel's, on a second stream beside the first launch.  (Tried: the density row as a 257th row of layer 7's product,
    // formed on the vector ALU from the staged h6 slice -- the skinny launch lost 75 us of 506 at 523 k samples, but the wide kernel paid
    // 80 us for the mere presence of the code and 50 more for running it; as an MFMA row block of that product: HISTORY.md.)
    static_assert(kGradY7 == 7 * 256, "dY blocks of layers 0..7 are uniformly strided");
    static_assert(kSaveEnc % 4 == 0 && kSaveH1 % 4 == 0 && kSaveFeat % 4 == 0 && kSaveHr % 4 == 0 && kGradY7 % 4 == 0 &&
                  kGradRgb0 % 4 == 0 && kGradRgb1 % 4 == 0 && kSaveSigma % 4 == 0, "operands of the quad-row images start on whole quads");
    const Piece wide[7] = {
        // layer, a_row, a_rows, b_row, b_rows, n_off, k_off, transposed, bias, wide
        {1, 1 * 256, 256, save_h(1), 256, 0, 0, 0, 1, 1}, {2, 2 * 256, 256, save_h(2), 256, 0, 0, 0, 1, 1},
        {3, 3 * 256, 256, save_h(3), 256, 0, 0, 0, 1, 1}, {4, 4 * 256, 256, save_h(4), 256, 0, 0, 0, 1, 1},
        {5, 5 * 256, 256, save_h(5), 256, 0, 0, 0, 1, 1}, {6, 6 * 256, 256, save_h(6), 256, 0, 0, 0, 1, 1},
        {7, 7 * 256, 256, save_h(7), 256, 0, 0, 0, 1, 1}};
    const Piece skinny[4] = {
        {0, 0 * 256, 256, kSaveEnc, 64, 0, 0, 0, 1, 0},                  // first layer: 256 x 64 encoding slots
        {4, 4 * 256, 256, kSaveEnc, 64, 0, 256, 0, 0, 0},                // skip connection: encoding columns of layer 4
        {7, save_h(7), 256, kGradY7 + 256, 1, 256, 0, 1, 1, 0},          // density row (transposed)
        {9, kSaveHr, 128, kGradRgb1, 3, 0, 0, 1, 1, 0}};                 // colour rows (transposed)
    const Piece colour[1] = {{8, kGradRgb0, 128, kSaveFeat, 288, 0, 0, 0, 1, 2}};   // feat rows and the 32 view-slot rows are contiguous
    // exact mode: the density row and the colour rows leave the skinny launch for dw_heads_kernel on a second stream (NIW_DW_HEADS=0: the
    // four-piece skinny launch as in the fast-precision modes, whose workspaces are bf16 images; =1: heads kernel on the caller's stream)
    // from 96,000 samples (NIW_DW_HEADS_MIN slices of 32; rounds 4: 131,072): below, fork + join cost more than the two pieces.  Measured in
    // round 5: 129 k samples 1121 -> 1067 us for the group; 98 k (the fine pass of a rank's 1/8 share of cfg2) level stand-alone and
    // the share's step 3.368 -> 3.337 ms; at 32 k the step is 0.6 % SLOWER with the heads kernel (0.944 vs 0.938 ms)
    static const int heads_mode = [] { const char* e = getenv("NIW_DW_HEADS"); return e ? atoi(e) : 2; }();
    static const long long heads_min = [] { const char* e = getenv("NIW_DW_HEADS_MIN"); return e ? atoll(e) : 3000ll; }();
    const bool heads = precision == NIW_PREC_FP32 && heads_mode > 0 && mpad / 32 >= heads_min;
    struct Group { const Piece* p; int n, wide, TN, TK; };
    const Group groups[3] = {{wide, 7, 1, 256, 256}, {skinny, heads ? 2 : 4, 0, 256, 64}, {colour, 1, 2, 128, kColourTK}};
    ReduceBatch rb{};
    int n_pieces = 0, max_tile = 0;
    long long off = 0;
    HeadsLane* lane = nullptr;
    HeadsLaneHold hold;
    if (heads) {
        static const int head_chunks = [] { const char* e = getenv("NIW_DW_HEAD_CHUNKS"); const int v = e ? atoi(e) : kHeadChunks; return v < 1 ? 1 : (v > kHeadChunks ? kHeadChunks : v); }();
        int per = (int)(((mpad + head_chunks - 1) / head_chunks + 127) / 128 * 128);
        per = per < 512 ? 512 : per;
        const int chunks = (int)((mpad + per - 1) / per);
        if (64 * mpad * 16 >= (1ll << 31)) {
            niw_set_error("niw_mlp_bwd_dw: %lld samples exceed the 2 GiB reach of a buffer descriptor", mpad);
            return NIW_ERR_INVALID_ARG;
        }
        HeadsArgs ha{};
        ha.X[0] = save + (long long)save_h(7) * mpad;
        ha.D[0] = gradws + (long long)(kGradY7 + 256) * mpad;
        ha.X[1] = save + (long long)kSaveHr * mpad;
        ha.D[1] = gradws + (long long)kGradRgb1 * mpad;
        ha.partial[0] = partial + off;
        ha.partial[1] = partial + off + (long long)chunks * 512;
        ha.mpad = mpad;
        ha.per = per;
        rb.r[n_pieces++] = ReduceArgs{off, chunks, 256, 1, 7, 256, 0, 1, 1};
        rb.r[n_pieces++] = ReduceArgs{off + (long long)chunks * 512, chunks, 128, 3, 9, 0, 0, 1, 1};
        off += (long long)chunks * (512 + 640);
        max_tile = 384 + 256;
        hipStream_t hs = st;
        if (heads_mode >= 2) {
            lane = heads_lane();
            if (!lane) {
                niw_set_error("niw_mlp_bwd_dw: cannot create the second stream");
                return NIW_ERR_LAUNCH;
            }
            hold.take(lane, st);
            if (hipEventRecord(lane->fork, st) != hipSuccess || hipStreamWaitEvent(lane->s, lane->fork, 0) != hipSuccess) {
                niw_set_error("niw_mlp_bwd_dw: cannot fork the second stream");
                return NIW_ERR_LAUNCH;
            }
            hold.forked = true;
            hs = lane->s;
        }
        dw_heads_kernel<<<chunks, 256, 0, hs>>>(ha);
        NIW_LAUNCH_CHECK("niw_mlp_bwd (dW heads)");
        if (lane && hipEventRecord(lane->join, hs) != hipSuccess) {
            niw_set_error("niw_mlp_bwd_dw: cannot record the join of the second stream");
            return NIW_ERR_LAUNCH;
        }
    }
    // short reduction (one round of the register-chained kernels, <= 32,768 samples): see launch_wide_quadrants.  NIW_DW_QUADRANTS=<max
    // slices> moves the threshold (diagnostic; 0 = never)
    static const long long quad_max = [] { const char* e = getenv("NIW_DW_QUADRANTS"); return e ? atoll(e) : 1024ll; }();
    const bool quadrants = precision == NIW_PREC_FP32 && mpad / 32 <= quad_max;
    for (const Group& g : groups) {
        if (quadrants && g.wide == 1) {
            GemmBatch gb{};
            for (int b = 0; b < g.n; ++b)
                for (int q = 0; q < 4; ++q) {
                    const Piece& p = g.p[b];
                    const int qn = q >> 1, qk = q & 1;
                    gb.A[4 * b + q] = NiwGemmOperand{gradws + (long long)(p.a_row + 128 * qn) * mpad, 128, 0, mpad};
                    gb.B[4 * b + q] = NiwGemmOperand{save + (long long)(p.b_row + 128 * qk) * mpad, 128, 0, mpad};
                    gb.bias_side[4 * b + q] = (p.bias && qk == 0) ? 1 : 0;
                }
            int nsplit = 0;
            if (int rc = launch_wide_quadrants(gb, g.n, mpad, partial + off, &nsplit, st)) return rc;
            const long long tile = 128ll * 128 + 256;
            for (int b = 0; b < g.n; ++b)
                for (int q = 0; q < 4; ++q) {
                    const Piece& p = g.p[b];
                    rb.r[n_pieces++] = ReduceArgs{off + (long long)(4 * b + q) * nsplit * tile, nsplit, 128, 128, p.layer, p.n_off + 128 * (q >> 1),
                                                  p.k_off + 128 * (q & 1), 0, (p.bias && (q & 1) == 0) ? 1 : 0};
                }
            off += 4ll * g.n * nsplit * tile;
            max_tile = (int)tile > max_tile ? (int)tile : max_tile;
            continue;
        }
        GemmBatch gb{};
        for (int b = 0; b < g.n; ++b) {
            const Piece& p = g.p[b];
            // workspaces are quad-row images of the feature-major [row][Mpad] matrix (every operand starts on a multiple of 4 rows)
            // (bf16 mode: bf16 quad-row images, rows half as far apart -- niw_mlp_fast.hip kHalfWorkspace)
            const long long row_floats = precision == NIW_PREC_BF16 ? mpad / 2 : mpad;
            gb.A[b] = NiwGemmOperand{(p.transposed ? save : gradws) + (long long)p.a_row * row_floats, p.a_rows, 0, mpad};
            gb.B[b] = NiwGemmOperand{(p.transposed ? gradws : save) + (long long)p.b_row * row_floats, p.b_rows, 0, mpad};
            gb.bias_side[b] = p.bias ? (p.transposed ? 2 : 1) : 0;
        }
        int nsplit = 0;
        int rc = precision == NIW_PREC_FP32   ? launch_shape<true>(g.wide, gb, mpad, g.n, partial + off, &nsplit, st)
                 : precision == NIW_PREC_BF16X3 ? launch_shape<true, 3>(g.wide, gb, mpad, g.n, partial + off, &nsplit, st)
                                                : launch_shape<true, 1>(g.wide, gb, mpad, g.n, partial + off, &nsplit, st);
        if (rc != NIW_OK) return rc;
        const long long tile = (long long)g.TN * g.TK + 256;
        for (int b = 0; b < g.n; ++b) {
            const Piece& p = g.p[b];
            rb.r[n_pieces++] = ReduceArgs{off + (long long)b * nsplit * tile, nsplit, g.TN, g.TK, p.layer, p.n_off, p.k_off, p.transposed, p.bias};
        }
        off += (long long)g.n * nsplit * tile;
        max_tile = (int)tile > max_tile ? (int)tile : max_tile;
    }
    if (hold.join() != hipSuccess) {
        niw_set_error("niw_mlp_bwd_dw: cannot join the second stream");
        return NIW_ERR_LAUNCH;
    }
    dw_reduce_kernel<<<dim3((max_tile + 255) / 256, n_pieces), 256, 0, st>>>(rb, partial, d_params);
    NIW_LAUNCH_CHECK("niw_mlp_bwd (dW reduce)");
    return NIW_OK;
}

extern "C" int niw_mlp_bwd(const float* packed, const float* center, const float* ray,
                           const float* depth, int64_t n_rays, int n_samples, int density_activ, int precision,
                           const float* rgb, const float* d_rgb, const float* d_sigma,
                           const float* save, float* gradws, float* partial,
                           float* d_params, float* d_center, float* d_ray, niw_stream_t stream) {
    int rc = niw_mlp_bwd_dx(packed, center, ray, depth, n_rays, n_samples, density_activ, precision, rgb, d_rgb, d_sigma, save, gradws,
                            d_center, d_ray, stream);
    if (rc != NIW_OK) return rc;
    return niw_mlp_bwd_dw(save, gradws, n_rays, n_samples, precision, partial, d_params, stream);
}
