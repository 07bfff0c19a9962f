// Field MLP in the opt-in fast-precision modes NIW_PREC_BF16X3 / NIW_PREC_BF16 (niw_mlp_fast.h): the same register-chained design as
// niw_mlp_fwd.hip -- every wave owns 32 samples and carries them through all ten layers, the 32x32 accumulator of one layer being,
// register for register, the B operand of the next -- on v_mfma_f32_32x32x16_bf16 with fp32 accumulation.  Each fp32 activation is split
// into bf16 planes (hi, mid) as it leaves the accumulator: one v_cvt_pk_bf16_f32 per pair and plane, the residue x - hi formed exactly
// in fp32.  Weights are split the same way once per optimizer step by the packing kernel.  Never the default: the headline numbers and
// every parity claim at the reference's fp32 tolerance are the exact-fp32 kernels'; this path has its own, looser, stated tolerance
// (tests/test_gpu_fast_precision.py).
// Reference arithmetic evaluated: model/nerf.py:416-456, 476-483 (+ c2f mask, model/barf_inn_llff.py:427-442).
#include "niw_common.h"
#include "niw_mlp_device.h"
#include "niw_mlp_encode.h"
#include "niw_mlp_fast.h"
#include "niw_bf16.h"

using namespace niw;

namespace {

__device__ __forceinline__ f32x16 mfma_bf16(const u32x4_t& a, const unsigned (&b)[4], f32x16 c) {
    return niw::mfma_bf16(a, u32x4_t{b[0], b[1], b[2], b[3]}, c);
}

// ---------------------------------------------------------------------------------------------------------------------------
// image builder (once per optimizer step): one thread per (chunk, lane) of the fragment sections -- it decodes the chunk once, gathers
// the lane's eight weights and writes their 16 bytes of BOTH planes -- and one thread per float of the bias section.  (One thread per
// dword, each decoding its chunk twice, took 52 us per network: a tenth of a millisecond of a 5.9 ms bf16 step.)
// ---------------------------------------------------------------------------------------------------------------------------
struct FragChunk { int l, blk, q; bool bwd, pad; };
__device__ __forceinline__ FragChunk frag_chunk(int chunk) {
    FragChunk c{0, 0, 0, chunk >= kFastFwdChunks, false};
    if (!c.bwd) {
        while (c.l + 1 < kLayers && chunk >= fast_fwd_chunk(c.l + 1)) ++c.l;
        const int local = chunk - fast_fwd_chunk(c.l);
        c.blk = local / fast_ks(c.l); c.q = local % fast_ks(c.l);
    } else {
        const int cc = chunk - kFastFwdChunks;
        int sgm = 0;
        while (sgm + 1 < kFastBwdSegs && cc >= fast_bwd_chunk(sgm + 1)) ++sgm;
        const FastBwdSeg sg = fast_bwd_seg(sgm);
        const int local = cc - fast_bwd_chunk(sgm);
        c.l = sg.layer;
        c.pad = local >= sg.nob * fast_rs(c.l);                    // padding chunks of the segment
        c.blk = sg.ob0 + local / fast_rs(c.l); c.q = local % fast_rs(c.l);
    }
    return c;
}
// index into the flat parameter vector of element j (0..7) of lane (i, h) of the chunk, or -1 (zero padding)
__device__ __forceinline__ int frag_source(const FragChunk& c, int i, int h, int j) {
    if (c.pad) return -1;
    const int red = 16 * c.q + fast_perm(h, j);                   // reduction index of this element
    const int row = out_row(c.l, c.bwd ? red : c.blk * 32 + i), col = fwd_slot_col(c.l, c.bwd ? c.blk * 32 + i : red);
    return (row >= 0 && col >= 0) ? weight_off(c.l) + row * layer_k(c.l) + col : -1;
}

constexpr int kFastChunks = kFastFwdChunks + kFastBwdChunks;
constexpr int kFastBiasFloats = (kFastImageBytes - kFastChunks * kChunkBytes) / 4;
static_assert(kFastChunks * kChunkBytes == fast_bias_off(0), "the bias section follows the chunk streams");

__global__ void pack_fast_kernel(const float* __restrict__ params, unsigned* __restrict__ image) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t < kFastChunks * 64) {
        const int chunk = t >> 6, lane = t & 63, i = lane & 31, h = lane >> 5;
        const FragChunk c = frag_chunk(chunk);
        u32x4_t hi, mid;
#pragma unroll
        for (int jp = 0; jp < 4; ++jp) {
            const int s0 = frag_source(c, i, h, 2 * jp), s1 = frag_source(c, i, h, 2 * jp + 1);
            unsigned a, b;
            split_pair(s0 >= 0 ? params[s0] : 0.f, s1 >= 0 ? params[s1] : 0.f, a, b);
            hi[jp] = a; mid[jp] = b;
        }
        u32x4_t* dst = reinterpret_cast<u32x4_t*>(reinterpret_cast<char*>(image) + (long long)chunk * kChunkBytes + lane * 16);
        dst[0] = hi;                     // plane 0
        dst[64] = mid;                   // plane 1: 1 KiB further
        return;
    }
    const int f = t - kFastChunks * 64;
    if (f >= kFastBiasFloats) return;
    const int byte = fast_bias_off(0) + 4 * f;
    int l = 0;
    while (l + 1 < kLayers && byte >= fast_bias_off(l + 1)) ++l;
    const int local = (byte - fast_bias_off(l)) / 4;              // [nb][h][16]
    const int r = local & 15, h = (local >> 4) & 1, nb = local >> 5;
    const int row = out_row(l, nb * 32 + acc_row(r, h));
    image[byte / 4] = __builtin_bit_cast(unsigned, row >= 0 ? params[bias_off(l) + row] : 0.f);
}

// ---------------------------------------------------------------------------------------------------------------------------
// Weight stream shared by the four waves of a workgroup.
// In the exact-fp32 kernel every wave fetches its own copy of every weight fragment (16 B / clk / CU from L1 / L2: far from a limit).
// At a sixteenth / three sixteenths of the matrix time the same habit asks 85 - 128 B / clk / CU of a path that delivers ~40
// (measured: the per-wave version of this kernel ran at 46 % / 33 % of its matrix-pipe time).  All four waves walk the SAME chunk
// stream (niw_mlp_fast.h), so a chunk is brought in ONCE per workgroup: stages of kStageChunks chunks, a ring of kRingStages stages
// in LDS, filled by LDS-DMA (buffer_load ... lds: no registers, 1 KiB per wave-instruction), wave w fetching chunks 2w, 2w + 1 of a
// stage kRingStages - 1 stages ahead of its use.  One s_barrier per stage: before the first read of stage t every wave waits for its
// own part of it (s_waitcnt vmcnt: the DMA of the later stages may still be in flight), the barrier makes the other waves' parts
// visible and -- because every wave has by then taken its last fragments of stage t - 1 into registers -- frees that stage's slot for
// the DMA of stage t + kRingStages - 1, issued right behind the barrier.
// ---------------------------------------------------------------------------------------------------------------------------
constexpr int kRingStages = 4;
constexpr int kStageBytes = kStageChunks * kChunkBytes;
constexpr int kFastRingBytes = kRingStages * kStageBytes;             // 64 KiB
// LDS requested per workgroup = the ring.  (NIW_FAST_LDS_RESERVE > 16 KiB pins the kernels to one workgroup per CU: a diagnostic.)
#ifndef NIW_FAST_LDS_RESERVE
#define NIW_FAST_LDS_RESERVE 0
#endif
constexpr int kFastLdsBytes = kFastRingBytes + NIW_FAST_LDS_RESERVE;

template <int TERMS>
struct WeightStream {
    static constexpr int PL = TERMS >= 3 ? 2 : 1;                     // planes this mode multiplies with
    rsrc_t rsrc;
    char* lds;
    int wave, lane16;
    int base_bytes;                                                    // image offset of the stream's chunk 0
    int n_stages;                                                      // stages of the stream

    // DMA of this wave's part of global stage t into ring slot `slot`
    __device__ __forceinline__ void issue(int t, int slot) const {
        // stage index and wave number are wave-uniform by construction; readfirstlane says so to the compiler (an LDS-DMA's M0 and scalar
        // offset built from values it holds in VGPRs become readfirstlane / compare / exec-mask "waterfall" loops, one per instruction)
        const int tu = __builtin_amdgcn_readfirstlane(t), wv = __builtin_amdgcn_readfirstlane(wave);
        if (tu >= n_stages) return;
        const unsigned dst0 = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(__attribute__((address_space(3))) char*)lds) + slot * kStageBytes;
#pragma unroll
        for (int k = 0; k < 2; ++k)
#pragma unroll
            for (int p = 0; p < PL; ++p) {
                const int off = (2 * wv + k) * kChunkBytes + p * 1024;
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (__attribute__((address_space(3))) void*)(size_t)(dst0 + off), 16, lane16,
                                                         base_bytes + tu * kStageBytes + off, 0, 0);
            }
    }
    __device__ __forceinline__ void start() const {
#pragma unroll
        for (int t = 0; t < kRingStages - 1; ++t) issue(t, t);
    }
    // before the first read of global stage t (ring slot `slot`, compile-time)
    template <int SLOT>
    __device__ __forceinline__ void boundary(int t) const {
        // own fragment reads of the previous stage have returned; own DMA of stage t has landed (younger: the 2 * PL * (kRingStages - 2)
        // DMA instructions of the stages behind it -- and whatever else this wave has issued since, which only makes the wait safer)
        if (PL == 2) asm volatile("s_waitcnt vmcnt(8) lgkmcnt(0)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(4) lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        issue(t + kRingStages - 1, (SLOT + kRingStages - 1) % kRingStages);
    }
    template <int SLOT>
    __device__ __forceinline__ u32x4_t frag(int plane, int chunk_in_stage) const {
        return *reinterpret_cast<const u32x4_t*>(lds + SLOT * kStageBytes + chunk_in_stage * kChunkBytes + plane * 1024 + lane16);
    }
};

// ---------------------------------------------------------------------------------------------------------------------------
// streaming register-chained layer on bf16 planes
//   out[n][m] = sum_k A[n][k] B[k][m],  A = the layer's chunks of the weight stream, B = this lane's operand planes b1 then b2
// Row blocks outer, k-steps inner, one dependent MFMA chain per block (a single chain of v_mfma_f32_32x32x16_bf16 issues back to back);
// TERMS = 3: hi*mid + mid*hi + hi*hi per k-step, TERMS = 1: hi*hi.  The fragments of chunk c + 1 are read from LDS before the MFMAs of
// chunk c; the epilogue of block nb-1 -- ReLU, split into planes, stores -- is spread in PAIRS of accumulator registers over the k-steps
// of block nb.  `stage0` = global stage of the layer's first chunk (run time: layers of a kind share one body), PHASE = stage0 modulo
// the ring (compile time: it is the same for every layer that runs a given call site).
//   pol.acc_init(nb, c)            the 16-register value block nb accumulates from (its bias fragment), fetched one block ahead
//   pol.epi2(nb, rp, a0, a1)       accumulator registers 2 rp, 2 rp + 1 of block nb
// ---------------------------------------------------------------------------------------------------------------------------
template <int KS1, int KS2, int NB, int TERMS, int PHASE, typename Policy>
__device__ __forceinline__ void stream_layer_shared(const WeightStream<TERMS>& ws, int stage0, const unsigned (&b1)[2][4 * KS1],
                                                    const unsigned (&b2)[2][4 * (KS2 > 0 ? KS2 : 1)], Policy& pol) {
    constexpr int KS = KS1 + KS2, N = NB * KS, PL = TERMS >= 3 ? 2 : 1, G = kStageChunks, R = kRingStages;
    static_assert(N % G == 0 || true, "");
    f32x16 cin[2], acc[2];
    u32x4_t fr[2][PL];                                      // fragments of the current / next chunk
    pol.acc_init(0, cin[0]);
    ws.template boundary<PHASE % R>(stage0);
#pragma unroll
    for (int p = 0; p < PL; ++p) fr[0][p] = ws.template frag<PHASE % R>(p, 0);
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) {
        f32x16& cur = acc[nb & 1];
        if (nb + 1 < NB) pol.acc_init(nb + 1, cin[(nb + 1) & 1]);
        cur = cin[nb & 1];
#pragma unroll
        for (int q = 0; q < KS; ++q) {
            constexpr int dummy = 0; (void)dummy;
            const int c = nb * KS + q;                      // compile-time after unrolling
            // next chunk's fragments (crossing into the next stage: synchronise first)
            if (c + 1 < N) {
                if ((c + 1) % G == 0) {
                    switch (((c + 1) / G + PHASE) % R) {     // compile-time slot of the stage being entered
                        case 0: ws.template boundary<0>(stage0 + (c + 1) / G); break;
                        case 1: ws.template boundary<1>(stage0 + (c + 1) / G); break;
                        case 2: ws.template boundary<2>(stage0 + (c + 1) / G); break;
                        default: ws.template boundary<3>(stage0 + (c + 1) / G); break;
                    }
                }
#pragma unroll
                for (int p = 0; p < PL; ++p) {
                    switch (((c + 1) / G + PHASE) % R) {
                        case 0: fr[(c + 1) & 1][p] = ws.template frag<0>(p, (c + 1) % G); break;
                        case 1: fr[(c + 1) & 1][p] = ws.template frag<1>(p, (c + 1) % G); break;
                        case 2: fr[(c + 1) & 1][p] = ws.template frag<2>(p, (c + 1) % G); break;
                        default: fr[(c + 1) & 1][p] = ws.template frag<3>(p, (c + 1) % G); break;
                    }
                }
            }
            unsigned bh[4], bm[4];
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                bh[t] = q < KS1 ? b1[0][4 * (q < KS1 ? q : 0) + t] : b2[0][4 * (q >= KS1 ? q - KS1 : 0) + t];
                bm[t] = q < KS1 ? b1[1][4 * (q < KS1 ? q : 0) + t] : b2[1][4 * (q >= KS1 ? q - KS1 : 0) + t];
            }
            if (TERMS >= 3) {
                cur = mfma_bf16(fr[c & 1][0], bm, cur);          // the two small terms first, the leading term last
                cur = mfma_bf16(fr[c & 1][PL - 1], bh, cur);
            }
            cur = mfma_bf16(fr[c & 1][0], bh, cur);
            if (nb > 0) {
#pragma unroll
                for (int rp = q * 8 / KS; rp < (q + 1) * 8 / KS; ++rp) pol.epi2(nb - 1, rp, acc[(nb - 1) & 1][2 * rp], acc[(nb - 1) & 1][2 * rp + 1]);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
#pragma unroll
    for (int rp = 0; rp < 8; ++rp) pol.epi2(NB - 1, rp, acc[(NB - 1) & 1][2 * rp], acc[(NB - 1) & 1][2 * rp + 1]);
}

// The same layer with every wave fetching its own copy of every chunk through a register ring (kFastRing chunks in flight), as the
// exact-fp32 kernel does.  Which of the two forms a mode uses is decided by measurement (kSharedWeights below).
#ifndef NIW_FAST_RING
#define NIW_FAST_RING 8
#endif
#ifndef NIW_FAST_SCHED
#define NIW_FAST_SCHED 1
#endif
constexpr int kFastRing = NIW_FAST_RING;
template <int KS1, int KS2, int NB, int TERMS, typename Policy>
__device__ __forceinline__ void stream_layer_ring(const WeightStream<TERMS>& ws, int stage0, const unsigned (&b1)[2][4 * KS1],
                                                  const unsigned (&b2)[2][4 * (KS2 > 0 ? KS2 : 1)], Policy& pol) {
    constexpr int KS = KS1 + KS2, N = NB * KS, D = kFastRing, PL = TERMS >= 3 ? 2 : 1;
    const int w_base = ws.base_bytes + stage0 * kStageBytes;
    u32x4_t ring[D][PL];
    auto fetch = [&](int i, int slot) {
#pragma unroll
        for (int p = 0; p < PL; ++p)
            ring[slot][p] = __builtin_bit_cast(u32x4_t, __builtin_amdgcn_raw_buffer_load_b128(ws.rsrc, ws.lane16, w_base + i * kChunkBytes + p * 1024, 0));
    };
    f32x16 cin[2], acc[2];
    pol.acc_init(0, cin[0]);
#pragma unroll
    for (int i = 0; i < D; ++i)
        if (i < N) fetch(i, i);
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) {
        f32x16& cur = acc[nb & 1];
        if (nb + 1 < NB) pol.acc_init(nb + 1, cin[(nb + 1) & 1]);
        cur = cin[nb & 1];
#pragma unroll
        for (int q = 0; q < KS; ++q) {
            const int i = nb * KS + q;
            u32x4_t a[PL];
#pragma unroll
            for (int p = 0; p < PL; ++p) a[p] = ring[i % D][p];
            if (i + D < N) fetch(i + D, i % D);
            unsigned bh[4], bm[4];
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                bh[t] = q < KS1 ? b1[0][4 * (q < KS1 ? q : 0) + t] : b2[0][4 * (q >= KS1 ? q - KS1 : 0) + t];
                bm[t] = q < KS1 ? b1[1][4 * (q < KS1 ? q : 0) + t] : b2[1][4 * (q >= KS1 ? q - KS1 : 0) + t];
            }
            if (TERMS >= 3) {
                cur = mfma_bf16(a[0], bm, cur);                  // the two small terms first, the leading term last
                cur = mfma_bf16(a[PL - 1], bh, cur);
            }
            cur = mfma_bf16(a[0], bh, cur);
            if (nb > 0) {
#pragma unroll
                for (int rp = q * 8 / KS; rp < (q + 1) * 8 / KS; ++rp) pol.epi2(nb - 1, rp, acc[(nb - 1) & 1][2 * rp], acc[(nb - 1) & 1][2 * rp + 1]);
            }
            if (NIW_FAST_SCHED) __builtin_amdgcn_sched_barrier(0);
        }
    }
#pragma unroll
    for (int rp = 0; rp < 8; ++rp) pol.epi2(NB - 1, rp, acc[(NB - 1) & 1][2 * rp], acc[(NB - 1) & 1][2 * rp + 1]);
}

// Measured on MI355X, 784,512 samples, us (eval forward / training forward / dX chain):
//   bf16x3   per-wave ring 2196 / 2634 / 2697     shared through LDS 6875 (spills) / 2889 / 2609
//   bf16     per-wave ring 1073 / 1617 / 2104     shared through LDS 1198 / 1495 / 1398
// Sharing removes three quarters of the L2 -> CU weight traffic and pays one barrier per 8 chunks.  With three matrix instructions per
// chunk the per-wave ring keeps up (and the shared form's barrier, fragment reads and waits cost as much as they save); with one
// instruction per chunk the ring's 128 B / clk / CU does not, and sharing wins where the kernel also streams stores.
template <int TERMS>
#ifndef NIW_FAST_SHARED
#define NIW_FAST_SHARED 1
#endif
constexpr bool kSharedWeights = TERMS == 1 && NIW_FAST_SHARED;

template <int KS1, int KS2, int NB, int TERMS, int PHASE, typename Policy>
__device__ __forceinline__ void stream_layer_bf(const WeightStream<TERMS>& ws, int stage0, const unsigned (&b1)[2][4 * KS1],
                                                const unsigned (&b2)[2][4 * (KS2 > 0 ? KS2 : 1)], Policy& pol) {
    if constexpr (kSharedWeights<TERMS>) stream_layer_shared<KS1, KS2, NB, TERMS, PHASE>(ws, stage0, b1, b2, pol);
    else stream_layer_ring<KS1, KS2, NB, TERMS>(ws, stage0, b1, b2, pol);
}

// planes of an fp32 register array in B-operand order: dword r of a plane = values 2 r, 2 r + 1
template <int NV>
__device__ __forceinline__ void split_regs(const float (&v)[NV], unsigned (&planes)[2][NV / 2]) {
#pragma unroll
    for (int r = 0; r < NV / 2; ++r) split_pair(v[2 * r], v[2 * r + 1], planes[0][r], planes[1][r]);
}

// bf16 workspaces (round 3).  In the one-term mode (NIW_PREC_BF16) the dW GEMMs multiply the LEADING bf16 plane of every saved activation
// and gradient and nothing else -- so that plane is what the forward and the dX chain store: the quad-row images of `save` / `gradws`
// hold four bf16 per (quad, sample), 8 bytes where the fp32 image has 16 (same quad numbering, half the pitch: quad q of sample m at
// (q * Mpad + m) * 8 bytes; the packed pairs are the very dwords the next layer consumes as its operand plane).  Rows that are not MFMA
// operands keep their fp32 places: the raw-density row and the mask records of `save`, the stash rows of `gradws` -- they lie behind
// the (now half as long) quad regions.  The kernels of the mode agree on this among themselves; the caller's buffers are unchanged
// (sized for fp32).  It halves the bytes all three kernels of the mode are bound by.  bf16x3 keeps fp32 workspaces (two planes = the
// same bytes).  Parameter gradients are bit-identical to the fp32-workspace form of the mode (the dW pass rounded to the same planes
// on its way into LDS); the ray gradients see the saved encodings rounded to bf16.
template <int TERMS>
constexpr bool kHalfWorkspace = TERMS == 1;

// Forward epilogue: the bias is in the accumulator already (acc_init); ReLU; planes of the next layer's operand; training: the fp32
// activation into the quad-row workspace and the ReLU sign bits into the wave's mask record -- byte for byte what the exact-fp32
// forward leaves behind, so either backward can consume it.
// KIND 0: hidden layer; 1: layer 7 (row block 8, register 0 of lane half 0 = raw density); 2: colour output (no ReLU, no planes).
template <int NBOUT, bool SAVE, int KIND, bool HALF = false>
struct FastFwdEpilogue {
    const PackedWeights& pw;
    int bias_bytes, hoff;                // packed bias of the layer ([row block][half][16] floats); h * 64
    unsigned (&out)[2][8 * NBOUT];
    RowWindow win;
    const char* mrec;
    int lane;
    unsigned mbits[4] = {0u, 0u, 0u, 0u};
    float keep[2] = {0.f, 0.f};          // registers 4g, 4g+1 waiting for 4g+2, 4g+3 (one 16-byte store per quad)
    float head[4] = {0.f, 0.f, 0.f, 0.f};   // KIND 1: [0] = raw density; KIND 2: the three colour logits

    __device__ __forceinline__ void acc_init(int nb, f32x16& c) const {
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const f32x4 v = buf_load4(pw.rsrc, hoff, bias_bytes + nb * 128 + g * 16);
            c[4 * g] = v[0]; c[4 * g + 1] = v[1]; c[4 * g + 2] = v[2]; c[4 * g + 3] = v[3];
        }
    }
    __device__ __forceinline__ void epi2(int nb, int rp, float a0, float a1) {
        if (KIND == 2) {                 // colour logits: rows 0..2 of the single row block = registers 0..2 of lane half 0
            if (rp == 0) { head[0] = a0; head[1] = a1; }
            if (rp == 1) head[2] = a0;
            return;
        }
        if (KIND == 1 && nb == NBOUT) {  // the density row block: row 0 only
            if (rp == 0) head[0] = a0;
            return;
        }
        const float v0 = __builtin_bit_cast(float, max(__builtin_bit_cast(int, a0), 0));
        const float v1 = __builtin_bit_cast(float, max(__builtin_bit_cast(int, a1), 0));
        split_pair(v0, v1, out[0][nb * 8 + rp], out[1][nb * 8 + rp]);
        if (SAVE) {
            if (HALF) {                  // the operand plane itself: rows 4q .. 4q+3 = the pairs 2q, 2q+1
                if (rp & 1) buf_store2(out[0][nb * 8 + rp - 1], out[0][nb * 8 + rp], win.rsrc(nb * 32 + 8 * (rp >> 1)), win.voff4, 0);
            } else if (rp & 1) {
                buf_store4(keep[0], keep[1], v0, v1, win.rsrc(nb * 32 + 8 * (rp >> 1)), win.voff4, 0);
            } else {
                keep[0] = v0; keep[1] = v1;
            }
            mbits[nb >> 1] = __builtin_amdgcn_alignbit(mbits[nb >> 1], __builtin_bit_cast(unsigned, v0) + 0x7fffffffu, 31);
            mbits[nb >> 1] = __builtin_amdgcn_alignbit(mbits[nb >> 1], __builtin_bit_cast(unsigned, v1) + 0x7fffffffu, 31);
            if (nb == NBOUT - 1 && rp == 7)
                __builtin_amdgcn_raw_buffer_store_b128(u32x4_t{mbits[0], mbits[1], mbits[2], mbits[3]}, make_rsrc(mrec), lane * 16, 0, 0);
        }
    }
};

struct FastFwdArgs {
    const unsigned char* image;
    const float* center;
    const float* ray;
    const float* depth;
    const float* noise;
    float* rgb;
    float* sigma;
    float* save;
    long long M, Mpad;
    int S, act;
    float w3d[NIW_L3D];
    float wview[NIW_LVIEW];
    const float* band_dev;
};

__device__ __forceinline__ float density_act_fast(float x, int kind) {
    if (kind == NIW_ACT_RELU) return fmaxf(x, 0.f);
    return x > 20.f ? x : log1pf(expf(x));      // F.softplus(beta=1, threshold=20)
}

static_assert((8 * 16 / kStageChunks) % kRingStages == 0, "a 256 -> 256 layer is a whole number of ring turns: the layers of one rolled loop share their ring phase");

// Store-data hazard (round 3, tools/store_war_hazard.hip, HISTORY.md).  The bf16 (one-term) kernels need fewer than 256
// registers, so two of their workgroups share a CU -- and on gfx950 a 16-byte buffer store with an SGPR soffset that is followed AT ONCE by
// a vector write of its data registers stores the new value when waves share a SIMD.  LLVM inserts the wait state only for stores whose
// soffset is an immediate, and hipcc did schedule that pair in these epilogues: every launch of more than 256 workgroups stored corrupted
// activations / gradients from its second workgroups (whole 4-lane groups carrying NaN or the next epilogue step's values) while the
// registers the next layer consumes -- rgb, sigma, the loss of that step -- were right.  All epilogue stores of this file therefore give
// the row offset to the DESCRIPTOR (scalar ALU) and use soffset 0: the compiler then sees the hazard and pads it.
// tools/check_store_hazard.py (a CPU test) scans the assembly of every kernel for the pair; test_fast_precision_beyond_one_round runs both
// modes beyond one round of workgroups.
template <int TERMS, bool SAVE>
__global__ __launch_bounds__(256, 1) void mlp_fwd_fast_kernel(FastFwdArgs a) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int j = lane & 31, h = lane >> 5;
    const long long m = ((long long)blockIdx.x * 4 + wave) * 32 + j;
    const bool valid = m < a.M;
    const long long mc = valid ? m : a.M - 1;
    const long long ri = mc / a.S;
    // ---- sample point and unit view direction, positional encodings: fp32, exactly as the exact-fp32 forward forms them
    float p[3], u[3];
    {
        const float d = a.depth[mc];
        const float rx = a.ray[ri * 3 + 0], ry = a.ray[ri * 3 + 1], rz = a.ray[ri * 3 + 2];
        p[0] = add_rn(a.center[ri * 3 + 0], mul_rn(rx, d));
        p[1] = add_rn(a.center[ri * 3 + 1], mul_rn(ry, d));
        p[2] = add_rn(a.center[ri * 3 + 2], mul_rn(rz, d));
        const float nrm = fmaxf(sqrtf(rx * rx + ry * ry + rz * rz), 1e-12f);
        u[0] = rx / nrm; u[1] = ry / nrm; u[2] = rz / nrm;
    }
    float enc[32], venc[16];
    {
        float w3[NIW_L3D], wv[NIW_LVIEW];
#pragma unroll
        for (int i = 0; i < NIW_L3D; ++i) w3[i] = a.band_dev ? a.band_dev[i] : a.w3d[i];
#pragma unroll
        for (int i = 0; i < NIW_LVIEW; ++i) wv[i] = a.band_dev ? a.band_dev[NIW_L3D + i] : a.wview[i];
        encode_slots<NIW_L3D, 8>(p, w3, h, enc);
        encode_slots<NIW_LVIEW, 4>(u, wv, h, venc);
    }
    const PackedWeights pw{make_rsrc(a.image), reinterpret_cast<const float*>(a.image), lane * 16};
    extern __shared__ __attribute__((aligned(16))) char wlds[];
    const WeightStream<TERMS> ws{pw.rsrc, wlds, __builtin_amdgcn_readfirstlane(wave), lane * 16, 0, kFastFwdChunks / kStageChunks};
    if (kSharedWeights<TERMS>) ws.start();
    constexpr int G = kStageChunks;
    constexpr bool HALF = kHalfWorkspace<TERMS>;          // bf16 quad-row images (see above): half the pitch, 8 bytes per (quad, sample)
    const int pitch4 = (int)(a.Mpad * (HALF ? 2 : 4)), voff4 = (int)(((long long)h * a.Mpad + m) * (HALF ? 8 : 16)), hoff = h * 64;
    const unsigned mpad32 = (unsigned)a.Mpad;
    auto row_off = [&](int r) { return (long long)((unsigned long long)(unsigned)r * (unsigned long long)mpad32); };       // floats (fp32 places)
    auto window = [&](int r) {                             // quad rows of the activation image
        return RowWindow{SAVE ? reinterpret_cast<const float*>(reinterpret_cast<const char*>(a.save) + row_off(r) * (HALF ? 2 : 4)) : nullptr, pitch4, voff4};
    };
    const long long wave_id = __builtin_amdgcn_readfirstlane((int)(blockIdx.x * 4 + wave));
    auto mask_rec = [&](int i) {
        return SAVE ? reinterpret_cast<const char*>(a.save + row_off(kSaveMask)) + (wave_id * kMaskRecords + i) * kMaskRecBytes : nullptr;
    };
    if (SAVE) {
        const RowWindow we = window(kSaveEnc), wv = window(kSaveVenc);
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            if (HALF) buf_store2(pack_bf16(enc[4 * q], enc[4 * q + 1]), pack_bf16(enc[4 * q + 2], enc[4 * q + 3]), we.rsrc(8 * q), voff4, 0);
            else buf_store4(enc[4 * q], enc[4 * q + 1], enc[4 * q + 2], enc[4 * q + 3], we.rsrc(8 * q), voff4, 0);
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            if (HALF) buf_store2(pack_bf16(venc[4 * q], venc[4 * q + 1]), pack_bf16(venc[4 * q + 2], venc[4 * q + 3]), wv.rsrc(8 * q), voff4, 0);
            else buf_store4(venc[4 * q], venc[4 * q + 1], venc[4 * q + 2], venc[4 * q + 3], wv.rsrc(8 * q), voff4, 0);
        }
    }
    unsigned encp[2][16], vencp[2][8];
    split_regs<32>(enc, encp);
    split_regs<16>(venc, vencp);
    const unsigned none[2][4] = {{0u, 0u, 0u, 0u}, {0u, 0u, 0u, 0u}};
    unsigned act[2][64], nxt[2][64];
    auto advance = [&]() {
#pragma unroll
        for (int pl = 0; pl < 2; ++pl)
#pragma unroll
            for (int i = 0; i < 64; ++i) act[pl][i] = nxt[pl][i];
    };
    // ---- layer 0: 63 -> 256
    {
        FastFwdEpilogue<8, SAVE, 0, HALF> ep{pw, fast_bias_off(0), hoff, nxt, window(save_h(1)), mask_rec(0), lane};
        stream_layer_bf<4, 0, 8, TERMS, (fast_fwd_chunk(0) / G) % kRingStages>(ws, fast_fwd_chunk(0) / G, encp, none, ep);
        advance();
    }
    // ---- layers 1..3
#pragma unroll 1
    for (int l = 1; l <= 3; ++l) {
        FastFwdEpilogue<8, SAVE, 0, HALF> ep{pw, fast_bias_off(1) + (l - 1) * 8 * 128, hoff, nxt, window(save_h(l + 1)), mask_rec(l), lane};
        stream_layer_bf<16, 0, 8, TERMS, (fast_fwd_chunk(1) / G) % kRingStages>(ws, fast_fwd_chunk(1) / G + (l - 1) * (8 * 16 / G), act, none, ep);
        advance();
    }
    // ---- layer 4: cat[feat, points_enc] (319) -> 256
    {
        FastFwdEpilogue<8, SAVE, 0, HALF> ep{pw, fast_bias_off(4), hoff, nxt, window(save_h(5)), mask_rec(4), lane};
        stream_layer_bf<16, 4, 8, TERMS, (fast_fwd_chunk(4) / G) % kRingStages>(ws, fast_fwd_chunk(4) / G, act, encp, ep);
        advance();
    }
    // ---- layers 5, 6
#pragma unroll 1
    for (int l = 5; l <= 6; ++l) {
        FastFwdEpilogue<8, SAVE, 0, HALF> ep{pw, fast_bias_off(5) + (l - 5) * 8 * 128, hoff, nxt, window(save_h(l + 1)), mask_rec(l), lane};
        stream_layer_bf<16, 0, 8, TERMS, (fast_fwd_chunk(5) / G) % kRingStages>(ws, fast_fwd_chunk(5) / G + (l - 5) * (8 * 16 / G), act, none, ep);
        advance();
    }
    // ---- layer 7: 256 -> 256 features + the density row (row block 8)
    {
        FastFwdEpilogue<8, SAVE, 1, HALF> ep{pw, fast_bias_off(7), hoff, nxt, window(kSaveFeat), mask_rec(7), lane};
        stream_layer_bf<16, 0, 9, TERMS, (fast_fwd_chunk(7) / G) % kRingStages>(ws, fast_fwd_chunk(7) / G, act, none, ep);
        advance();
        float sig_raw = ep.head[0];
        if (a.noise != nullptr) sig_raw += a.noise[mc];
        if (h == 0) {
            if (SAVE) (a.save + row_off(kSaveSigma))[m] = sig_raw;
            if (valid) a.sigma[m] = density_act_fast(sig_raw, a.act);
        }
    }
    // ---- colour layer 0: cat[feat, view_enc] (283) -> 128
    unsigned hr[2][32];
    {
        FastFwdEpilogue<4, SAVE, 0, HALF> ep{pw, fast_bias_off(8), hoff, hr, window(kSaveHr), mask_rec(8), lane};
        stream_layer_bf<16, 2, 4, TERMS, (fast_fwd_chunk(8) / G) % kRingStages>(ws, fast_fwd_chunk(8) / G, act, vencp, ep);
    }
    // ---- colour layer 1: 128 -> 3, sigmoid
    {
        unsigned unused[2][8];
        FastFwdEpilogue<1, false, 2> ep{pw, fast_bias_off(9), hoff, unused, RowWindow{nullptr, 0, 0}, nullptr, lane};
        stream_layer_bf<8, 0, 1, TERMS, (fast_fwd_chunk(9) / G) % kRingStages>(ws, fast_fwd_chunk(9) / G, hr, none, ep);
        if (h == 0 && valid) {
#pragma unroll
            for (int c = 0; c < 3; ++c) a.rgb[m * 3 + c] = 1.f / (1.f + expf(-ep.head[c]));
        }
    }
}

}  // namespace

extern "C" int64_t niw_mlp_fast_image_bytes(void) { return kFastImageBytes; }

extern "C" int niw_mlp_pack_weights_fast(const float* params, void* image, niw_stream_t stream) {
    NIW_REQUIRE(params && image, "niw_mlp_pack_weights_fast: null pointer");
    pack_fast_kernel<<<(kFastChunks * 64 + kFastBiasFloats + 255) / 256, 256, 0, (hipStream_t)stream>>>(params, reinterpret_cast<unsigned*>(image));
    NIW_LAUNCH_CHECK("niw_mlp_pack_weights_fast");
    return NIW_OK;
}

int niw_launch_mlp_fwd_fast(int precision, const void* image, const float* center, const float* ray, const float* depth, const float* noise,
                            int64_t n_rays, int n_samples, const float* band_w3d, const float* band_wview, const float* band_dev,
                            int density_activ, float* rgb, float* sigma, float* save, hipStream_t stream) {
    FastFwdArgs a;
    a.image = reinterpret_cast<const unsigned char*>(image);
    a.center = center; a.ray = ray; a.depth = depth; a.noise = noise; a.rgb = rgb; a.sigma = sigma; a.save = save;
    a.M = n_rays * (int64_t)n_samples; a.Mpad = niw_mlp_padded_rows(n_rays, n_samples);
    a.S = n_samples; a.act = density_activ;
    for (int i = 0; i < NIW_L3D; ++i) a.w3d[i] = band_w3d ? band_w3d[i] : 1.f;
    for (int i = 0; i < NIW_LVIEW; ++i) a.wview[i] = band_wview ? band_wview[i] : 1.f;
    a.band_dev = band_dev;
    const int blocks = (int)(a.Mpad / 128);
#define NIW_FAST_FWD(T, S)                                                                                                  \
    do {                                                                                                                    \
        static std::atomic<unsigned long long> attr{0ull};                                                                  \
        if (int rc = niw_ensure_dynamic_lds(reinterpret_cast<const void*>(mlp_fwd_fast_kernel<T, S>), kFastLdsBytes, attr, "niw_mlp_fwd (fast)")) return rc; \
        mlp_fwd_fast_kernel<T, S><<<blocks, 256, kFastLdsBytes, stream>>>(a);                                               \
    } while (0)
    if (precision == NIW_PREC_BF16X3) {
        if (save) NIW_FAST_FWD(3, true); else NIW_FAST_FWD(3, false);
    } else {
        if (save) NIW_FAST_FWD(1, true); else NIW_FAST_FWD(1, false);
    }
#undef NIW_FAST_FWD
    NIW_LAUNCH_CHECK("niw_mlp_fwd (fast precision)");
    return NIW_OK;
}

// ---------------------------------------------------------------------------------------------------------------------------
// dX chain in the fast modes: dX[k][m] = sum_n W[n][k] dY[n][m] with the transposed fragments of the image (backward section), the
// ReLU masks the forward recorded, every dY stored in fp32 for the dW pass -- the same workspace contents as the exact chain
// (niw_mlp_bwd.hip), formed on split-bf16 operands.
// ---------------------------------------------------------------------------------------------------------------------------
namespace {

// sign bit record of the forward -> all ones / zero -> one AND: v_bfe_i32 + v_and_b32.  `zero` is a run-time 0 (kernel argument) that
// keeps the bit position from being a compile-time constant -- with one LLVM rewrites the pair as test-bit / compare / select, three
// instructions and the compare's wait states (niw_mlp_bwd.hip MaskEpilogue).
__device__ __forceinline__ float relu_keep(const u32x4_t& mk, int nb, int r, float a, int zero) {
    const int keep = __builtin_amdgcn_sbfe((int)mk[nb >> 1], 31 - (16 * (nb & 1) + r) + zero, 1);
    return __builtin_bit_cast(float, __builtin_bit_cast(int, a) & keep);
}
// mask, planes for the next product, fp32 store of dY for the dW pass
template <int NBOUT, bool HALF = false>
struct FastMaskEpilogue {
    u32x4_t mk;
    int zero;
    unsigned (&out)[2][8 * NBOUT];
    RowWindow grad;
    float keep[2] = {0.f, 0.f};
    __device__ __forceinline__ void acc_init(int, f32x16& c) const {
#pragma unroll
        for (int r = 0; r < 16; ++r) c[r] = 0.f;
    }
    __device__ __forceinline__ void epi2(int nb, int rp, float a0, float a1) {
        const float g0 = relu_keep(mk, nb, 2 * rp, a0, zero), g1 = relu_keep(mk, nb, 2 * rp + 1, a1, zero);
        split_pair(g0, g1, out[0][nb * 8 + rp], out[1][nb * 8 + rp]);
        if (HALF) {                      // bf16 workspace: the operand plane itself (see kHalfWorkspace)
            if (rp & 1) buf_store2(out[0][nb * 8 + rp - 1], out[0][nb * 8 + rp], grad.rsrc(nb * 32 + 8 * (rp >> 1)), grad.voff4, 0);
        } else if (rp & 1) {
            buf_store4(keep[0], keep[1], g0, g1, grad.rsrc(nb * 32 + 8 * (rp >> 1)), grad.voff4, 0);
        } else {
            keep[0] = g0; keep[1] = g1;
        }
    }
};
// park a result in the stash rows (d encoding slots of the skip connection, d view-encoding slots)
struct FastStashEpilogue {
    RowWindow win;
    bool store;
    float keep[2] = {0.f, 0.f};
    __device__ __forceinline__ void acc_init(int, f32x16& c) const {
#pragma unroll
        for (int r = 0; r < 16; ++r) c[r] = 0.f;
    }
    __device__ __forceinline__ void epi2(int nb, int rp, float a0, float a1) {
        if (rp & 1) { if (store) buf_store4(keep[0], keep[1], a0, a1, win.rsrc(nb * 32 + 8 * (rp >> 1)), win.voff4, 0); }
        else { keep[0] = a0; keep[1] = a1; }
    }
};
// the parked part is the value the accumulation STARTS from; the sum stays in registers
template <int NBOUT>
struct FastAddStashEpilogue {
    RowWindow win;
    float (&out)[16 * NBOUT];
    __device__ __forceinline__ void acc_init(int nb, f32x16& c) const {
        const rsrc_t r0 = win.rsrc(nb * 32);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const f32x4 v = buf_load4(r0, win.voff4, 8 * q * win.pitch4);
            c[4 * q] = v[0]; c[4 * q + 1] = v[1]; c[4 * q + 2] = v[2]; c[4 * q + 3] = v[3];
        }
    }
    __device__ __forceinline__ void epi2(int nb, int rp, float a0, float a1) { out[nb * 16 + 2 * rp] = a0; out[nb * 16 + 2 * rp + 1] = a1; }
};

struct FastBwdArgs {
    const unsigned char* image;
    const float* center;
    const float* ray;
    const float* depth;
    const float* rgb;
    const float* d_rgb;
    const float* d_sigma;
    const float* save;
    float* grad;
    long long M, Mpad;
    int S, act, ray_grad;
    int zero;          // always 0 (relu_keep)
};

template <int TERMS>
__global__ __launch_bounds__(256, 1) void mlp_bwd_dx_fast_kernel(FastBwdArgs a) {

    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int j = lane & 31, h = lane >> 5;
    const long long m = ((long long)blockIdx.x * 4 + wave) * 32 + j;
    const bool valid = m < a.M;
    const long long mc = valid ? m : a.M - 1;
    const unsigned qoff = (unsigned)((long long)h * a.Mpad + m);
    const long long P = a.Mpad;
    const PackedWeights pw{make_rsrc(a.image), reinterpret_cast<const float*>(a.image), lane * 16};
    extern __shared__ __attribute__((aligned(16))) char wlds[];
    const WeightStream<TERMS> ws{pw.rsrc, wlds, __builtin_amdgcn_readfirstlane(wave), lane * 16, kFastBwdOffBytes, kFastBwdChunks / kStageChunks};
    if (kSharedWeights<TERMS>) ws.start();
    constexpr int G = kStageChunks;
    constexpr bool HALF = kHalfWorkspace<TERMS>;          // bf16 quad-row images for everything the dW pass multiplies (see kHalfWorkspace)
    const int pitch4 = (int)(P * 4), voff4 = (int)(((long long)h * P + m) * 16);
    auto swin = [&](int r) { return RowWindow{a.grad + (long long)r * P, pitch4, voff4}; };            // stash rows: fp32, at their fp32 places
    auto gwin = [&](int r) {                                                                            // dY rows
        return HALF ? RowWindow{reinterpret_cast<const float*>(reinterpret_cast<const char*>(a.grad) + (long long)r * P * 2), (int)(P * 2), (int)(((long long)h * P + m) * 8)}
                    : swin(r);
    };
    // a quad of four values of this lane's sample at quad row `r` of the gradient image (rows 4h + t of an 8-row slot block)
    auto store_quad = [&](int r, float g0, float g1, float g2, float g3) {
        if (HALF) {
            typedef unsigned u32x2_t __attribute__((ext_vector_type(2)));
            reinterpret_cast<u32x2_t*>(reinterpret_cast<char*>(a.grad) + (long long)r * P * 2)[qoff] = u32x2_t{pack_bf16(g0, g1), pack_bf16(g2, g3)};
        } else {
            reinterpret_cast<f32x4*>(a.grad + (long long)r * P)[qoff] = f32x4{g0, g1, g2, g3};
        }
    };
    const long long wave_id = __builtin_amdgcn_readfirstlane((int)(blockIdx.x * 4 + wave));
    const char* mbase = reinterpret_cast<const char*>(a.save + (long long)kSaveMask * P) + wave_id * kMaskRecords * kMaskRecBytes;
    auto mask_rec = [&](int i) {
        const u32x4_t v = __builtin_bit_cast(u32x4_t, __builtin_amdgcn_raw_buffer_load_b128(make_rsrc(mbase), lane * 16, i * kMaskRecBytes, 0));
        return valid ? v : u32x4_t{0u, 0u, 0u, 0u};
    };
    u32x4_t mk_cur = mask_rec(8), mk_nxt = mask_rec(7);
    const unsigned none[2][4] = {{0u, 0u, 0u, 0u}, {0u, 0u, 0u, 0u}};
    unsigned dy[2][64], nxt[2][64];
    auto advance = [&]() {
#pragma unroll
        for (int pl = 0; pl < 2; ++pl)
#pragma unroll
            for (int i = 0; i < 64; ++i) dy[pl][i] = nxt[pl][i];
    };
    // ---- colour head: sigmoid' ; rows 0..2 of the reduction step = elements 0..2 of lane half 0
    unsigned dy9p[2][4] = {{0u, 0u, 0u, 0u}, {0u, 0u, 0u, 0u}};
    {
        float g[4] = {0.f, 0.f, 0.f, 0.f};
        if (h == 0 && valid) {
#pragma unroll
            for (int t = 0; t < 3; ++t) {
                const float o = a.rgb[mc * 3 + t];
                g[t] = a.d_rgb[mc * 3 + t] * o * (1.f - o);
            }
        }
        store_quad(kGradRgb1, g[0], g[1], g[2], g[3]);
        split_pair(g[0], g[1], dy9p[0][0], dy9p[1][0]);
        split_pair(g[2], g[3], dy9p[0][1], dy9p[1][1]);
    }
    unsigned dyr[2][32];
    {
        FastMaskEpilogue<4, HALF> ep{mk_cur, a.zero, dyr, gwin(kGradRgb0)};
        stream_layer_bf<1, 0, 4, TERMS, (fast_bwd_chunk(0) / G) % kRingStages>(ws, fast_bwd_chunk(0) / G, dy9p, none, ep);
    }
    // ---- colour layer 0 transposed: 128 -> 256 features (+ 32 view-encoding slots = slot block 8 of 9)
    {   // (always run: the weight stream is linear; without a ray gradient only its stores are dropped)
        FastStashEpilogue ep{swin(kGradStashVenc), a.ray_grad != 0};
        stream_layer_bf<8, 0, 1, TERMS, (fast_bwd_chunk(1) / G) % kRingStages>(ws, fast_bwd_chunk(1) / G, dyr, none, ep);
    }
    {
        mk_cur = mk_nxt; mk_nxt = mask_rec(6);
        FastMaskEpilogue<8, HALF> ep{mk_cur, a.zero, dy, gwin(kGradY7)};
        stream_layer_bf<8, 0, 8, TERMS, (fast_bwd_chunk(2) / G) % kRingStages>(ws, fast_bwd_chunk(2) / G, dyr, none, ep);
    }
    // ---- density head: d sigma_raw = reduction row 256 of layer 7 = element 0 of lane half 0 of the 17th step
    unsigned dsigp[2][4] = {{0u, 0u, 0u, 0u}, {0u, 0u, 0u, 0u}};
    {
        float g = 0.f;
        if (h == 0 && valid) {
            const float raw = (a.save + (long long)kSaveSigma * P)[m];
            const float dact = a.act == NIW_ACT_RELU ? (raw > 0.f ? 1.f : 0.f) : (raw > 20.f ? 1.f : 1.f / (1.f + expf(-raw)));
            g = a.d_sigma[mc] * dact;
        }
        store_quad(kGradY7 + 256, g, 0.f, 0.f, 0.f);
        split_pair(g, 0.f, dsigp[0][0], dsigp[1][0]);
    }
    // ---- layer 7 transposed (257 -> 256), mask with h7
    {
        mk_cur = mk_nxt; mk_nxt = mask_rec(5);
        FastMaskEpilogue<8, HALF> ep{mk_cur, a.zero, nxt, gwin(6 * 256)};
        stream_layer_bf<16, 1, 8, TERMS, (fast_bwd_chunk(3) / G) % kRingStages>(ws, fast_bwd_chunk(3) / G, dy, dsigp, ep);
        advance();
    }
    // ---- layers 6, 5 transposed
#pragma unroll 1
    for (int l = 6; l >= 5; --l) {
        mk_cur = mk_nxt; mk_nxt = mask_rec(l - 2);
        FastMaskEpilogue<8, HALF> ep{mk_cur, a.zero, nxt, gwin((l - 1) * 256)};
        stream_layer_bf<16, 0, 8, TERMS, (fast_bwd_chunk(4) / G) % kRingStages>(ws, fast_bwd_chunk(4) / G + (6 - l) * (8 * 16 / G), dy, none, ep);
        advance();
    }
    // ---- layer 4 transposed: 256 -> 256 features (+ 64 encoding slots = slot blocks 8, 9 of 10)
    {
        FastStashEpilogue ep{swin(kGradStashEnc), a.ray_grad != 0};
        stream_layer_bf<16, 0, 2, TERMS, (fast_bwd_chunk(6) / G) % kRingStages>(ws, fast_bwd_chunk(6) / G, dy, none, ep);
    }
    {
        mk_cur = mk_nxt; mk_nxt = mask_rec(2);
        FastMaskEpilogue<8, HALF> ep{mk_cur, a.zero, nxt, gwin(3 * 256)};
        stream_layer_bf<16, 0, 8, TERMS, (fast_bwd_chunk(7) / G) % kRingStages>(ws, fast_bwd_chunk(7) / G, dy, none, ep);
        advance();
    }
    // ---- layers 3, 2, 1 transposed
#pragma unroll 1
    for (int l = 3; l >= 1; --l) {
        mk_cur = mk_nxt; mk_nxt = mask_rec(l >= 2 ? l - 2 : 0);
        FastMaskEpilogue<8, HALF> ep{mk_cur, a.zero, nxt, gwin((l - 1) * 256)};
        stream_layer_bf<16, 0, 8, TERMS, (fast_bwd_chunk(8) / G) % kRingStages>(ws, fast_bwd_chunk(8) / G + (3 - l) * (8 * 16 / G), dy, none, ep);
        advance();
    }
    if (!a.ray_grad) return;
    // ---- layer 0 transposed: 256 -> 64 encoding slots, starting from the parked skip-connection part
    float denc[32], dvenc[16];
    {
        FastAddStashEpilogue<2> ep{swin(kGradStashEnc), denc};
        stream_layer_bf<16, 0, 2, TERMS, (fast_bwd_chunk(11) / G) % kRingStages>(ws, fast_bwd_chunk(11) / G, dy, none, ep);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const f32x4 v = reinterpret_cast<const f32x4*>(a.grad + (long long)(kGradStashVenc + 8 * q) * P)[qoff];
            dvenc[4 * q] = v[0]; dvenc[4 * q + 1] = v[1]; dvenc[4 * q + 2] = v[2]; dvenc[4 * q + 3] = v[3];
        }
    }
    // ---- encodings -> point / direction -> per-sample ray gradients (fp32, as in the exact chain), parked for the per-ray reduction
    float dp[3], du[3];
    enc_backward<NIW_L3D, 8, HALF>(denc, a.save + (long long)kSaveEnc * P / (HALF ? 2 : 1), P, qoff, h, dp);
    enc_backward<NIW_LVIEW, 4, HALF>(dvenc, a.save + (long long)kSaveVenc * P / (HALF ? 2 : 1), P, qoff, h, du);
    const long long ri = mc / a.S;
    const float d = a.depth[mc];
    const float rx = a.ray[ri * 3], ry = a.ray[ri * 3 + 1], rz = a.ray[ri * 3 + 2];
    const float nrm = fmaxf(sqrtf(rx * rx + ry * ry + rz * rz), 1e-12f);
    const float ux = rx / nrm, uy = ry / nrm, uz = rz / nrm;
    const float dot = ux * du[0] + uy * du[1] + uz * du[2];
    float gc[3] = {dp[0], dp[1], dp[2]};
    float gr[3] = {dp[0] * d + (du[0] - ux * dot) / nrm, dp[1] * d + (du[1] - uy * dot) / nrm, dp[2] * d + (du[2] - uz * dot) / nrm};
    if (!valid) {
#pragma unroll
        for (int c = 0; c < 3; ++c) gc[c] = gr[c] = 0.f;
    }
    reinterpret_cast<f32x4*>(a.grad + (long long)kGradStashEnc * P)[qoff] =
        h == 0 ? f32x4{gc[0], gc[1], gc[2], gr[0]} : f32x4{gr[1], gr[2], 0.f, 0.f};
}

}  // namespace

int niw_launch_ray_grad_reduce(const float* gradws, long long mpad, int64_t n_rays, int n_samples, float* d_center, float* d_ray, hipStream_t stream);

int niw_launch_mlp_bwd_dx_fast(int precision, const void* image, const float* center, const float* ray, const float* depth, int64_t n_rays,
                               int n_samples, int density_activ, const float* rgb, const float* d_rgb, const float* d_sigma, const float* save,
                               float* gradws, float* d_center, float* d_ray, hipStream_t stream) {
    FastBwdArgs a;
    a.image = reinterpret_cast<const unsigned char*>(image);
    a.center = center; a.ray = ray; a.depth = depth; a.rgb = rgb; a.d_rgb = d_rgb; a.d_sigma = d_sigma; a.save = save; a.grad = gradws;
    a.M = n_rays * (int64_t)n_samples; a.Mpad = niw_mlp_padded_rows(n_rays, n_samples);
    a.S = n_samples; a.act = density_activ; a.ray_grad = (d_center != nullptr && d_ray != nullptr) ? 1 : 0;
    a.zero = 0;
    const int blocks = (int)(a.Mpad / 128);
#define NIW_FAST_BWD(T)                                                                                                     \
    do {                                                                                                                    \
        static std::atomic<unsigned long long> attr{0ull};                                                                  \
        if (int rc = niw_ensure_dynamic_lds(reinterpret_cast<const void*>(mlp_bwd_dx_fast_kernel<T>), kFastLdsBytes, attr, "niw_mlp_bwd_dx (fast)")) return rc; \
        mlp_bwd_dx_fast_kernel<T><<<blocks, 256, kFastLdsBytes, stream>>>(a);                                               \
    } while (0)
    if (precision == NIW_PREC_BF16X3) NIW_FAST_BWD(3); else NIW_FAST_BWD(1);
#undef NIW_FAST_BWD
    NIW_LAUNCH_CHECK("niw_mlp_bwd (dX chain, fast precision)");
    if (a.ray_grad) return niw_launch_ray_grad_reduce(gradws, a.Mpad, n_rays, n_samples, d_center, d_ray, stream);
    return NIW_OK;
}
