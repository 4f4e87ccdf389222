// 256 x 256 x 64 fp16 MFMA GEMM with fused epilogues for the encoder (gfx950):
//   C[m][n] = sum_k A[m][k] * B[n][k],  A [M,K] and B [N,K] row-major f16 (nn.Linear weight layout,
//   so y = x W^T needs no transpose), fp32 accumulation.
// Epilogues (what the reference computes after each nn.Linear, fused here):
//   EPI_QK     out16 = (acc + bias[n]) * (n < scale_cols ? scale : 1)      Q | K projection, Q/8
//   EPI_GELU   out16 = gelu_erf(acc + bias[n])                            intermediate.dense
//   EPI_RES32  out32 = acc + bias[n] + res32[m][n]                        attention.output.dense / output.dense
//   EPI_VT     out16[m][col[n]] = acc + bias[m]   (n < n_valid)           V^T = Wv . h^T, key-contiguous
//
// Why this tile: a 128 x 128 tile has 64 FLOP per staged byte, i.e. 39 TB/s of L2 traffic at the
// 2.5 PFLOP/s MFMA rate -- more than the 34.5 TB/s the eight L2s deliver -- so it is L2-bound by
// construction.  256 x 256 halves that.  8 waves (2 per SIMD), one workgroup per CU, 128 KiB of LDS.
// The main loop is the ping-pong pipeline of pipe256.h (half-tile staging with counted vmcnt, the
// two waves of a SIMD alternating between LDS reads and MFMAs).
//
// MFMA orientation is SWAPPED with respect to the output: MFMA rows (A operand) are B-matrix rows
// n, MFMA columns (B operand) are A-matrix rows m.  In the C-layout a lane then owns one output
// row m and, per register quad, 4 CONSECUTIVE n -- so the epilogue writes 8/16-byte pieces into a
// wave-private row-major LDS slab and reads it back as whole 16-byte row segments: global stores
// (and the residual read) are full-width and coalesced instead of 4-byte scalars.
//
// Staging is direct-to-LDS (global_load_lds_dwordx4): the LDS image is lane-linear, so the XOR
// swizzle is applied to the per-lane SOURCE address.  Measured alternatives on MI355X are listed in
// DESIGN.md section 9 (register staging, a BK=32 ring, LDS-DMA placements, staggered starts, ...).
// Template ABLATE compiles the measurement switches of ance_debug_gemm in (the two-phase loop the
// pipeline replaced, with or without loads / MFMA; the pipeline's own ablations); the product
// instance has none of them.
#include "common.h"
#include "gemm_f16.h"
#include "gemm256_epilogue.h"
#include "pipe256.h"
#include <stdlib.h>
#include <string.h>

namespace ance {
namespace {

constexpr int TM = 256, TN = 256, TK = 64;      // TK halves = 128 B per LDS row
constexpr int OPER_HALVES = 256 * TK;           // one operand tile (32 KiB)
constexpr int STAGE_HALVES = 2 * OPER_HALVES;   // A-rows tile + B-rows tile (64 KiB)
constexpr int G256_THREADS = 512;
constexpr size_t G256_LDS_BYTES = (size_t)2 * STAGE_HALVES * sizeof(_Float16);  // 128 KiB

typedef __attribute__((address_space(3))) void lds_void_t;
typedef const __attribute__((address_space(1))) void glb_void_t;

// Two-phase loop (one barrier + full vmcnt drain per K-tile): the structure the ping-pong pipeline
// replaced.  Kept only behind ance_debug_gemm's ablation switches (1: no loads after the first tile,
// 2: no MFMA, 4: every block loads tile (0,0), 8: none -- plain A/B reference).
__device__ __forceinline__ void two_phase_loop(const GemmArgs &G, f32x16 (&acc)[2][4], _Float16 *smem, int m0, int n0, int w,
                                               int l) {
    const int g = l >> 5, i = l & 31, wm = w >> 2, wn = w & 3;
    // ---- staging ------------------------------------------------------------------------------
    // LDS image of an operand tile: row r (128 B), 16-byte chunk c stored in slot c ^ ((r >> 1) & 7).
    // One wave-instruction covers 8 rows x 8 chunks = 1 KiB; wave w handles row groups w, w+8, ...
    // lane L: row-in-group L >> 3, LDS slot L & 7  ->  source chunk = slot ^ swizzle(row).
    const int rg = l >> 3, slot = l & 7;
    const _Float16 *srcA[4], *srcB[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int row = (j * 8 + w) * 8 + rg;
        const int ch = slot ^ ((row >> 1) & 7);
        const int ml = ((G.debug_mode & 4)) ? 0 : m0, nl = ((G.debug_mode & 4)) ? 0 : n0;  // ablation: L2-resident operands
        srcA[j] = G.A + (size_t)(ml + row) * G.lda + ch * 8;
        srcB[j] = G.B + (size_t)(nl + row) * G.ldb + ch * 8;
    }
    auto stage_issue = [&](int kt, int buf) {
        const int k0 = kt * TK;
        _Float16 *sa = smem + buf * STAGE_HALVES;
        _Float16 *sb = sa + OPER_HALVES;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int piece = ((j * 8 + w) * 8) * TK;  // wave-uniform LDS base of this 1 KiB piece
            __builtin_amdgcn_global_load_lds((glb_void_t *)(srcA[j] + k0), (lds_void_t *)(sa + piece), 16, 0, 0);
            __builtin_amdgcn_global_load_lds((glb_void_t *)(srcB[j] + k0), (lds_void_t *)(sb + piece), 16, 0, 0);
        }
    };

    // ---- fragment rows: MFMA rows <- B-matrix rows (n), MFMA cols <- A-matrix rows (m) -----------
    int nrow[2], nsw[2], mrow[4], msw[4];
#pragma unroll
    for (int x = 0; x < 2; ++x) {
        nrow[x] = wn * 64 + x * 32 + i;
        nsw[x] = (nrow[x] >> 1) & 7;
    }
#pragma unroll
    for (int x = 0; x < 4; ++x) {
        mrow[x] = wm * 128 + x * 32 + i;
        msw[x] = (mrow[x] >> 1) & 7;
    }

    const int NK = G.K / TK;
    stage_issue(0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int kt = 0; kt < NK; ++kt) {
        const int buf = kt & 1;
        if (kt + 1 < NK && !((G.debug_mode & 1))) stage_issue(kt + 1, buf ^ 1);
        const _Float16 *sa = smem + buf * STAGE_HALVES;  // A-matrix rows (m)
        const _Float16 *sb = sa + OPER_HALVES;           // B-matrix rows (n)
        // fragments of k-step s+1 are requested before the MFMAs of k-step s are issued
        f16x8 fn[2][2], fm[2][4];
        auto load_frags = [&](int s, int set) {
            const int ch = 2 * s + g;
#pragma unroll
            for (int x = 0; x < 2; ++x)
                fn[set][x] = *reinterpret_cast<const f16x8 *>(sb + nrow[x] * TK + ((ch ^ nsw[x]) * 8));
#pragma unroll
            for (int y = 0; y < 4; ++y)
                fm[set][y] = *reinterpret_cast<const f16x8 *>(sa + mrow[y] * TK + ((ch ^ msw[y]) * 8));
        };
        auto mfma_step = [&](int set) {
            if ((G.debug_mode & 2)) {  // ablation: keep the LDS reads alive, skip the matrix pipe
#pragma unroll
                for (int x = 0; x < 2; ++x) asm volatile("" ::"v"(fn[set][x]));
#pragma unroll
                for (int y = 0; y < 4; ++y) asm volatile("" ::"v"(fm[set][y]));
                return;
            }
#pragma unroll
            for (int x = 0; x < 2; ++x)
#pragma unroll
                for (int y = 0; y < 4; ++y)
                    acc[x][y] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fn[set][x], fm[set][y], acc[x][y], 0, 0, 0);
        };
        load_frags(0, 0);
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            if (s + 1 < 4) load_frags(s + 1, (s + 1) & 1);
            mfma_step(s & 1);
        }
        // an LDS-DMA is a pending LDS write on the VM counter: retire it before the barrier publishes it
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
    }

}

// Product kernel, second form (ANCE_GEMM_DESC=0 selects the first one below for A/B): operands through buffer
// descriptors (PipeSrcDesc), epilogue passes ordered inside the wave instead of by workgroup barriers.
#ifdef ANCE_MEASURE
// measurement library: per-workgroup 100 MHz stamps of the folded kernels (ance_debug_gemm_stamps): [block][8] =
// start, main loop done, statistics ready, epilogue done, and for EPI_RESLN the end of each of its four passes
__device__ unsigned long long *g_gemm_stamps = nullptr;
#define GSTAMP(slot) do { if (stamps_ && tid == 0) stamps_[(size_t)blockIdx.x * 8 + (slot)] = __builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define GSTAMP(slot) do { } while (0)
#endif

// XCD-aware tile order (speed only).  Blocks b, b + 8, ... share an XCD (and its 4 MiB L2).  The dimension with more tiles
// is dealt round-robin to the XCDs, the other one is swept fastest, so the panel of the outer dimension stays in that XCD's L2
// while the inner panels stream through it.  N-SPLIT (round 4): when the inner dimension's operand does not fit the L2 -- FFN1:
// 12 weight tiles x 384 KiB = 4.5 MiB, re-fetched for every token panel: 994 MB of L2 fills + writes per launch against
// 510 MB algorithmic (profiles/pmc_traffic.json, r03) -- the XCDs pair up: XCD x sweeps only the N-tiles of half x & 1 (2.25 MiB,
// resident) for the token panels = x >> 1 (mod 4).  A token panel is then fetched by two XCDs instead of one, the weights
// by every XCD once.
__device__ __forceinline__ bool tile_of_block(const GemmArgs &G, int b, int *mt_, int *nt_) {
    const int NT = G.N / TN, MT = G.M / TM;
    const int xcd = b & 7, jx = b >> 3;
    int mt, nt;
    if (G.n_split == 2) {
        const int nh = NT >> 1;
        mt = (jx / nh) * 4 + (xcd >> 1);
        nt = (xcd & 1) * nh + jx % nh;
    } else if (MT >= NT) {
        mt = (jx / NT) * 8 + xcd;
        nt = jx % NT;
    } else {
        nt = (jx / MT) * 8 + xcd;
        mt = jx % MT;
    }
    *mt_ = mt;
    *nt_ = nt;
    return mt < MT && nt < NT;
}

template <int EPI>
__global__ void __launch_bounds__(G256_THREADS, 2) gemm256_f16_desc_kernel(const GemmArgs G) {
    extern __shared__ __attribute__((aligned(16))) float smem_f[];
    _Float16 *smem = reinterpret_cast<_Float16 *>(smem_f);
    int mt, nt;
    if (!tile_of_block(G, blockIdx.x, &mt, &nt)) return;
    const int m0 = mt * TM, n0 = nt * TN;
    const int tid = threadIdx.x;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l = tid & 63;
    f32x16 acc[2][4];
#pragma unroll
    for (int x = 0; x < 2; ++x)
#pragma unroll
        for (int y = 0; y < 4; ++y) acc[x][y] = f32x16{0};
    Pipe256T<PipeSrcDesc, false, true, true> P;
    P.init(smem, w, l);
    // descriptors of this tile's 256 rows of each operand (bases are wave-uniform: kernel arguments and blockIdx)
    P.S.ra = __builtin_amdgcn_make_buffer_rsrc(const_cast<_Float16 *>(G.A + (size_t)m0 * G.lda), 0, (int)(256u * (uint32_t)G.lda * 2u), 0x00020000);
    P.S.rb = __builtin_amdgcn_make_buffer_rsrc(const_cast<_Float16 *>(G.B + (size_t)n0 * G.ldb), 0, (int)(256u * (uint32_t)G.ldb * 2u), 0x00020000);
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int r = pipe_stage_row(w, l, j), ch = pipe_stage_chunk(r, l);
            P.S.voff[h][j] = (uint32_t)(pipe_a_tile_row(h, r) * G.lda + ch) * 2u;
            P.S.voff[2 + h][j] = (uint32_t)(pipe_b_tile_row(h, r) * G.ldb + ch) * 2u;
        }
    constexpr bool EPB = EPI >= EPI_RESLN;  // folded-LayerNorm epilogues: parameter block by LDS-DMA, ahead of the pipeline's own
#ifdef ANCE_MEASURE
    unsigned long long *stamps_ = (EPI == EPI_RESLN && G.debug_mode == 64) ? g_gemm_stamps : nullptr;
#endif
    GSTAMP(0);
    if constexpr (EPB) epb_issue<EPI>(G, smem_f, m0, n0, w, l);
    constexpr bool FOLDK = EPI == EPI_QK_F || EPI == EPI_GELU_F || EPI == EPI_VT_F;
    if constexpr (FOLDK) {
        // A tile with a token whose |mean| >> std runs the K loop a second time over the lo halves of the token operand:
        // acc += lo . W^T (the identity  r (acc - mu c) + b'  holds for any operand, so nothing else changes).  Uniform branch,
        // taken by no tile of a random-init model.  The second pass is MASKED per row: only the wide-mean tokens feed their lo
        // halves, every other lane feeds zeros -- the bits of a row therefore depend on that row alone, not on which tokens
        // share its tile (micro-batch composition, shard boundaries, stale pad rows): ADVICE r4, tests/test_gpu_encoder.py::
        // test_fp16_rows_do_not_depend_on_their_tile_mates.
        P.run(G.K / TK, acc);
        GSTAMP(1);
        const bool wide = epb_stats(G, smem_f, tid);
        if (wide && G.tok_lo) {
            Pipe256T<PipeSrcDesc, false, true, true, false, true> P2;
            P2.init(smem, w, l);
            P2.S = P.S;
            const float *st = smem_f + EPB_OFF + EPB_STATS;
            const int i = l & 31;
            auto tok_wide = [&](int t) { return __builtin_fabsf(st[2 * t]) * st[2 * t + 1] > FOLD_WIDE_MEAN; };
            if constexpr (EPI == EPI_VT_F) {  // tokens are the B-operand rows n = wn * 64 + x * 32 + i
                P2.S.rb = __builtin_amdgcn_make_buffer_rsrc(const_cast<_Float16 *>(G.tok_lo + (size_t)n0 * G.ldb), 0, (int)(256u * (uint32_t)G.ldb * 2u), 0x00020000);
                P2.keep_b = (tok_wide((w & 3) * 64 + i) ? 1u : 0u) | (tok_wide((w & 3) * 64 + 32 + i) ? 2u : 0u);
            } else {                          // tokens are the A-operand rows m = wm * 128 + y * 32 + i
                P2.S.ra = __builtin_amdgcn_make_buffer_rsrc(const_cast<_Float16 *>(G.tok_lo + (size_t)m0 * G.lda), 0, (int)(256u * (uint32_t)G.lda * 2u), 0x00020000);
                unsigned k = 0;
#pragma unroll
                for (int y = 0; y < 4; ++y) k |= tok_wide((w >> 2) * 128 + y * 32 + i) ? (1u << y) : 0u;
                P2.keep_a = k;
            }
            P2.run(G.K / TK, acc);
        }
    } else {
        P.run(G.K / TK, acc);
        GSTAMP(1);
        if constexpr (EPB) (void)epb_stats(G, smem_f, tid);
    }
    GSTAMP(2);
#ifdef ANCE_MEASURE
    gemm256_epilogue<EPI, true>(G, acc, smem_f, m0, n0, w, l, stamps_ ? stamps_ + (size_t)blockIdx.x * 8 + 4 : nullptr);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#else
    gemm256_epilogue<EPI, true>(G, acc, smem_f, m0, n0, w, l);
#endif
    GSTAMP(3);
}

// SPLIT (fp32-grade) GEMM: C = A B^T with both operands as fp16 (hi, lo) pair rows (common.h; v = hi + lo),
//   C = sum_k  A_hi B_hi + A_lo B_hi + A_hi B_lo                  (the dropped lo x lo term is 2^-22 relative)
// Round 5: the pair rows are BLOCKED -- 32 columns of hi, then the same 32 columns of lo -- so the operand matrices look like
// plain [rows, 2 K] fp16 matrices to the staging side (PipeSrcDesc, 128 contiguous bytes per row and K-tile, exactly the fp16
// GEMM's access pattern) and a K-tile in LDS holds a 32-deep k-slice of hi AND lo of both operands: FOUR operand tiles staged
// and read once for THREE products (Pipe256T<.., PAIR3>: 48 MFMAs per K-tile and wave instead of 32), all into one accumulator
// set.  Round 4 ran three passes over [hi | lo'] rows (six staged tiles per three products, a 2^-11 rescale between the
// passes): per MFMA this loop issues two thirds of the LDS-DMAs, ds_reads and barriers -- the resources the K loop is bound by
// (DESIGN.md 3.2).  One accumulator means one scale: lo is NOT multiplied by 2^11 any more; activations are O(1) (their
// elements below 2^-3 have subnormal lo halves: at most 2^-25 absolute), weights are stored times a per-matrix power of two
// (G.wscale_inv undoes it in the epilogue).  Every partial product is exact in fp32 (11 x 11 bits); fp32 accumulation is what
// is left.  tests/test_split_model.py restates the rounding points on the CPU (3.0e-6 against the fp64 oracle at 12 layers;
// plain fp32: 2.8e-6).
template <int EPI>
__global__ void __launch_bounds__(G256_THREADS, 2) gemm256_split_kernel(const GemmArgs G) {
    extern __shared__ __attribute__((aligned(16))) float smem_f[];
    _Float16 *smem = reinterpret_cast<_Float16 *>(smem_f);
    int mt, nt;
    if (!tile_of_block(G, blockIdx.x, &mt, &nt)) return;
    const int m0 = mt * TM, n0 = nt * TN;
    const int tid = threadIdx.x;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l = tid & 63;
    f32x16 acc[2][4];
#pragma unroll
    for (int x = 0; x < 2; ++x)
#pragma unroll
        for (int y = 0; y < 4; ++y) acc[x][y] = f32x16{0};
    Pipe256T<PipeSrcDesc, false, true, true, true> P;
    P.init(smem, w, l);
    P.S.ra = __builtin_amdgcn_make_buffer_rsrc(const_cast<_Float16 *>(G.A + (size_t)m0 * G.lda), 0, (int)(256u * (uint32_t)G.lda * 2u), 0x00020000);
    P.S.rb = __builtin_amdgcn_make_buffer_rsrc(const_cast<_Float16 *>(G.B + (size_t)n0 * G.ldb), 0, (int)(256u * (uint32_t)G.ldb * 2u), 0x00020000);
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int r = pipe_stage_row(w, l, j), ch = pipe_stage_chunk(r, l);
            P.S.voff[h][j] = (uint32_t)(pipe_a_tile_row(h, r) * G.lda + ch) * 2u;
            P.S.voff[2 + h][j] = (uint32_t)(pipe_b_tile_row(h, r) * G.ldb + ch) * 2u;
        }
    const float winv = G.wscale_inv ? *G.wscale_inv : 1.0f;  // wave-uniform: a scalar load, long back when the epilogue starts
#ifdef ANCE_MEASURE
    unsigned long long *stamps_ = G.debug_mode == 64 ? g_gemm_stamps : nullptr;  // [block][8]: start, main loop done, statistics, end
#endif
    GSTAMP(0);
    epb_issue<EPI>(G, smem_f, m0, n0, w, l);
    P.run(G.K / 32, acc);  // K-tile = 64 halves of a blocked pair row = 32 k of hi and lo
    GSTAMP(1);
    (void)epb_stats(G, smem_f, tid);
    GSTAMP(2);
    gemm256_epilogue_split<EPI>(G, acc, smem_f, m0, n0, w, l, winv);
#ifdef ANCE_MEASURE
    if (stamps_) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
    GSTAMP(3);
}

// PERSISTENT, STREAMING form of the split GEMM (round 6): the product's kernel for the QKV and FFN1 GEMMs (ANCE_GEMM_STREAM, below).
// One workgroup per CU walks its own sequence of output tiles; the K loop of tile i ends as Pipe256T::tiles_streaming -- its last
// two K-tiles stage K-tile 0 and A0 / B0 / B1 of K-tile 1 of tile i + 1 in the steady-state rhythm -- so tile i's epilogue runs with
// the next tile's first operand bytes in flight and the next K loop starts where a steady-state K-tile starts: no pipeline fill
// (3 us of a 58-62 us tile at K = 768), no workgroup launch between two tiles, no launch ramp; the epilogue's stores drain under
// the next tile's first MFMA phase.  What that costs is LDS: the stage buffers are busy during the epilogue, so the epilogue lives
// in a quarter of the slab space (gemm256_epilogue_split32: 32 x 32 passes) --
//   [0, 128 KiB)          stage buffers; the A-half1 slot of buffer 1 is free between two tiles (K-tile 1's A-half1 is staged by
//                         P0 of K-tile 0): slabs of waves 0-2
//   STATS 2 KiB, VEC 3 KiB   (mean, rstd) of the tile's 256 tokens; bias | csum or gamma | beta of its 256 features
//   R 27 KiB              during the K loop the slice partials of the tile's tokens (24 KiB, LDS-DMA issued after the previous
//                         epilogue, retired by the pipeline's counted waits); during the epilogue the slabs of waves 3-7
// = 160 KiB exactly.  Tile order: workgroup b is on XCD b & 7 (round-robin dispatch) and takes the virtual blocks
// ((i * slots + (b >> 3)) << 3) | xcd, i = 0, 1, ... of tile_of_block's order -- at any time the 32 CUs of an XCD work on the 32
// consecutive blocks the launch-per-tile kernel would have had in flight there, so the L2 behaviour (and the N-split order of FFN1)
// carries over.  Results are bit-identical to the kernel above (same K order, same epilogue arithmetic, same reduction trees).
constexpr int EPS_STATS = 32768;                 // floats: above the 128 KiB of stage buffers
constexpr int EPS_VEC = EPS_STATS + 512;
constexpr int EPS_R = EPS_VEC + 768;             // 27,648 bytes: slice partials (24 KiB) | slabs of waves 3-7 (5 x 4,608 B)
constexpr size_t GS_LDS_BYTES = (size_t)(EPS_R + 256 * 24 + 768) * sizeof(float);  // 163,840
static_assert(GS_LDS_BYTES == 160 * 1024, "the streaming kernel uses the whole LDS of a CU");
static_assert(5 * EPS_SLAB_FLOATS <= 256 * 24 + 768 && 3 * EPS_SLAB_FLOATS * 4 <= 16384, "slab layout");
#ifndef ANCE_STREAM_LOOSE_FIRST
#define ANCE_STREAM_LOOSE_FIRST 1  // 1: K-tile 0 of a prefetched tile does not wait for the previous epilogue's stores (pipe256.h: tile2); 0: steady-state waits
#endif
// vector-memory operations EVERY wave issues between the hand-over's last LDS-DMA and K-tile 0 of the next output tile: the
// epilogue's stores (gemm256_epilogue_split32: 8 passes x 4 16-byte stores of fp32 rows for QKV, 8 x 2 x (hi + lo) for the pair rows of
// GELU: 32 either way) and the three LDS-DMAs of the slice partials (eps_issue; the vector DMAs are issued by two waves only and do not
// count)
constexpr int EPS_FOREIGN_OPS = 32 + 3;

template <int EPI_>
__device__ __forceinline__ void eps_issue(const GemmArgs &G, float *smem_f, int m0, int n0, int w, int l) {
    typedef __attribute__((address_space(3))) void lds_t;
    typedef const __attribute__((address_space(1))) void glb_t;
    const float *psrc = G.part_in + (size_t)m0 * 24;
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        const int piece = w + 8 * j;  // 24 pieces of 1 KiB
        __builtin_amdgcn_global_load_lds((glb_t *)(psrc + piece * 256 + l * 4), (lds_t *)(smem_f + EPS_R + piece * 256), 16, 0, 0);
    }
    if (w == 0) __builtin_amdgcn_global_load_lds((glb_t *)(G.bias + n0 + l * 4), (lds_t *)(smem_f + EPS_VEC), 16, 0, 0);
    if (w == 1) __builtin_amdgcn_global_load_lds((glb_t *)(G.csum + n0 + l * 4), (lds_t *)(smem_f + EPS_VEC + 256), 16, 0, 0);
}

// the kernel's GemmArgs re-read from the kernarg segment (first argument) behind an opaque pointer: see the call site
__device__ __forceinline__ GemmArgs kernarg_reload(const GemmArgs &G) {
#if defined(__HIP_DEVICE_COMPILE__)
    typedef const __attribute__((address_space(4))) GemmArgs *kernarg_ptr_t;
    kernarg_ptr_t gp = (kernarg_ptr_t)__builtin_amdgcn_kernarg_segment_ptr();
    asm volatile("" : "+s"(gp));
    (void)G;
    return *gp;
#else
    return G;
#endif
}

// workgroup barrier that does NOT drain the vector-memory counter (the next tile's LDS-DMAs and this tile's stores stay in flight)
__device__ __forceinline__ void eps_barrier() {
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
}

template <int EPI>
__global__ void __launch_bounds__(G256_THREADS, 2) gemm256_split_stream_kernel(const GemmArgs G, const int n_blocks) {
    extern __shared__ __attribute__((aligned(16))) float smem_f[];
    _Float16 *smem = reinterpret_cast<_Float16 *>(smem_f);
    const int tid = threadIdx.x;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l = tid & 63;
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3, nslots = gridDim.x >> 3;
    int round = 0;
    // next valid tile of this workgroup's sequence (virtual blocks past the matrix -- the padding of tile_of_block's order -- are skipped)
    auto next_tile = [&](int *mt, int *nt) -> bool {
        for (;;) {
            const int b = ((round * nslots + slot) << 3) | xcd;
            if (b >= n_blocks) return false;
            ++round;
            if (tile_of_block(G, b, mt, nt)) return true;
        }
    };
    int mt, nt;
    if (!next_tile(&mt, &nt)) return;
    Pipe256T<PipeSrcStream, false, true, true, true> P;
    P.init(smem, w, l);
    P.S.ra = __builtin_amdgcn_make_buffer_rsrc(const_cast<_Float16 *>(G.A), 0, (int)((uint32_t)G.M * (uint32_t)G.lda * 2u), 0x00020000);
    P.S.rb = __builtin_amdgcn_make_buffer_rsrc(const_cast<_Float16 *>(G.B), 0, (int)((uint32_t)G.N * (uint32_t)G.ldb * 2u), 0x00020000);
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int r = pipe_stage_row(w, l, j), ch = pipe_stage_chunk(r, l);
            P.S.voff[h][j] = (uint32_t)(pipe_a_tile_row(h, r) * G.lda + ch) * 2u;
            P.S.voff[2 + h][j] = (uint32_t)(pipe_b_tile_row(h, r) * G.ldb + ch) * 2u;
        }
    const int NK = G.K / 32;  // K-tile = 64 halves of a blocked pair row = 32 k of hi and lo
    P.S.NK = NK;
    P.S.a_cur = P.S.a_nxt = (uint32_t)mt * 256u * (uint32_t)G.lda * 2u;
    P.S.b_cur = P.S.b_nxt = (uint32_t)nt * 256u * (uint32_t)G.ldb * 2u;
    const float winv = G.wscale_inv ? *G.wscale_inv : 1.0f;
    // (slab of a wave: waves 0-2 in the A-half1 slot of stage buffer 1, waves 3-7 in R)
    eps_issue<EPI>(G, smem_f, mt * TM, nt * TN, w, l);
    if constexpr (ANCE_STREAM_LOOSE_FIRST) P.prologue_landed(); else P.prologue();
    P.enter();
    for (;;) {
        int mtn = 0, ntn = 0;
        const bool have_n = next_tile(&mtn, &ntn);
        f32x16 acc[2][4];
#pragma unroll
        for (int x = 0; x < 2; ++x)
#pragma unroll
            for (int y = 0; y < 4; ++y) acc[x][y] = f32x16{0};
        if (have_n) {
            P.S.a_nxt = (uint32_t)mtn * 256u * (uint32_t)G.lda * 2u;
            P.S.b_nxt = (uint32_t)ntn * 256u * (uint32_t)G.ldb * 2u;
        }
        constexpr int F = ANCE_STREAM_LOOSE_FIRST ? EPS_FOREIGN_OPS : 0;
        constexpr int VM0F = 6 + F > 63 ? 63 : 6 + F, VM1F = 2 + F > 63 ? 63 : 2 + F;
        // (the first tile of the workgroup comes from prologue_landed: nothing in flight, the loose waits of its K-tile 0 are trivially enough)
        if (have_n) P.template tiles_streaming<VM0F, VM1F>(NK, acc); else P.template tiles_final<VM0F, VM1F>(NK, acc);
        P.leave();
        // (the lane id is laundered through an empty asm: hipcc otherwise hoists every lane-derived address of the epilogue out of the
        // tile loop, runs out of registers next to the 128 accumulators and reloads them from scratch here -- and a scratch reload is a
        // vmcnt(0) wait that drains the next tile's prefetch; ip_topk_fast.hip found the same)
        int lf = l;
        asm volatile("" : "+v"(lf));
        const int tf = (w << 6) | lf;
        // (the same for the kernel arguments: ~50 SGPRs of pointers and strides that only the epilogue uses would otherwise stay live
        // across the K loop -- re-read from the kernarg segment per tile instead)
        const GemmArgs Ge = kernarg_reload(G);
        // (mean, rstd) of the tile's tokens from their slice partials (in R since before this K loop), published by one barrier;
        // after it R belongs to the slabs
        if (tf < 256) {
            float mean, rstd;
            stats_from_parts(smem_f + EPS_R + tf * 24, Ge.ln_eps, &mean, &rstd);
            smem_f[EPS_STATS + 2 * tf] = mean;
            smem_f[EPS_STATS + 2 * tf + 1] = rstd;
        }
        eps_barrier();
        float *slab = w < 3 ? smem_f + (PIPE_BUF_HALVES + PIPE_HALF_HALVES) / 2 + w * EPS_SLAB_FLOATS : smem_f + EPS_R + (w - 3) * EPS_SLAB_FLOATS;
        gemm256_epilogue_split32<EPI>(Ge, acc, slab, smem_f + EPS_STATS, smem_f + EPS_VEC, mt * TM, nt * TN, w, lf, winv);
        if (!have_n) break;
        eps_barrier();  // every wave is done with STATS, VEC and its slab: R and the A-half1 slot may be refilled
        mt = mtn;
        nt = ntn;
        P.S.a_cur = P.S.a_nxt;
        P.S.b_cur = P.S.b_nxt;
        eps_issue<EPI>(Ge, smem_f, mt * TM, nt * TN, w, lf);
        P.enter();
    }
}

template <int EPI, bool ABLATE>
__global__ void __launch_bounds__(G256_THREADS, 2) gemm256_f16_kernel(const GemmArgs G) {
    extern __shared__ __attribute__((aligned(16))) float smem_f[];
    _Float16 *smem = reinterpret_cast<_Float16 *>(smem_f);

    const int NT = G.N / TN, MT = G.M / TM;
    // XCD-aware tile order (speed only): blocks b, b+8, ... share an XCD.  The dimension with more
    // tiles is dealt round-robin to the XCDs, the other one is swept fastest, so the panel of the
    // outer dimension stays in that XCD's L2 while the inner panels stream through it.
    const int b = blockIdx.x, xcd = b & 7, jx = b >> 3;
    int mt, nt;
    if (MT >= NT) {
        mt = (jx / NT) * 8 + xcd;
        nt = jx % NT;
    } else {
        nt = (jx / MT) * 8 + xcd;
        mt = jx % MT;
    }
    if (mt >= MT || nt >= NT) return;
    const int m0_ = mt * TM, n0_ = nt * TN;

    const int tid = threadIdx.x;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l = tid & 63;  // wave w: output rows m wm*128.. (wm = w >> 2), columns n wn*64.. (wn = w & 3)

    f32x16 acc[2][4];
#pragma unroll
    for (int x = 0; x < 2; ++x)
#pragma unroll
        for (int y = 0; y < 4; ++y) acc[x][y] = f32x16{0};

    if (!ABLATE || (G.debug_mode & (16 | 32))) {
        // product path: ping-pong pipeline of pipe256.h (debug_mode 16 + bits: its ablations)
        Pipe256T<PipeSrcFixed, ABLATE, true, !ABLATE> P;  // product: coarse (2-phase) schedule; ablation build: the 4-phase one
        P.init(smem, w, l);
        P.dbg = G.debug_mode;
        P.S.dbg = ABLATE ? G.debug_mode : 0;
        const int m0 = (ABLATE && (G.debug_mode & 4)) ? 0 : m0_, n0 = (ABLATE && (G.debug_mode & 4)) ? 0 : n0_;
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int r = pipe_stage_row(w, l, j), ch = pipe_stage_chunk(r, l);
                P.S.src[h][j] = G.A + (size_t)(m0 + pipe_a_tile_row(h, r)) * G.lda + ch;
                P.S.src[2 + h][j] = G.B + (size_t)(n0 + pipe_b_tile_row(h, r)) * G.ldb + ch;
            }
        if (ABLATE && (G.debug_mode & 32)) {
            // timeline mode (epi 0 / 1 only): 100 MHz real-time stamps of this workgroup's phases go to the
            // otherwise unused res32 buffer as uint64[workgroup][5]: start, prologue done, main loop done,
            // epilogue issued, stores drained
            unsigned long long *ts = reinterpret_cast<unsigned long long *>(const_cast<float *>(G.res32)) + (size_t)b * 5;
            const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
            P.prologue();
            const unsigned long long t1 = __builtin_amdgcn_s_memrealtime();
            P.enter();
            P.tiles_final(G.K / TK, acc);
            P.leave();
            const unsigned long long t2 = __builtin_amdgcn_s_memrealtime();
            gemm256_epilogue<EPI>(G, acc, smem_f, m0_, n0_, w, l);
            const unsigned long long t3 = __builtin_amdgcn_s_memrealtime();
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            const unsigned long long t4 = __builtin_amdgcn_s_memrealtime();
            if (tid == 0) { ts[0] = t0; ts[1] = t1; ts[2] = t2; ts[3] = t3; ts[4] = t4; }
            return;
        }
        P.run(G.K / TK, acc);
    } else {
        two_phase_loop(G, acc, smem, m0_, n0_, w, l);
    }

    gemm256_epilogue<EPI>(G, acc, smem_f, m0_, n0_, w, l);
}

#ifdef ANCE_MEASURE
unsigned long long *g_gemm_stamps_host = nullptr;
int g_gemm_stamps_epi = EPI_RESLN;  // which epilogue's launches are stamped (ance_debug_gemm_stamps_epi)
#endif

// ANCE_GEMM_STREAM: 1 (default) = the persistent streaming kernel for the QKV and FFN1 GEMMs (same-box A/Bs: -0.6 .. -1.3 % and
// -1.7 .. -3.0 % per launch), the launch-per-tile kernel for the two RESLN GEMMs (streaming them measured +1.0 % / +1.5 %:
// DESIGN_REJECTED.md round 6); 0 = launch-per-tile everywhere (A/B).  Read once per process, ance_reload_env re-reads it.
int g_gemm_stream = -1;
int gemm_stream_mode() {
    if (g_gemm_stream < 0) {
        const char *e = getenv("ANCE_GEMM_STREAM");
        g_gemm_stream = (e && e[0] == '0') ? 0 : 1;
    }
    return g_gemm_stream;
}
int device_cu_count() {
    static int cus[64] = {0};
    int dev = 0;
    (void)hipGetDevice(&dev);
    if (dev < 0 || dev >= 64) dev = 0;
    if (cus[dev] == 0) {
        int n = 0;
        if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n < 8) n = 256;
        cus[dev] = n;
    }
    return cus[dev];
}

template <bool ABLATE>
int launch256(int epi, const GemmArgs &G, hipStream_t st) {
    const int MT = G.M / TM, NT = G.N / TN;
    if (G.n_split != 0 && (G.n_split != 2 || (NT & 1) || ABLATE || epi < EPI_RESLN)) {
        set_last_error("gemm256: n_split needs an even number of N tiles and a descriptor-form kernel");
        return ANCE_E_INVALID;
    }
    const unsigned blocks = G.n_split == 2 ? (unsigned)((MT + 3) / 4 * 4) * (unsigned)NT
                            : MT >= NT     ? (unsigned)((MT + 7) / 8 * 8) * (unsigned)NT
                                           : (unsigned)((NT + 7) / 8 * 8) * (unsigned)MT;
    void (*k)(const GemmArgs) = nullptr;
    static int use_desc = -1;
    if (use_desc < 0) {
        const char *e = getenv("ANCE_GEMM_DESC");
        use_desc = (e && atoi(e) == 0) ? 0 : 1;
    }
    const bool desc = !ABLATE && use_desc;
    switch (epi) {
        case EPI_QK: k = desc ? gemm256_f16_desc_kernel<EPI_QK> : gemm256_f16_kernel<EPI_QK, ABLATE>; break;
        case EPI_GELU: k = desc ? gemm256_f16_desc_kernel<EPI_GELU> : gemm256_f16_kernel<EPI_GELU, ABLATE>; break;
        case EPI_RES32: k = desc ? gemm256_f16_desc_kernel<EPI_RES32> : gemm256_f16_kernel<EPI_RES32, ABLATE>; break;
        case EPI_VT: k = desc ? gemm256_f16_desc_kernel<EPI_VT> : gemm256_f16_kernel<EPI_VT, ABLATE>; break;
        case EPI_RESLN: k = gemm256_f16_desc_kernel<EPI_RESLN>; break;
        case EPI_QK_F: k = gemm256_f16_desc_kernel<EPI_QK_F>; break;
        case EPI_GELU_F: k = gemm256_f16_desc_kernel<EPI_GELU_F>; break;
        case EPI_VT_F: k = gemm256_f16_desc_kernel<EPI_VT_F>; break;
        case EPI_S_QKV: k = gemm256_split_kernel<EPI_S_QKV>; break;
        case EPI_S_GELU: k = gemm256_split_kernel<EPI_S_GELU>; break;
        case EPI_S_RESLN: k = gemm256_split_kernel<EPI_S_RESLN>; break;
        default: set_last_error("gemm256: bad epilogue"); return ANCE_E_INVALID;
    }
    // the dynamic-LDS attribute is per template instance AND per device
    static unsigned long long attr_done[2 * EPI_COUNT] = {0};
    const int ai = epi + ((desc || epi >= EPI_RESLN) ? EPI_COUNT : 0);
    if (attr_needed(&attr_done[ai])) {
        if (hipFuncSetAttribute(reinterpret_cast<const void *>(k), hipFuncAttributeMaxDynamicSharedMemorySize,
                                (int)(G256_LDS_BYTES + (size_t)EPB_FLOATS * sizeof(float))) != hipSuccess)
            return check_launch("gemm256 attr");
        attr_mark(&attr_done[ai]);
    }
    if (!ABLATE && (epi == EPI_S_QKV || epi == EPI_S_GELU) && gemm_stream_mode() == 1 && G.K >= 96 && (uint64_t)G.M * (uint64_t)G.lda * 2u < (1ull << 31) &&
        (uint64_t)G.N * (uint64_t)G.ldb * 2u < (1ull << 31)
#ifdef ANCE_MEASURE
        && !(epi == g_gemm_stamps_epi && g_gemm_stamps_host)
#endif
    ) {
        void (*ks)(const GemmArgs, int) = epi == EPI_S_QKV ? gemm256_split_stream_kernel<EPI_S_QKV> : gemm256_split_stream_kernel<EPI_S_GELU>;
        static unsigned long long sattr_done[2] = {0, 0};
        if (attr_needed(&sattr_done[epi - EPI_S_QKV])) {
            if (hipFuncSetAttribute(reinterpret_cast<const void *>(ks), hipFuncAttributeMaxDynamicSharedMemorySize, (int)GS_LDS_BYTES) != hipSuccess)
                return check_launch("gemm256 stream attr");
            attr_mark(&sattr_done[epi - EPI_S_QKV]);
        }
        const unsigned cus = (unsigned)device_cu_count() & ~7u;  // one workgroup per CU (160 KiB of LDS each), a multiple of the 8 XCDs
        const unsigned grid = blocks < cus ? blocks : cus;
        hipLaunchKernelGGL(ks, dim3(grid), dim3(G256_THREADS), GS_LDS_BYTES, st, G, (int)blocks);
        return ANCE_OK;
    }
    const size_t lds = epi >= EPI_RESLN ? G256_LDS_BYTES + (size_t)EPB_FLOATS * sizeof(float) : G256_LDS_BYTES;
#ifdef ANCE_MEASURE
    if (epi == g_gemm_stamps_epi && g_gemm_stamps_host) {
        GemmArgs G2 = G;
        G2.debug_mode = 64;
        hipLaunchKernelGGL(k, dim3(blocks), dim3(G256_THREADS), lds, st, G2);
        return ANCE_OK;
    }
#endif
    hipLaunchKernelGGL(k, dim3(blocks), dim3(G256_THREADS), lds, st, G);
    return ANCE_OK;
}

}  // namespace

void reload_gemm_knobs() { g_gemm_stream = -1; }

bool gemm256_applicable(const GemmArgs &G) {
    return G.M > 0 && G.N > 0 && G.K >= 2 * TK && G.M % TM == 0 && G.N % TN == 0 && G.K % TK == 0;
}

int launch_gemm_f16(int epi, const GemmArgs &G, hipStream_t st) {
    if (!gemm256_applicable(G)) {
        set_last_error("gemm_f16: M,N must be multiples of 256 and K a multiple of 64, >= 128");
        return ANCE_E_INVALID;
    }
#ifdef ANCE_MEASURE  // the ablation / timeline builds of the kernel exist in the measurement library only
    if (G.debug_mode) return launch256<true>(epi, G, st);
#else
    if (G.debug_mode) {
        set_last_error("gemm_f16: ablation and timeline modes need the measurement library (make -C ance_amd/csrc measure)");
        return ANCE_E_INVALID;
    }
#endif
    return launch256<false>(epi, G, st);
}

}  // namespace ance

#ifdef ANCE_MEASURE
// measurement library only: while d_stamps != NULL the RES GEMMs of the encoder leave uint64[8] per workgroup at d_stamps
extern "C" void ance_debug_gemm_stamps(void *d_stamps) {
    ance::g_gemm_stamps_host = reinterpret_cast<unsigned long long *>(d_stamps);
    unsigned long long *p = ance::g_gemm_stamps_host;
    (void)hipMemcpyToSymbol(HIP_SYMBOL(ance::g_gemm_stamps), &p, sizeof(p));
}
// measurement library only: the epilogue (gemm_f16.h: EPI_*) whose launches leave stamps; default EPI_RESLN.  The split kernels
// (EPI_S_QKV 8, EPI_S_GELU 9, EPI_S_RESLN 10) fill slots 0..3: start, main loop done, statistics ready, stores drained
extern "C" void ance_debug_gemm_stamps_epi(int epi) { ance::g_gemm_stamps_epi = epi; }
// measurement library only (WRONG results): see g_res_ablate in gemm256_epilogue.h
extern "C" void ance_debug_gemm_res_ablate(int bits) {
    (void)hipMemcpyToSymbol(HIP_SYMBOL(ance::g_res_ablate), &bits, sizeof(bits));
}
#endif

// Test / measurement hook (include/ance_amd.h): the encoder's GEMM kernel on caller-provided data.
extern "C" int ance_debug_gemm(int ablate, int epi, const void *d_a_f16, const void *d_b_f16, int M, int N, int K,
                               const float *d_bias, void *d_out, const float *d_res32, void *stream) {
    using namespace ance;
    if (!d_a_f16 || !d_b_f16 || !d_bias || !d_out || epi < 0 || epi > 2 || (epi == EPI_RES32 && !d_res32)) {
        set_last_error("ance_debug_gemm: invalid argument");
        return ANCE_E_INVALID;
    }
    GemmArgs G;
    memset(&G, 0, sizeof(G));
    G.A = (const _Float16 *)d_a_f16; G.lda = K; G.B = (const _Float16 *)d_b_f16; G.ldb = K;
    G.M = M; G.N = N; G.K = K; G.bias = d_bias; G.ldc = N; G.scale = 1.0f; G.scale_cols = 0;
    G.debug_mode = ablate;
    if (epi == EPI_RES32) { G.out32 = (float *)d_out; G.res32 = d_res32; }
    else G.out16 = (_Float16 *)d_out;
    if (epi != EPI_RES32 && (ablate & 32)) {  // timeline buffer
        if (!d_res32) {
            set_last_error("ance_debug_gemm: timeline mode needs d_res32 (uint64[M/256 * N/256][5])");
            return ANCE_E_INVALID;
        }
        G.res32 = d_res32;
    }
    ProfScope ps(PC_GEMM_FFN1, (hipStream_t)stream, 2.0 * M * (double)N * K);
    int rc = launch_gemm_f16(epi, G, (hipStream_t)stream);
    return rc ? rc : check_launch("ance_debug_gemm");
}

// Test hook (include/ance_amd.h): the SPLIT GEMM with each of its three epilogues on caller-provided pair operands (blocked pair
// rows: ance_pair_layout); d_wscale_inv: optional device scalar the accumulators are multiplied by (the inverse of the power of
// two the B operand was stored with).
extern "C" int ance_debug_gemm_split(int epi, const void *d_a_pair, const void *d_b_pair, int M, int N, int K, const float *d_bias,
                                     const float *d_vec1, const float *d_vec2, const float *d_part, float ln_eps,
                                     const void *d_res_pair, void *d_out, float *d_part_out, const float *d_wscale_inv, void *stream) {
    using namespace ance;
    if (!d_a_pair || !d_b_pair || !d_bias || !d_vec1 || !d_part || !d_out || epi < EPI_S_QKV || epi > EPI_S_RESLN ||
        (epi == EPI_S_RESLN && (!d_vec2 || !d_res_pair || !d_part_out || N != 768)) || K % 64 != 0) {
        set_last_error("ance_debug_gemm_split: invalid argument");
        return ANCE_E_INVALID;
    }
    GemmArgs G;
    memset(&G, 0, sizeof(G));
    G.A = (const _Float16 *)d_a_pair; G.lda = 2 * K; G.B = (const _Float16 *)d_b_pair; G.ldb = 2 * K;
    G.M = M; G.N = N; G.K = K; G.bias = d_bias; G.part_in = d_part; G.ln_eps = ln_eps; G.wscale_inv = d_wscale_inv;
    if (epi == EPI_S_QKV) {
        G.csum = d_vec1; G.out32 = (float *)d_out; G.ldc = N;
    } else if (epi == EPI_S_GELU) {
        G.csum = d_vec1; G.out16 = (_Float16 *)d_out; G.ldc = 2 * N;
    } else {
        G.res_gamma = d_vec1; G.res_beta = d_vec2; G.res_hi = (const _Float16 *)d_res_pair; G.ldr = 2 * N;
        G.out16 = (_Float16 *)d_out; G.ldc = 2 * N; G.part_out = d_part_out;
    }
    int rc = launch_gemm_f16(epi, G, (hipStream_t)stream);
    return rc ? rc : check_launch("ance_debug_gemm_split");
}

// Layout of the split mode's pair rows for tests and tools: column n of a W-wide row -> positions of its hi and lo halves in the
// 2 W-half row, and the factor lo was multiplied by (1: unscaled; the round-4 A/B build reports 2048 and rows [hi (W) | lo' (W)]).
extern "C" void ance_pair_layout(int n, int W, int *hi_col, int *lo_col, float *lo_scale) {
    if (hi_col) *hi_col = ance::pair_hi_col(n, W);
    if (lo_col) *lo_col = ance::pair_lo_col(n, W);
    if (lo_scale) *lo_scale = ance::PAIR_LO_SCALE;
}
