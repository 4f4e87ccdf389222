// Ping-pong main loop of the 256 x 256 x 64 fp16 MFMA tile (gfx950), shared by the encoder GEMMs (gemm256_f16.hip: the split
// GEMM of the default arithmetic and the fp16 GEMM of the fast mode) and the search filter (ip_topk_fast.hip).
//
//   acc[x][y] += sum_k  B[n][k] * A[m][k]      n = wn*64 + x*32 + (C-layout row)   m = wm*128 + y*32 + lane&31
//
// 8 waves (2 along m x 4 along n), 128 KiB of LDS = 2 K-tile buffers x 4 half-tiles of 128 rows x 64 halves (16 KiB each):
// A-half h holds the 64-row blocks {h, h+2} of the A tile (so a wave's fragments y = 0,1 come from A-half 0 and y = 2,3 from
// A-half 1), B-half h holds the 32-row blocks {h, h+2, h+4, h+6} of the B tile (x = h).  Half-tiles are staged by LDS-DMA
// (16 bytes per lane, lane-linear LDS image, XOR swizzle on the SOURCE address) 2-3 phases ahead with counted s_waitcnt vmcnt(N),
// never 0 in steady state; the four waves with wm = 1 run one barrier behind the four with wm = 0, so on every SIMD one wave is
// in its MFMA half-phase while the other one reads LDS.
//
// THE SCHEDULE OF THE PRODUCT KERNELS IS THE COARSE ONE (COARSE: two phases per K-tile, both B halves of a K-tile in registers):
//   phase P0: read A-half0, B-half0, B-half1   MFMAs on acc[0][0..1], acc[1][0..1]   + stage A-half1 of tile t+1
//   phase P1: read A-half1                     MFMAs on acc[1][2..3], acc[0][2..3]   + stage A0, B0, B1 of tile t+2
// with 16 MFMAs per phase on plain fp16 operands and -- PAIR3, the split GEMM of the DEFAULT arithmetic -- 24: both operands are
// BLOCKED pair rows (common.h: 32 columns of hi, then the same 32 columns of lo), so the 64 halves of an LDS row are
//   [hi k 0..15 | hi k 16..31 | lo k 0..15 | lo k 16..31]          (fragment index s = 0..3)
// i.e. a K-tile is a 32-deep k-slice of hi AND lo of both operands, and per output quadrant and k-step j the wave issues the three
// products  hi_j x hi_j,  lo_j x hi_j,  hi_j x lo_j  (fragment pairs (j, j), (j + 2, j), (j, j + 2)) into ONE accumulator: four
// operand tiles staged and read once for three products.  Staging, ds_reads, barriers and waits are identical in both forms.
//   RAW  the wait at the end of the P1 reads of tile t-1 (vmcnt(2)) leaves only A-half1 of tile t in flight, so A0 / B0 / B1 of
//        tile t are retired and the following barrier publishes them; the wait at the end of the P0 reads of tile t (vmcnt(6))
//        leaves only A0 / B0 / B1 of tile t+1 in flight, so A-half1 of tile t is retired before the barrier that precedes its read.
//   WAR  a half-tile last read in phase p is restaged in the MFMA half-phase of phase p+1 at the earliest (A0 / B0 / B1 read in
//        P0 of tile t, restaged in P1 of tile t; A1 read in P1 of tile t, restaged in P0 of t+1): the last ds_read of it (by the
//        wm = 1 group, one slot later) has returned before that wave's MFMAs of the next slot are issued (they consume it), and a
//        barrier separates that slot from the restage.
// STREAMING (tiles_streaming + a source policy that maps K-tile t >= NK onto what follows): the K loop of the NEXT tile of a
// workgroup's sequence -- the next corpus tile of the search filter, the next OUTPUT tile of the persistent split GEMM -- is
// prefetched by the last two K-tiles of the current one, in the steady-state rhythm; the epilogue / filter step runs with those
// LDS-DMAs in flight and the next K loop starts without a pipeline fill.
//
// The original FOUR-phase schedule (one 64 x 32 output quadrant of every wave per phase, 8 MFMAs each, B-half0 read twice or
// kept in 16 registers: KEEP_B0) is what the measurement builds of the fp16 GEMM and the search filter still instantiate:
//   phase c0: read A-half0 + B-half0   MFMA acc[0][0..1]  + stage B-half0 of tile t+1
//   phase c1: read B-half1             MFMA acc[1][0..1]  + stage A-half0 of tile t+2
//   phase c2: read A-half1             MFMA acc[1][2..3]  + stage B-half1 of tile t+2
//   phase c3: read B-half0 (again)     MFMA acc[0][2..3]  + stage A-half1 of tile t+2
// Every phase is  { ds_reads }  s_barrier  { MFMAs with the LDS-DMAs issued between them }  s_barrier.  Where the LDS-DMAs are
// issued was chosen by cycle counts (profiles/attic/r01_gemm_schedule_variants_cycles.txt): between the MFMAs 1.36 M cycles per
// XCD on 8192^3, in the read half-phase 1.57-1.96 M, at the start / end of the MFMA half-phase 1.48-1.50 M; the wm stagger itself
// is worth 1.36 vs 1.78 M and the two-phase loop this replaced took 1.88 M.  Its hazards (slot = interval between two barriers;
// wm = 0 reads phase p in slot 2p and computes it in slot 2p+1, wm = 1 one slot later): RAW -- the wait at the end of the c3 reads
// of tile t-1 leaves at most two half-tiles in flight (A-half0 and B-half1 of tile t+1), so every wave has retired its pieces of
// all four half-tiles of tile t and the barrier that follows publishes them; WAR -- as above.
// Needs NK >= 2 K-tiles.  The last two tiles are peeled (nothing left to stage, smaller counts).
// tests/test_pipe_schedule_model.py replays these tables (prologue, steady state, peeled tiles, streaming hand-over, both wave
// groups, every form) on a slot timeline and asserts the RAW / WAR conditions for every K-tile count.
#pragma once
#include "common.h"

namespace ance {

typedef __attribute__((address_space(3))) void pipe_lds_t;
typedef const __attribute__((address_space(1))) void pipe_glb_t;

constexpr int PIPE_HALF_HALVES = 128 * 64;            // one half-tile: 16 KiB
constexpr int PIPE_BUF_HALVES = 4 * PIPE_HALF_HALVES;  // A0 A1 B0 B1: 64 KiB
constexpr size_t PIPE_LDS_BYTES = (size_t)2 * PIPE_BUF_HALVES * sizeof(_Float16);

#define PIPE_WAIT_VM(n) asm volatile("s_waitcnt vmcnt(" #n ")" ::: "memory")
#ifndef PIPE_PAIR3_STAGE_GAP
#define PIPE_PAIR3_STAGE_GAP 2  // MFMA pairs between two LDS-DMA pieces of a PAIR3 phase (1: as the plain schedule)
#endif
#define PIPE_WAIT_LGKM0() asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory")
// PAIR3 only: where the LDS-DMAs of a K-tile are issued.  0: all eight between the MFMAs of the two MFMA half-phases (as the plain
// coarse schedule).  1: A-half1 of tile t+1 and A-half0 of tile t+2 (two pieces each) in the READ half-phases of P0 / P1 -- after the
// phase's ds_reads, before its barrier, i.e. by the wave of the SIMD that is NOT in its MFMA half-phase, whose 24 MFMAs take 768
// cycles while a read half-phase needs ~250 -- and only B-half0 / B-half1 of tile t+2 between the MFMAs of P1.  2: all eight in the
// read half-phases.  The early forms wait for their own ds_reads BEFORE the barrier (s_waitcnt lgkmcnt(0)), so that a read has
// completed in its read half-phase: the WAR distance of an early restage is then one full slot for both wave groups
// (tests/test_pipe_schedule_model.py replays all three forms).
#ifndef PIPE_PAIR3_EARLY
#define PIPE_PAIR3_EARLY 0
#endif

// Row of the 256-row operand tile held at row r of half-tile h.
__device__ __forceinline__ int pipe_a_tile_row(int h, int r) { return (((r >> 6) * 2 + h) << 6) + (r & 63); }
__device__ __forceinline__ int pipe_b_tile_row(int h, int r) { return (((r >> 5) * 2 + h) << 5) + (r & 31); }

// Staging geometry: a half-tile is staged as 16 pieces of 1 KiB; wave w issues pieces w and w + 8.
// Piece j of wave w covers half-tile rows (w + 8 j) * 8 + (lane >> 3); lane L fills LDS slot L & 7 of
// its row and must read 16-byte chunk (L & 7) ^ ((row >> 1) & 7) of it (the XOR swizzle lives in the
// source address because the LDS image of an LDS-DMA is lane-linear).
__device__ __forceinline__ int pipe_stage_row(int w, int l, int j) { return (w + 8 * j) * 8 + (l >> 3); }
__device__ __forceinline__ int pipe_stage_chunk(int row, int l) { return ((l & 7) ^ ((row >> 1) & 7)) * 8; }

// Source policy of a plain GEMM: fixed per-lane pointers at k = 0, K-tile t at +64 t halves.
struct PipeSrcFixed {
    const _Float16 *src[4][2];  // [A0 A1 B0 B1][piece]
    int dbg = 0;                // measurement ablation bit 0: always re-read K-tiles 0/1
    template <int TYPE, int J>
    __device__ __forceinline__ const _Float16 *addr(int t) const {
        return src[TYPE][J] + ((dbg & 1) ? (t & 1) * 64 : t * 64);
    }
    template <int TYPE, int J>
    __device__ __forceinline__ void issue(int t, pipe_lds_t *dst) const {
        __builtin_amdgcn_global_load_lds((pipe_glb_t *)addr<TYPE, J>(t), dst, 16, 0, 0);
    }
};

// Source policy of a plain GEMM through buffer descriptors: one descriptor per operand matrix (wave-uniform SGPRs), one
// 32-bit per-lane byte offset per staged piece that never changes, the K offset in an SGPR -- no 64-bit address
// arithmetic next to the MFMAs and 8 address VGPRs instead of 16.  Needs matrices below 4 GiB.
struct PipeSrcDesc {
    __amdgpu_buffer_rsrc_t ra, rb;
    uint32_t voff[4][2];  // [A0 A1 B0 B1][piece]
    template <int TYPE, int J>
    __device__ __forceinline__ void issue(int t, pipe_lds_t *dst) const {
        __builtin_amdgcn_raw_ptr_buffer_load_lds(TYPE < 2 ? ra : rb, dst, 16, voff[TYPE][J], t * 128, 0, 0);
    }
};

// Source policy of the STREAMING (persistent) GEMM: one descriptor per operand MATRIX (wave-uniform), the first row of the
// workgroup's current and next output tile as byte offsets in SGPRs.  K-tile t < NK belongs to the current output tile, K-tile
// t >= NK is K-tile t - NK of the NEXT one (the hand-over of Pipe256T::tiles_streaming).  Needs matrices below 2 GiB.
struct PipeSrcStream {
    __amdgpu_buffer_rsrc_t ra, rb;
    uint32_t voff[4][2];  // [A0 A1 B0 B1][piece]: offset inside the 256-row tile
    int NK;
    uint32_t a_cur, b_cur, a_nxt, b_nxt;  // (tile row) * (row stride) * 2 bytes
    __device__ __forceinline__ uint32_t off_a(int t) const { return t >= NK ? a_nxt + (uint32_t)(t - NK) * 128u : a_cur + (uint32_t)t * 128u; }
    __device__ __forceinline__ uint32_t off_b(int t) const { return t >= NK ? b_nxt + (uint32_t)(t - NK) * 128u : b_cur + (uint32_t)t * 128u; }
    template <int TYPE, int J>
    __device__ __forceinline__ void issue(int t, pipe_lds_t *dst) const {
        __builtin_amdgcn_raw_ptr_buffer_load_lds(TYPE < 2 ? ra : rb, dst, 16, voff[TYPE][J], (int)(TYPE < 2 ? off_a(t) : off_b(t)), 0, 0);
    }
    // The LDS-DMAs of the coarse schedule sit between MFMAs, fenced by sched_barriers: the scalar compare / select / add chain of
    // off_a / off_b in front of each of the eight costs the wave issue time exactly there (measured on the first form of the
    // streaming GEMM: +14 scalar instructions per K-tile, FFN2 -- 96 K-tiles per output tile -- 2-4 % slower than the launch-per-tile
    // kernel).  prepare(t) computes the three offsets a K-tile needs (A-half1 of tile t + 1; A and B of tile t + 2) once, in the read
    // half-phase, and issue_pre picks one by the half-tile's type.
    static constexpr bool PRECOMPUTE = true;
    uint32_t so_a1, so_a2, so_b2;
    __device__ __forceinline__ void prepare(int t) {
        so_a1 = off_a(t + 1);
        so_a2 = off_a(t + 2);
        so_b2 = off_b(t + 2);
    }
    template <int TYPE, int J>
    __device__ __forceinline__ void issue_pre(pipe_lds_t *dst) const {
        __builtin_amdgcn_raw_ptr_buffer_load_lds(TYPE < 2 ? ra : rb, dst, 16, voff[TYPE][J], (int)(TYPE == 1 ? so_a1 : TYPE == 0 ? so_a2 : so_b2), 0, 0);
    }
};

template <class S, class = void>
struct pipe_src_precomputes { static constexpr bool value = false; };
template <class S>
struct pipe_src_precomputes<S, decltype((void)S::PRECOMPUTE)> { static constexpr bool value = S::PRECOMPUTE; };

// SRC provides  template <int TYPE, int J> void issue(int t, pipe_lds_t *dst)  : the LDS-DMA (16 bytes per lane, 1 KiB per
// wave, lane-linear at dst) of piece J of half-tile TYPE (0 A-half0, 1 A-half1, 2 B-half0, 3 B-half1) of K-tile t.
// DBG compiles measurement ablations in (dbg bit 1: no MFMA, bit 3: no staging); product code uses DBG = false.
// KEEP_B0: hold the B-half0 fragments of a K-tile in 16 more VGPRs from phase c0 to c3 instead of reading
// them from LDS a second time (the GEMM has the registers, the search filter does not).
// COARSE (needs KEEP_B0): two phases per K-tile instead of four -- half as many barriers.
//   phase P0: read A-half0, B-half0, B-half1   16 MFMA acc[0][0..1], acc[1][0..1]   + stage A-half1 of tile t+1
//   phase P1: read A-half1                     16 MFMA acc[1][2..3], acc[0][2..3]   + stage A0, B0, B1 of tile t+2
//   RAW  the wait at the end of the P1 reads of tile t-1 (vmcnt(2)) leaves only A-half1 of tile t in flight, so
//        A0 / B0 / B1 of tile t are retired and the following barrier publishes them; the wait at the end of the
//        P0 reads of tile t (vmcnt(6)) leaves only A0 / B0 / B1 of tile t+1 in flight, so A-half1 of tile t is
//        retired before the barrier that precedes its read.
//   WAR  as above: a half-tile last read in phase p is restaged in the MFMA half-phase of phase p+1 at the earliest
//        (A0 / B0 / B1 read in P0 of tile t, restaged in P1 of tile t; A1 read in P1 of tile t, restaged in P0 of t+1).
// PAIR3 (needs COARSE; the split GEMM, round 5): both operands are BLOCKED pair rows (common.h), so the 64 halves of an LDS row are
//   [hi k 0..15 | hi k 16..31 | lo k 0..15 | lo k 16..31]  (fragment index s = 0..3)
// and a K-tile is a 32-deep k-slice of hi AND lo of both operands.  Staging, LDS reads, barriers and waits are those of the
// coarse schedule; only the products change: per output quadrant and k-step j the three MFMAs  hi_j x hi_j,  lo_j x hi_j,
// hi_j x lo_j  (fragment pairs (j, j), (j + 2, j), (j, j + 2)) into the same accumulator -- 24 MFMAs per phase instead of 16,
// i.e. per MFMA two thirds of the LDS-DMAs, ds_reads and barriers of three passes over [hi | lo'] rows (round 4).
// MASKED: per-lane row masks on the fragments -- a lane whose keep_a bit y (A-tile row block y = 0..3 of its wave) / keep_b bit x
// (B-tile row block x = 0, 1) is clear feeds zeros for that row, so the pass adds nothing to it.  Used by the folded GEMMs' second
// pass over the lo halves of the token operand, which must touch the rows with a wide mean and ONLY those (a row's bits must not
// depend on which other rows share its tile: gemm256_f16.hip).
template <class SRC, bool DBG = false, bool KEEP_B0 = false, bool COARSE = false, bool PAIR3 = false, bool MASKED = false>
struct Pipe256T {
    static_assert(!COARSE || KEEP_B0, "the coarse schedule keeps both B halves in registers");
    static_assert(!PAIR3 || COARSE, "the pair products are built on the coarse schedule");
    SRC S;
    _Float16 *smem;
    int w, dbg = 0;
    int ra[2], rb, kx[4];  // per-lane read offsets (halves)
    unsigned keep_a = 0xFu, keep_b = 0x3u;  // MASKED only
    f16x8 fa[2][4], fb[4], fbk[KEEP_B0 ? 4 : 1];

    __device__ __forceinline__ void init(_Float16 *smem_, int w_, int l) {
        smem = smem_;
        w = w_;
        const int g = l >> 5, i = l & 31, wm = w >> 2, wn = w & 3;
        const int c0 = g ^ ((i >> 1) & 7);  // row offsets below are multiples of 16: swizzle depends on i only
#pragma unroll
        for (int s = 0; s < 4; ++s) kx[s] = (c0 ^ (2 * s)) * 8;
        ra[0] = (wm * 64 + i) * 64;
        ra[1] = (wm * 64 + 32 + i) * 64;
        rb = (wn * 32 + i) * 64;
    }

    template <int TYPE, int J, bool PRE = false>
    __device__ __forceinline__ void stage_piece(int t) {
        if (DBG && (dbg & 8)) return;            // ablation: stage nothing (prologue included)
        _Float16 *dst = smem + (t & 1) * PIPE_BUF_HALVES + TYPE * PIPE_HALF_HALVES + (w + 8 * J) * 512;
        if constexpr (PRE && pipe_src_precomputes<SRC>::value) S.template issue_pre<TYPE, J>((pipe_lds_t *)dst);  // offsets of S.prepare(t)
        else S.template issue<TYPE, J>(t, (pipe_lds_t *)dst);
    }
    template <int TYPE>
    __device__ __forceinline__ void stage(int t) {
        stage_piece<TYPE, 0>(t);
        stage_piece<TYPE, 1>(t);
    }
    template <int H>
    __device__ __forceinline__ void read_a(int t) {
        const _Float16 *base = smem + (t & 1) * PIPE_BUF_HALVES + H * PIPE_HALF_HALVES;
#pragma unroll
        for (int s = 0; s < 4; ++s)
#pragma unroll
            for (int yy = 0; yy < 2; ++yy) {
                fa[yy][s] = *reinterpret_cast<const f16x8 *>(base + ra[yy] + kx[s]);
                if constexpr (MASKED)
                    if (!((keep_a >> (2 * H + yy)) & 1u)) fa[yy][s] = f16x8{0, 0, 0, 0, 0, 0, 0, 0};
            }
    }
    template <int H>
    __device__ __forceinline__ void read_b(int t) {
        const _Float16 *base = smem + (t & 1) * PIPE_BUF_HALVES + (2 + H) * PIPE_HALF_HALVES;
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            f16x8 v = *reinterpret_cast<const f16x8 *>(base + rb + kx[s]);
            if constexpr (MASKED)
                if (!((keep_b >> H) & 1u)) v = f16x8{0, 0, 0, 0, 0, 0, 0, 0};
            if constexpr (KEEP_B0 && H == 0) fbk[s] = v;
            else fb[s] = v;
        }
    }
    // MFMA half-phase: barrier, 8 MFMAs with the two LDS-DMA pieces of half-tile STAGE (of K-tile ts)
    // issued in the shadow of the matrix pipe (an LDS-DMA costs 60-185 issue cycles in a read
    // half-phase, which is the one with no slack), barrier.  STAGE < 0: nothing to stage.
    template <int X, int YH, int STAGE>
    __device__ __forceinline__ void mfma(f32x16 (&acc)[2][4], int ts) {
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        // (no s_setprio around the MFMAs: 1.31-1.34 M cycles per XCD on 8192^3 without it, 1.36-1.38 M with it)
#pragma unroll
        for (int s = 0; s < 4; ++s) {
#pragma unroll
            for (int yy = 0; yy < 2; ++yy) {
                const f16x8 &bf = (KEEP_B0 && X == 0) ? fbk[s] : fb[s];
                if (DBG && (dbg & 2)) {  // ablation: keep the LDS reads alive, skip the matrix pipe
                    asm volatile("" ::"v"(bf), "v"(fa[yy][s]));
                } else {
                    acc[X][2 * YH + yy] = __builtin_amdgcn_mfma_f32_32x32x16_f16(bf, fa[yy][s], acc[X][2 * YH + yy], 0, 0, 0);
                }
            }
            if constexpr (STAGE >= 0) {
                if (s == 0 || s == 2) {
                    __builtin_amdgcn_sched_barrier(0);
                    if (s == 0) stage_piece<(STAGE >= 0 ? STAGE : 0), 0>(ts);
                    else stage_piece<(STAGE >= 0 ? STAGE : 0), 1>(ts);
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
        }

        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
    }

    // MODE 0: steady state (tiles t+1, t+2 exist); 1: t = NK-2; 2: t = NK-1
    template <int MODE>
    __device__ __forceinline__ void tile(int t, f32x16 (&acc)[2][4]) {
        // c0
        read_a<0>(t);
        read_b<0>(t);
        mfma<0, 0, (MODE <= 1 ? 2 : -1)>(acc, t + 1);
        // c1
        read_b<1>(t);
        mfma<1, 0, (MODE == 0 ? 0 : -1)>(acc, t + 2);
        // c2
        read_a<1>(t);
        mfma<1, 1, (MODE == 0 ? 3 : -1)>(acc, t + 2);
        // c3: the wait retires every half-tile of tile t+1 (see RAW above)
        if constexpr (!KEEP_B0) read_b<0>(t);
        if constexpr (MODE == 0) PIPE_WAIT_VM(4);
        if constexpr (MODE == 1) PIPE_WAIT_VM(0);
        mfma<0, 1, (MODE == 0 ? 1 : -1)>(acc, t + 2);
    }

    // ---- coarse schedule ------------------------------------------------------------------------------
    // MFMA half-phase of the coarse schedule: 16 MFMAs on A fragments fa (A-half YH) against both B halves, with
    // N_STAGE LDS-DMA pieces (two per listed half-tile of K-tile ts) issued after every second MFMA.
    template <int YH, int ST0, int ST1, int ST2>
    __device__ __forceinline__ void mfma16(f32x16 (&acc)[2][4], int ts) {
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        constexpr int n_stage = (ST0 >= 0) + (ST1 >= 0) + (ST2 >= 0);
        constexpr int types[3] = {ST0 >= 0 ? ST0 : 0, ST1 >= 0 ? ST1 : 0, ST2 >= 0 ? ST2 : 0};
        // steps of two MFMAs (the two 32-row A fragments of this phase against one B fragment); PER = steps per B half
        constexpr int PER = PAIR3 ? 6 : 4;
        // PAIR3: (A fragment, B fragment) of step c -- hi_j x hi_j, lo_j x hi_j, hi_j x lo_j for j = 0, 1
        constexpr int pa[6] = {0, 2, 0, 1, 3, 1}, pb[6] = {0, 0, 2, 1, 1, 3};
#pragma unroll
        for (int step = 0; step < 2 * PER; ++step) {
            // P0 runs B-half0 first (it was read first); P1 runs B-half1 first (either order is fine for the result:
            // the two halves accumulate into different registers)
            // (PAIR3 with the two B halves alternating -- an accumulator every fourth MFMA instead of every second -- measures the
            // same: profiles/r05_ab_pair3_mfma_order.jsonl)
            const int xx = (step / PER) ^ YH, c = step % PER;
            const int sa = PAIR3 ? pa[c] : c, sb = PAIR3 ? pb[c] : c;
#pragma unroll
            for (int yy = 0; yy < 2; ++yy) {
                const f16x8 &bf = xx == 0 ? fbk[sb] : fb[sb];
                if (DBG && (dbg & 2)) {
                    asm volatile("" ::"v"(bf), "v"(fa[yy][sa]));
                } else {
                    acc[xx][2 * YH + yy] = __builtin_amdgcn_mfma_f32_32x32x16_f16(bf, fa[yy][sa], acc[xx][2 * YH + yy], 0, 0, 0);
                }
            }
            // one LDS-DMA piece after each of the first 2 * n_stage MFMA pairs (PAIR3, STAGE_GAP = 2: after every second pair --
            // the phase is half as long again, the pieces keep their distance in MFMA time)
            constexpr int GAP = PAIR3 ? PIPE_PAIR3_STAGE_GAP : 1;
            if (step % GAP == 0 && step / GAP < 2 * n_stage) {
                const int pc = step / GAP;
                __builtin_amdgcn_sched_barrier(0);
                if (pc == 0) stage_piece<types[0], 0, true>(ts);
                if (pc == 1) stage_piece<types[0], 1, true>(ts);
                if (pc == 2) stage_piece<types[1], 0, true>(ts);
                if (pc == 3) stage_piece<types[1], 1, true>(ts);
                if (pc == 4) stage_piece<types[2], 0, true>(ts);
                if (pc == 5) stage_piece<types[2], 1, true>(ts);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
    }

    // VM0 / VM1: the P0 / P1 waits of MODE 0 / 1 -- 6 and 2 in steady state (P0 leaves A0 B0 B1 of tile t+1 in flight, P1 A-half1 of
    // tile t+1).  The FIRST K-tile of an output tile that was prefetched under an epilogue (tiles_streaming hand-over) has that
    // epilogue's loads / stores and the parameter block's LDS-DMAs in the counter as well, YOUNGER than the half-tiles these two
    // waits are for (A-half1 of tile t; A0 B0 B1 of tile t+1 -- all issued before the epilogue): the kernel may add the number of
    // vector-memory operations EVERY wave is sure to have issued in between, capped at the counter's 63.  Larger counts only avoid
    // waiting for the epilogue's stores to drain (vmcnt retires in order); from P0 of tile t+1 on the waits are the steady-state
    // ones -- A-half1 of tile t+1 is younger than those stores.
    template <int MODE, int VM0 = 6, int VM1 = 2>
    __device__ __forceinline__ void tile2(int t, f32x16 (&acc)[2][4]) {
        if constexpr (pipe_src_precomputes<SRC>::value) S.prepare(t);
        if constexpr (PAIR3 && PIPE_PAIR3_EARLY > 0) {
            // in flight when a tile starts (oldest first): A1(t), A0 B0 B1 (t+1) -- what the prologue leaves, too
            // P0
            read_a<0>(t);
            read_b<0>(t);
            read_b<1>(t);
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (MODE <= 1) {
                stage<1>(t + 1);
                PIPE_WAIT_VM(8);  // retires A1(t); leaves A0 B0 B1 A1 of tile t+1
            } else {
                PIPE_WAIT_VM(0);
            }
            PIPE_WAIT_LGKM0();
            mfma16<0, -1, -1, -1>(acc, t + 1);
            // P1
            read_a<1>(t);
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (MODE == 0) {
                stage<0>(t + 2);
                if constexpr (PIPE_PAIR3_EARLY == 2) {
                    stage<2>(t + 2);
                    stage<3>(t + 2);
                    PIPE_WAIT_VM(8);  // retires A0 B0 B1 of tile t+1; leaves A1(t+1) and A0 B0 B1 of tile t+2
                } else {
                    PIPE_WAIT_VM(4);  // retires A0 B0 B1 of tile t+1; leaves A1(t+1), A0(t+2)
                }
            } else if constexpr (MODE == 1) {
                PIPE_WAIT_VM(2);      // retires A0 B0 B1 of tile t+1; leaves A1(t+1)
            }
            PIPE_WAIT_LGKM0();
            if constexpr (MODE == 0 && PIPE_PAIR3_EARLY == 1) mfma16<1, 2, 3, -1>(acc, t + 2);
            else mfma16<1, -1, -1, -1>(acc, t + 2);
            return;
        }
        // P0
        read_a<0>(t);
        read_b<0>(t);
        read_b<1>(t);
        if constexpr (MODE <= 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(VM0) : "memory"); else PIPE_WAIT_VM(0);
        if constexpr (MODE <= 1) mfma16<0, 1, -1, -1>(acc, t + 1); else mfma16<0, -1, -1, -1>(acc, t + 1);
        // P1
        read_a<1>(t);
        if constexpr (MODE <= 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(VM1) : "memory");
        if constexpr (MODE == 0) mfma16<1, 0, 2, 3>(acc, t + 2); else mfma16<1, -1, -1, -1>(acc, t + 2);
    }

    // ---- building blocks of a K loop (all 512 threads) ----------------------------------------------
    // prologue: stage K-tile 0 and A0 B1 A1 of K-tile 1 (what the steady state has issued when a tile
    // starts), publish tile 0
    __device__ __forceinline__ void prologue() {
        if constexpr (COARSE) {
            // tile 0 complete except its A-half1 (retired by the first P0 wait), then A0 B0 B1 of tile 1
            stage<0>(0); stage<2>(0); stage<3>(0); stage<1>(0);
            PIPE_WAIT_VM(2);
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
            stage<0>(1); stage<2>(1); stage<3>(1);
        } else {
            stage<0>(0); stage<2>(0); stage<3>(0); stage<1>(0); stage<0>(1); stage<3>(1); stage<1>(1);
            PIPE_WAIT_VM(6);
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    // coarse schedule: the prologue's half-tiles (K-tile 0, A0 B0 B1 of K-tile 1) with ALL of them retired and published -- what a
    // K-tile 0 with loose waits (tile2: VM0 / VM1) needs when no epilogue preceded it
    __device__ __forceinline__ void prologue_landed() {
        static_assert(COARSE, "coarse schedule only");
        stage<0>(0); stage<2>(0); stage<3>(0); stage<1>(0);
        stage<0>(1); stage<2>(1); stage<3>(1);
        PIPE_WAIT_VM(0);
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
    }
    // enter / leave the staggered section: wm = 1 runs one barrier behind wm = 0 in between
    __device__ __forceinline__ void enter() {
        __builtin_amdgcn_sched_barrier(0);
        if (w >= 4) __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
    }
    __device__ __forceinline__ void leave() {
        __builtin_amdgcn_sched_barrier(0);
        if (w < 4) __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
    }
    // K-tiles 0..NK-1 of a stream that CONTINUES (the source policy maps t >= NK onto what follows):
    // on return K-tiles NK and NK+1 are staged exactly as the prologue leaves tiles 0 and 1 (the wait
    // of the last c3 has retired tile NK).
    // VM0L / VM1L (coarse schedule, != 6 / 2): K-tile 0 was prefetched under an epilogue and waits with these counts (tile2).  The
    // FIRST output tile of a persistent workgroup then starts from prologue_landed (every LDS-DMA of the prologue retired: a loose
    // wait has nothing to wait for), so that K-tile 0 is the same code for every output tile.
    template <int VM0L = 6, int VM1L = 2>
    __device__ __forceinline__ void tiles_streaming(int NK, f32x16 (&acc)[2][4]) {
        if constexpr (COARSE && (VM0L != 6 || VM1L != 2)) {
            tile2<0, VM0L, VM1L>(0, acc);
            for (int t = 1; t < NK; ++t) tile2<0>(t, acc);
        } else {
            for (int t = 0; t < NK; ++t) {
                if constexpr (COARSE) tile2<0>(t, acc); else tile<0>(t, acc);
            }
        }
    }
    // K-tiles T0..NK-1 of a stream that ENDS (T0 even: the buffer parity of a tile is t & 1; NK - T0 >= 2): nothing beyond
    // NK-1 is staged, no LDS-DMA left in flight.
    // (VM0L / VM1L as tiles_streaming; they need NK - T0 >= 3: K-tile T0 is then a steady-state tile)
    template <int VM0L = 6, int VM1L = 2>
    __device__ __forceinline__ void tiles_final(int NK, f32x16 (&acc)[2][4], int T0 = 0) {
        if constexpr (COARSE && (VM0L != 6 || VM1L != 2)) {
            tile2<0, VM0L, VM1L>(T0, acc);
            for (int t = T0 + 1; t < NK - 2; ++t) tile2<0>(t, acc);
            tile2<1>(NK - 2, acc);
            tile2<2>(NK - 1, acc);
        } else if constexpr (COARSE) {
            for (int t = T0; t < NK - 2; ++t) tile2<0>(t, acc);
            tile2<1>(NK - 2, acc);
            tile2<2>(NK - 1, acc);
        } else {
            for (int t = T0; t < NK - 2; ++t) tile<0>(t, acc);
            tile<1>(NK - 2, acc);
            tile<2>(NK - 1, acc);
        }
    }

    // Whole K loop of one output tile.  On return every wave has passed the same number of barriers.
    __device__ __forceinline__ void run(int NK, f32x16 (&acc)[2][4]) {
        prologue();
        enter();
        tiles_final(NK, acc);
        leave();
    }
};

}  // namespace ance
