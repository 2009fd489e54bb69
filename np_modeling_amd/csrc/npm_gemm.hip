// fp32 GEMM on the exact-fp32 MFMA of gfx950 (v_mfma_f32_32x32x2_f32).
//
// Replaces np.matmul (reference layers/mlp.py:23,35,36) and every np.einsum
// contraction of layers/attentions.py (flat or batched GEMM views).
//
// Shape of the kernel (CDNA4-first, see DESIGN.md "GEMM"):
//   * 256 threads = 4 wavefronts (64 lanes) arranged 2 (M) x 2 (N); block tile 128x128,
//     K step 32; each wave owns 64x64 = 2x2 MFMA tiles of 32x32 (64 accumulator VGPRs).
//   * The f32 MFMA takes ONE f32 per lane per operand: lane l supplies A[i=l&31][k=l>>5]
//     and B[k=l>>5][j=l&31].  Inside each group of 8 k values the lane half h = l>>5
//     walks k = 8g + 4h + s (s = 0..3), the same permutation for A and B, so a K-major
//     LDS image is read with one ds_read_b128 per 4 MFMA steps.
//   * Operands whose K index is contiguous in memory ("K-major": A of NN/NT, B of NT) are
//     staged as [row][k] with a 36-float pitch (conflict-free b128 reads); operands whose
//     M/N index is contiguous (A of TN, B of NN/TN) are staged as [k][row] and read with
//     ds_read_b32 (32 consecutive floats per half-wave: conflict-free).
//   * Global -> register -> LDS staging: the next K tile's global loads are issued before
//     the 64 MFMAs of the current tile and written to LDS after them; >= 2 blocks per CU
//     cover the barriers.  The f32 MFMA is 1/16 of the bf16 rate, so a 128^2 tile needs
//     only ~8 B/clk/CU of operand traffic: the pipe to keep busy is the matrix core.
//   * blockIdx is remapped so that each XCD (private L2) owns a contiguous run of tiles,
//     rasterised in groups of tile rows so neighbouring blocks share A and B panels.
//   * Split-K (tall-K weight-gradient GEMMs) writes fp32 slabs and a second kernel sums
//     them in a fixed order: bitwise reproducible, no float atomics.
#include <algorithm>

#include "npm_mfma_tile.h"

namespace {

using namespace npm_tile;

struct GemmArgs {
    const float *A, *B;
    long lda, ldb;
    long sA0, sA1, sB0, sB1, sC0, sC1;
    int M, N, K;
    int batch1;
    int tiles_m, tiles_n, splits, k_per_split;
    int group_m;
    float *ksum;     // optional [splits][N or M] partial sums over k of B (ksum_op 1) or A (2): TN layout, batch 1
    int ksum_op;
    int math;        // 0 exact-f32 MFMA, 1 three-way bf16 split on the bf16 MFMA (NPM_TUNE_GEMM_MATH)
    long long *trace; // diagnostics: 8 words per block (hw id, xcc id, 4 s_memtime stamps) or null
    int wide;        // 128 x 256 block tile (2 x 4 waves) instead of 128 x 128
    int ablate;      // TIMING-ONLY diagnostics (results are wrong): 1 skip operand loads, 2 skip LDS stores, 4 skip barriers, 8 skip the epilogue
    long slab;       // split-K: batch * M * N
    Epilogue e;
};

int g_pipe = 2;       // tuning knobs (npm_set_tuning); 2 = LDS-DMA pipeline where eligible
int g_group_m = 8;
int g_split_gens = 1;             // NPM_TUNE_GEMM_SPLIT_GENS
int g_ablate = 0;
int g_wave_prio = 0;               // NPM_TUNE_GEMM_WAVE_PRIO
int g_math = 0;                    // NPM_TUNE_GEMM_MATH
int g_wide_tile = 0;               // NPM_TUNE_GEMM_WIDE_TILE: 128 x 256 tile where n % 256 == 0: 0 never, 1 always, 2 NN/NT, 3 NT only
long long *g_trace = nullptr;    // diagnostics: per-block timeline stamps (npm_debug_gemm_trace)
int g_buf_epilogue = 1;

// PIPE 0: one LDS buffer, two barriers per K tile (36 KB LDS, 3 blocks/CU).
// PIPE 1: two LDS buffers, ONE barrier per K tile: tile t+1 is written into the other buffer
//         between the two halves of tile t's MFMAs (72 KB LDS, 2 blocks/CU).
template <bool A_KMAJ, bool B_KMAJ, bool VEC, int PIPE>
__global__ void __launch_bounds__(NTHREADS)
sgemm_mfma_kernel(const GemmArgs p) {
    __shared__ __attribute__((aligned(16))) float smem[(PIPE ? 4 : 2) * TILE_FLOATS];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int l32 = lane & 31, half = lane >> 5;

    // block -> (batch z, split, tile): tiles fastest so that neighbours share operand panels
    const int logical = xcd_remap(blockIdx.x, gridDim.x);
    const int tiles = p.tiles_m * p.tiles_n;
    const int t = logical % tiles;
    const int rest = logical / tiles;
    const int split = rest % p.splits;
    const int z = rest / p.splits;
    int tm, tn;
    tile_coords(t, p.tiles_m, p.tiles_n, p.group_m, tm, tn);
    const int m0 = tm * BM, n0 = tn * BN;
    const int z0 = z / p.batch1, z1 = z - z0 * p.batch1;
    const float *A = p.A + z0 * p.sA0 + z1 * p.sA1;
    const float *B = p.B + z0 * p.sB0 + z1 * p.sB1;

    const int kbeg = split * p.k_per_split;
    const int kend = min(p.K, kbeg + p.k_per_split);
    const int nkt = (kend - kbeg + BK - 1) / BK;

    f32x16 acc[2][2];
    zero_acc(acc);

    float4 ra[4], rb[4];
    load_tile<A_KMAJ, VEC>(A, p.lda, m0, p.M, kbeg, kend, tid, ra);
    load_tile<B_KMAJ, VEC>(B, p.ldb, n0, p.N, kbeg, kend, tid, rb);
    store_tile<A_KMAJ>(smem, tid, ra);
    store_tile<B_KMAJ>(smem + TILE_FLOATS, tid, rb);
    __syncthreads();

    const int arow = wm * 64 + l32;
    const int brow = wn * 64 + l32;

    for (int kt = 0; kt < nkt; ++kt) {
        const bool more = kt + 1 < nkt;
        if (more && !(p.ablate & 1)) {   // next tile's global loads fly under this tile's MFMAs
            const int k0 = kbeg + (kt + 1) * BK;
            load_tile<A_KMAJ, VEC>(A, p.lda, m0, p.M, k0, kend, tid, ra);
            load_tile<B_KMAJ, VEC>(B, p.ldb, n0, p.N, k0, kend, tid, rb);
        }
        if (PIPE == 0) {
            mma_tile<A_KMAJ, B_KMAJ>(smem, smem + TILE_FLOATS, arow, brow, half, acc);
            if (!(p.ablate & 4)) __syncthreads();
            if (more && !(p.ablate & 2)) {
                store_tile<A_KMAJ>(smem, tid, ra);
                store_tile<B_KMAJ>(smem + TILE_FLOATS, tid, rb);
            }
            if (!(p.ablate & 4)) __syncthreads();
        } else {
            const float *cur = smem + (kt & 1) * 2 * TILE_FLOATS;
            float *nxt = smem + ((kt + 1) & 1) * 2 * TILE_FLOATS;
            mma_groups<A_KMAJ, B_KMAJ, 0, 2>(cur, cur + TILE_FLOATS, arow, brow, half, acc);
            if (more) {   // every wave passed the barrier that ended tile kt-1: nobody reads `nxt` any more
                store_tile<A_KMAJ>(nxt, tid, ra);
                store_tile<B_KMAJ>(nxt + TILE_FLOATS, tid, rb);
            }
            mma_groups<A_KMAJ, B_KMAJ, 2, 4>(cur, cur + TILE_FLOATS, arow, brow, half, acc);
            __syncthreads();
        }
    }

    Epilogue e = p.e;
    if (p.splits > 1) {
        e.ws += (long)split * p.slab + (long)z * p.M * p.N;
        write_tile(acc, e, true, m0, n0, p.M, p.N, wm, wn, l32, half);
        return;
    }
    const long coff = z0 * p.sC0 + z1 * p.sC1;
    e.C += coff;
    if (e.R) e.R += coff;
    if (e.aux) e.aux += coff;
    if (e.rowvec) e.rowvec += (long)z * p.M;
    write_tile(acc, e, false, m0, n0, p.M, p.N, wm, wn, l32, half);
}

// ------------------------------------------------------------------------------------------
// PIPE 2: operands go global -> LDS by LDS-DMA (buffer_load_dwordx4 ... lds): no staging
// VGPRs, no ds_write pass, no per-load bounds branches (the buffer descriptor's range check
// returns 0 beyond the operand; rows beyond M / N only feed outputs that are never stored).
// K step 16, two LDS stages of 16 KB, ONE barrier per K tile; 128 registers -> 4 blocks per CU.
// The LDS image of a DMA is lane-linear, so the bank-conflict swizzle of the K-major tiles
// ([128 rows][64 B]) is applied to the SOURCE address: LDS chunk c' of row r holds global chunk
// c' ^ ((r >> 2) & 3); reads apply the same XOR (conflict-free ds_read_b128).
// Needs 16-byte aligned operands and K (and every split) a multiple of 16.
// ------------------------------------------------------------------------------------------
// WM x WN wavefronts per block, each owning 64 x 64 of the (64 WM) x (64 WN) block tile.
//   2 x 2 (256 threads, 4 blocks/CU): the default.   2 x 4 (512 threads, 2 blocks/CU): 128 x 256 tile --
//   each activation panel is re-read by half as many column tiles (less L2-miss traffic per FLOP).
// KSUM (TN layout only): 1 = also sum the B tiles over k (the column sums of B [K, N]), 2 = the A tiles (the
// column sums of A [K, M]) -- the bias gradient that goes with a weight gradient x^T dy (mlp.py:34-35) or
// dproj^T x (attentions.py:167-188), instead of a separate pass over dy.  The tiles_m (tiles_n) blocks that stage
// the same B (A) tile split its 16 k rows between them, so every block carries the same small load (a launch of
// co-resident blocks lasts as long as its slowest block): one or two LDS reads and adds per thread and K tile.
// Partial sums: one row per (split, sharing block); the host adds the rows.
template <bool A_KMAJ, bool B_KMAJ, bool WITH_COLSUM, int WM = 2, int WN = 2, int KSUM = 0, int MATH = 0, bool ROWDOT = false>
__global__ void __launch_bounds__(64 * WM * WN, (MATH == 2 ? 8 : (WITH_COLSUM || MATH) ? 12 : 16) / (WM * WN))   // 16 (12, 8) waves per CU
sgemm_glds_kernel(const GemmArgs p) {
    static_assert(!ROWDOT || (WM == 2 && WN == 2 && !WITH_COLSUM && !KSUM && !MATH), "ROWDOT: the plain 128 x 128 exact-fp32 instance");
    static_assert(!KSUM || (!A_KMAJ && !B_KMAJ && WM == 2 && WN == 2 && !WITH_COLSUM), "KSUM: TN layout, 128 x 128 tile");
    constexpr int TM = 64 * WM, TN = 64 * WN;
    constexpr int A_TILE = TM * GK, B_TILE = TN * GK, STAGE = A_TILE + B_TILE;
    constexpr int A_PW = 4 / WN, B_PW = 4 / WM;          // 1 KiB DMA pieces per wave and K tile
    static_assert(A_PW * WM * WN * 16 == TM && B_PW * WM * WN * 16 == TN, "pieces must tile the operands");
    // MATH 2 runs two blocks per CU (200 registers): three stages, tile kt + 2 in flight while kt is consumed, a
    // counted vmcnt and a bare s_barrier so that the newest DMA stays in flight (measured +1..2 % in the C5 step;
    // no gain for the modes that keep three or four blocks per CU).
#ifndef NPM_MATH_STAGES
#define NPM_MATH_STAGES 3
#endif
    constexpr int NSTAGE = MATH == 2 ? NPM_MATH_STAGES : 2;
    __shared__ __attribute__((aligned(16))) float smem[NSTAGE * STAGE];

    prio_high(p.e.prio & 1);
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN, wn = wave % WN;
    const int l32 = lane & 31, half = lane >> 5;

    const int logical = xcd_remap(blockIdx.x, gridDim.x);
    const int tiles = p.tiles_m * p.tiles_n;
    const int t = logical % tiles;
    const int rest = logical / tiles;
    const int split = rest % p.splits;
    const int z = rest / p.splits;
    int tm, tn;
    tile_coords(t, p.tiles_m, p.tiles_n, p.group_m, tm, tn);
    const int m0 = tm * TM, n0 = tn * TN;
    const int z0 = z / p.batch1, z1 = z - z0 * p.batch1;

    const int kbeg = split * p.k_per_split;
    const int kend = min(p.K, kbeg + p.k_per_split);
    const int nkt = (kend - kbeg) / GK;

    // Buffer descriptors rebased to this block's panel; num_records = bytes up to the end of the
    // operand's valid extent, so every access beyond the matrix reads 0 instead of faulting.
    const long a_batch = z0 * p.sA0 + z1 * p.sA1, b_batch = z0 * p.sB0 + z1 * p.sB1;
    const long a_extent = A_KMAJ ? (long)(p.M - 1) * p.lda + p.K : (long)(p.K - 1) * p.lda + p.M;
    const long b_extent = B_KMAJ ? (long)(p.N - 1) * p.ldb + p.K : (long)(p.K - 1) * p.ldb + p.N;
    const long a_panel = A_KMAJ ? (long)m0 * p.lda + kbeg : (long)kbeg * p.lda + m0;
    const long b_panel = B_KMAJ ? (long)n0 * p.ldb + kbeg : (long)kbeg * p.ldb + n0;
    const long a_left = max(a_extent - a_panel, 0L) * 4, b_left = max(b_extent - b_panel, 0L) * 4;
    const auto rsrcA = __builtin_amdgcn_make_buffer_rsrc((void *)(p.A + a_batch + a_panel), 0,
                                                         (int)min(a_left, 0xFFFFFFFFL), 0x00020000);
    const auto rsrcB = __builtin_amdgcn_make_buffer_rsrc((void *)(p.B + b_batch + b_panel), 0,
                                                         (int)min(b_left, 0xFFFFFFFFL), 0x00020000);
    // this wave's DMA pieces of both operand tiles
    unsigned va[A_PW], vb[B_PW];
#pragma unroll
    for (int i = 0; i < A_PW; ++i) va[i] = glds_voffset<A_KMAJ, TM>(lane, A_PW * wave + i, p.lda);
#pragma unroll
    for (int i = 0; i < B_PW; ++i) vb[i] = glds_voffset<B_KMAJ, TN>(lane, B_PW * wave + i, p.ldb);
    const unsigned a_kstep = A_KMAJ ? GK * 4u : (unsigned)(GK * p.lda * 4);
    const unsigned b_kstep = B_KMAJ ? GK * 4u : (unsigned)(GK * p.ldb * 4);

    auto issue = [&](int kt, int stage) {
        float *sa = smem + stage * STAGE + (A_PW * wave) * 256;        // 1 KiB per piece
        float *sb = smem + stage * STAGE + A_TILE + (B_PW * wave) * 256;
        const unsigned ka = kt * a_kstep, kb = kt * b_kstep;
#pragma unroll
        for (int i = 0; i < A_PW; ++i) lds_dma16(rsrcA, sa + 256 * i, va[i], ka);
#pragma unroll
        for (int i = 0; i < B_PW; ++i) lds_dma16(rsrcB, sb + 256 * i, vb[i], kb);
    };

    f32x16 acc[2][2];
    zero_acc(acc);
    f32x16 small[MATH == 2 ? 2 : 1][MATH == 2 ? 2 : 1];      // MATH 2: the split's small terms, added once at the end
    if (MATH == 2) zero_acc(reinterpret_cast<f32x16 (&)[2][2]>(small));
    const int arow = wm * 64 + l32;
    const int brow = wn * 64 + l32;
    const int ks_idx = KSUM == 1 ? tm : tn;                       // which of the blocks sharing this operand tile
    const int ks_share = KSUM == 1 ? p.tiles_m : p.tiles_n;
    const bool do_ksum = KSUM && z == 0 && ks_idx < GK;
    const int ks_row0 = ks_idx + (wave >> 1) * ks_share;          // this thread's k rows: ks_row0 + 2 j ks_share
    float ks_acc = 0.f;                                           // column tid & 127

    long long t_start = 0, t_first = 0, t_loop = 0, r_start = 0;
    if (p.trace) { t_start = __builtin_amdgcn_s_memtime(); r_start = __builtin_amdgcn_s_memrealtime(); }
    if (nkt > 0) issue(0, 0);
    if (NSTAGE == 3 && nkt > 1) issue(1, 1);
    prio_low(p.e.prio & 1);
    // one K tile out of stage STG: tile KT has landed (every wave's pieces) and everybody is done reading the stage about
    // to be refilled; the next tile goes out, then the 32 MFMAs
#define NPM_GEMM_TILE(KT, STG)                                                                                       \
    do {                                                                                                             \
        if (!(p.ablate & 4)) dma_barrier();                                                                          \
        if (p.trace && (KT) == 0) t_first = __builtin_amdgcn_s_memtime();                                            \
        if ((KT) + 1 < nkt && !(p.ablate & 1)) issue((KT) + 1, (STG) ^ 1);                                           \
        const float *sA = smem + (STG) * STAGE;                                                                      \
        const float *sB = sA + A_TILE;                                                                               \
        if (KSUM && do_ksum) {                                                                                       \
            const float *st = (KSUM == 1 ? sB : sA) + (tid & 127);          /* both tiles are [16 k][128] here */    \
            for (int r = ks_row0; r < GK; r += 2 * ks_share) ks_acc += st[r * 128];                                  \
        }                                                                                                            \
        mma_tile16_math<MATH, A_KMAJ, B_KMAJ, TN, TM>(sA, sB, arow, brow, half, acc, reinterpret_cast<f32x16 (&)[2][2]>(small)); \
    } while (0)
    if (NSTAGE == 2) {
        // unrolled by two: the stage is a compile-time constant, so every LDS address of the loop is a loop-invariant
        // register plus an immediate -- no vector-ALU instruction between the MFMAs (each one costs the matrix pipe about
        // six cycles, tools/microbench/mfma_f32_16x16.hip; LDS reads, scalar instructions and waits cost nothing)
        for (int kt = 0; kt < nkt; kt += 2) {
            NPM_GEMM_TILE(kt, 0);
            if (kt + 1 < nkt) NPM_GEMM_TILE(kt + 1, 1);
        }
    } else {
        int stage = 0;
        for (int kt = 0; kt < nkt; ++kt) {
            if (kt + 1 < nkt) asm volatile("s_waitcnt vmcnt(%0) ; npm:wait" ::"n"(A_PW + B_PW) : "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            if (kt + 2 < nkt) issue(kt + 2, stage == 0 ? 2 : stage - 1);
            const float *sA = smem + stage * STAGE;
            const float *sB = sA + A_TILE;
            stage = stage + 1 == NSTAGE ? 0 : stage + 1;
            if (KSUM && do_ksum) {
                const float *st = (KSUM == 1 ? sB : sA) + (tid & 127);
                for (int r = ks_row0; r < GK; r += 2 * ks_share) ks_acc += st[r * 128];
            }
            mma_tile16_math<MATH, A_KMAJ, B_KMAJ, TN, TM>(sA, sB, arow, brow, half, acc, reinterpret_cast<f32x16 (&)[2][2]>(small));
        }
    }
#undef NPM_GEMM_TILE
    if (MATH == 2) {
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) acc[i][j] += reinterpret_cast<f32x16 (&)[2][2]>(small)[i][j];
    }
    if (KSUM && do_ksum) {       // the two thread halves -> one sum per column; the stage buffers are free after a barrier
        __syncthreads();
        smem[tid] = ks_acc;
        __syncthreads();
        if (tid < 128) {
            const int extent = KSUM == 1 ? p.N : p.M, at = (KSUM == 1 ? n0 : m0) + tid;
            const int rows = min(ks_share, GK);
            if (at < extent) p.ksum[((long)split * rows + ks_idx) * extent + at] = smem[tid] + smem[tid + 128];
        }
    }

    if (p.trace) t_loop = __builtin_amdgcn_s_memtime();
    prio_high(p.e.prio & 2);
    struct TraceOnExit {
        long long *buf, t0, t1, t2, r0;
        int tid;
        __device__ __forceinline__ ~TraceOnExit() {
            if (buf && tid == 0) {
                const long long t_issued = __builtin_amdgcn_s_memtime();
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // stores of this wave have left
                long long *w = buf + (long)blockIdx.x * 8;
                w[6] = t_issued;
                w[7] = __builtin_amdgcn_s_memrealtime() - r0;      // 100 MHz ticks over the block's lifetime
                w[0] = __builtin_amdgcn_s_getreg((31 << 11) | 4);        // HW_REG_HW_ID
                w[1] = __builtin_amdgcn_s_getreg((31 << 11) | 20);       // HW_REG_XCC_ID
                w[2] = t0; w[3] = t1; w[4] = t2; w[5] = __builtin_amdgcn_s_memtime();
            }
        }
    } trace_guard{p.trace, t_start, t_first, t_loop, r_start, tid};
    if (p.ablate & 8) {          // timing only: one store per lane keeps the accumulators alive
        if (acc[0][0][0] + acc[0][1][5] + acc[1][0][9] + acc[1][1][15] == 12345.f) p.e.C[tid] = 0.f;
        return;
    }
    Epilogue e = p.e;
    if (!ROWDOT && p.splits > 1) {
        e.ws += (long)split * p.slab + (long)z * p.M * p.N;
        if (e.buf_ok) write_tile_buf(acc, e, true, m0, n0, p.M, p.N, wm, wn, l32, half);
        else write_tile(acc, e, true, m0, n0, p.M, p.N, wm, wn, l32, half);
        return;
    }
    const long coff = z0 * p.sC0 + z1 * p.sC1;
    e.C += coff;
    if (e.R) e.R += coff;
    if (e.aux) e.aux += coff;
    if (e.rowvec) e.rowvec += (long)z * p.M;
    if (WITH_COLSUM) {   // partial rows laid out [z0][tile row][wave row][z1][N]
        const int nb1 = p.batch1;
        e.cs += (((long)(z0 * p.tiles_m + tm) * 2) * nb1 + z1) * p.N;
        e.cs_wm = (long)nb1 * p.N;
        write_tile_buf<true>(acc, e, false, m0, n0, p.M, p.N, wm, wn, l32, half);
        return;
    }
    if (ROWDOT) {                // NPM_EPI_ROWDOT (the host admits it only where the buffer epilogue runs)
        write_tile_buf<false, true>(acc, e, false, m0, n0, p.M, p.N, wm, wn, l32, half);
        return;
    }
    if (e.buf_ok) write_tile_buf(acc, e, false, m0, n0, p.M, p.N, wm, wn, l32, half);
    else write_tile(acc, e, false, m0, n0, p.M, p.N, wm, wn, l32, half);
}

inline bool aligned16(const void *ptr) { return ((uintptr_t)ptr & 15) == 0; }

template <bool A_KMAJ, bool B_KMAJ>
void launch(const GemmArgs &a, bool vec, bool dma, int grid, hipStream_t stream) {
    // The split-bf16 modes exist on the LDS-DMA pipeline with the 128 x 128 tile (with or without the fused k-sums);
    // the wide tile, the epilogue column sums and the register-staged fallback (unaligned operands, K % 16 != 0) run
    // the exact-f32 MFMA whatever was requested: npm_last_math() tells which it was.
    const bool math_honoured = dma && (a.ksum || (!a.wide && !a.e.cs));
    npm::note_math(math_honoured ? a.math : 0);
    if (dma && a.ksum) {
        if constexpr (!A_KMAJ && !B_KMAJ) {
            if (a.math == 1) {
                if (a.ksum_op == 1) hipLaunchKernelGGL((sgemm_glds_kernel<false, false, false, 2, 2, 1, 1>), dim3(grid), dim3(NTHREADS), 0, stream, a);
                else hipLaunchKernelGGL((sgemm_glds_kernel<false, false, false, 2, 2, 2, 1>), dim3(grid), dim3(NTHREADS), 0, stream, a);
            } else if (a.math == 2) {
                if (a.ksum_op == 1) hipLaunchKernelGGL((sgemm_glds_kernel<false, false, false, 2, 2, 1, 2>), dim3(grid), dim3(NTHREADS), 0, stream, a);
                else hipLaunchKernelGGL((sgemm_glds_kernel<false, false, false, 2, 2, 2, 2>), dim3(grid), dim3(NTHREADS), 0, stream, a);
            } else {
                if (a.ksum_op == 1) hipLaunchKernelGGL((sgemm_glds_kernel<false, false, false, 2, 2, 1>), dim3(grid), dim3(NTHREADS), 0, stream, a);
                else hipLaunchKernelGGL((sgemm_glds_kernel<false, false, false, 2, 2, 2>), dim3(grid), dim3(NTHREADS), 0, stream, a);
            }
        }
    } else if (dma && a.math == 1 && !a.wide && !a.e.cs)
        hipLaunchKernelGGL((sgemm_glds_kernel<A_KMAJ, B_KMAJ, false, 2, 2, 0, 1>), dim3(grid), dim3(NTHREADS), 0, stream, a);
    else if (dma && a.math == 2 && !a.wide && !a.e.cs)
        hipLaunchKernelGGL((sgemm_glds_kernel<A_KMAJ, B_KMAJ, false, 2, 2, 0, 2>), dim3(grid), dim3(NTHREADS), 0, stream, a);
    else if (dma && a.wide)
        hipLaunchKernelGGL((sgemm_glds_kernel<A_KMAJ, B_KMAJ, false, 2, 4>), dim3(grid), dim3(512), 0, stream, a);
    else if (dma && a.e.cs)
        hipLaunchKernelGGL((sgemm_glds_kernel<A_KMAJ, B_KMAJ, true>), dim3(grid), dim3(NTHREADS), 0, stream, a);
    else if (dma && a.e.rowdot) {
        if constexpr (A_KMAJ && !B_KMAJ)       // dctx = dy wo (attentions.py:136): the one product that takes it
            hipLaunchKernelGGL((sgemm_glds_kernel<true, false, false, 2, 2, 0, 0, true>), dim3(grid), dim3(NTHREADS), 0, stream, a);
    } else if (dma)
        hipLaunchKernelGGL((sgemm_glds_kernel<A_KMAJ, B_KMAJ, false>), dim3(grid), dim3(NTHREADS), 0, stream, a);
    else if (!vec)
        hipLaunchKernelGGL((sgemm_mfma_kernel<A_KMAJ, B_KMAJ, false, 0>), dim3(grid), dim3(NTHREADS), 0, stream, a);
    else if (g_pipe == 1)
        hipLaunchKernelGGL((sgemm_mfma_kernel<A_KMAJ, B_KMAJ, true, 1>), dim3(grid), dim3(NTHREADS), 0, stream, a);
    else
        hipLaunchKernelGGL((sgemm_mfma_kernel<A_KMAJ, B_KMAJ, true, 0>), dim3(grid), dim3(NTHREADS), 0, stream, a);
}

}  // namespace

namespace npm_tile {

__global__ void __launch_bounds__(256) splitk_reduce_kernel(const ReduceArgs p) {
    const long mn = (long)p.M * p.N;
    const long total = p.slab;
    for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
         idx += (long)gridDim.x * blockDim.x) {
        const int z = (int)(idx / mn);
        const long el = idx - (long)z * mn;
        const int row = (int)(el / p.N), col = (int)(el - (long)row * p.N);
        float s = 0.f;
        for (int sp = 0; sp < p.splits; ++sp) s += p.ws[(long)sp * p.slab + idx];
        float v = p.e.alpha * s;
        if (p.e.flags & NPM_EPI_BIAS) v += p.e.bias[col];
        const int z0 = z / p.batch1, z1 = z - z0 * p.batch1;
        const long coff = z0 * p.sC0 + z1 * p.sC1;
        if (p.e.flags & NPM_EPI_RESIDUAL) v += p.e.R[coff + (long)row * p.e.ldr + col];
        p.e.C[coff + (long)row * p.e.ldc + col] = v;
    }
}

int launch_splitk_reduce(const ReduceArgs &r, hipStream_t stream) {
    const int grid = (int)std::min<long>((r.slab + 255) / 256, 8192);
    hipLaunchKernelGGL(splitk_reduce_kernel, dim3(grid), dim3(256), 0, stream, r);
    NPM_CHECK_LAUNCH();
    return NPM_OK;
}

}  // namespace npm_tile

extern "C" int npm_conv_set_dma(int on);
extern "C" int npm_conv_set_wgrad_blocks(int per_cu);
extern "C" int npm_conv_set_wave_prio(int bits);
extern "C" int npm_conv_set_korder(int order);
extern "C" int npm_conv_set_math(int mode);
extern "C" int npm_conv_set_wgrad_fused(int mode);
extern "C" int npm_attn_set_bwd16(int on);
extern "C" int npm_attn_set_fwd8(int mode);
extern "C" int npm_attn_set_stagger(int units);

extern "C" int npm_set_math(int mode) { return npm_set_tuning(NPM_TUNE_GEMM_MATH, mode); }
extern "C" int npm_get_math(void) { return g_math; }
extern "C" int npm_debug_gemm_trace(long long *buf) { g_trace = buf; return NPM_OK; }

extern "C" int npm_set_tuning(int knob, int value) {
    switch (knob) {
        case NPM_TUNE_GEMM_PIPELINE: g_pipe = value; return NPM_OK;
        case NPM_TUNE_GEMM_GROUP_M: g_group_m = value > 0 ? value : 8; return NPM_OK;
        case NPM_TUNE_GEMM_ABLATE: g_ablate = value; return NPM_OK;
        case NPM_TUNE_GEMM_WIDE_TILE: g_wide_tile = value; return NPM_OK;
        case NPM_TUNE_GEMM_BUF_EPILOGUE: g_buf_epilogue = value; return NPM_OK;
        case NPM_TUNE_CONV_DMA: return npm_conv_set_dma(value);
        case NPM_TUNE_CONV_WGRAD_BLOCKS: return npm_conv_set_wgrad_blocks(value);
        case NPM_TUNE_CONV_WGRAD_FUSED: return npm_conv_set_wgrad_fused(value);
        case NPM_TUNE_GEMM_MATH:
            if (value < 0 || value > 3) return npm::fail(NPM_E_BAD_ARGUMENT, "npm_set_tuning: NPM_TUNE_GEMM_MATH takes 0, 1, 2 or 3");
            g_math = value;
            return npm_conv_set_math(value == 3 ? 2 : value);        // the conv kernels have no f16 form: the bf16 split
        case NPM_TUNE_GEMM_WAVE_PRIO: g_wave_prio = value; return npm_conv_set_wave_prio(value);
        case NPM_TUNE_LN_BWD_BLOCKS: npm::set_ln_bwd_blocks(value); return NPM_OK;
        case NPM_TUNE_LN_NT_SPLIT: npm::set_ln_nt_split(value); return NPM_OK;
        case NPM_TUNE_EW_GRID_CAP: npm::set_ew_grid_cap(value); return NPM_OK;
        case NPM_TUNE_ATTN_STAGGER: return npm_attn_set_stagger(value);
        case NPM_TUNE_ATTN_BWD16: return npm_attn_set_bwd16(value);
        case NPM_TUNE_ATTN_FWD8: return npm_attn_set_fwd8(value);
        case NPM_TUNE_KSYNC: npm::set_ksync_every(value); return NPM_OK;
        case NPM_TUNE_CONV_KORDER: return npm_conv_set_korder(value);
        case NPM_TUNE_STREAM_NT: npm::set_stream_nt(value); return NPM_OK;
        case NPM_TUNE_GEMM_SPLIT_GENS: g_split_gens = value != 0; return NPM_OK;
        default: return npm::fail(NPM_E_BAD_ARGUMENT, "npm_set_tuning: unknown knob %d", knob);
    }
}

extern "C" int npm_sgemm(const npm_gemm *g) {
    NPM_REQUIRE_INIT();
    NPM_ARG(g != nullptr);
    NPM_ARG(g->m >= 0 && g->n >= 0 && g->k >= 0 && g->batch0 >= 1 && g->batch1 >= 1);
    if (g->m == 0 || g->n == 0) return NPM_OK;
    NPM_ARG(g->a != nullptr && g->b != nullptr && g->c != nullptr);
    const int epi = g->epilogue;
    NPM_ARG(!(epi & NPM_EPI_BIAS) || g->bias != nullptr);
    NPM_ARG(!(epi & NPM_EPI_RESIDUAL) || g->residual != nullptr);
    NPM_ARG(!(epi & (NPM_EPI_RELU_SAVE | NPM_EPI_RELU_MASK)) || g->aux != nullptr);
    NPM_ARG(!((epi & NPM_EPI_RELU_SAVE) && (epi & NPM_EPI_RELU_MASK)));
    NPM_ARG(!(epi & NPM_EPI_SOFTMAX_BWD) || (g->aux != nullptr && g->rowvec != nullptr &&
            !(epi & (NPM_EPI_RESIDUAL | NPM_EPI_RELU_SAVE | NPM_EPI_RELU_MASK | NPM_EPI_BIAS))));
    const bool rowdot = (epi & NPM_EPI_ROWDOT) != 0;
    NPM_ARG(!rowdot || (epi == NPM_EPI_ROWDOT && g->aux != nullptr && g->rowdot != nullptr && g->alpha == 1.f));

    const bool a_kmaj = g->trans_a == 0;   // A[M,K]: K contiguous
    const bool b_kmaj = g->trans_b != 0;   // B stored [N,K]: K contiguous
    if (!a_kmaj && b_kmaj) return npm::fail(NPM_E_UNSUPPORTED, "npm_sgemm: trans_a && trans_b is not used by the hot path");

    GemmArgs a{};
    a.A = g->a; a.B = g->b; a.e.C = g->c;
    a.lda = g->lda; a.ldb = g->ldb; a.e.ldc = g->ldc;
    a.sA0 = g->stride_a0; a.sA1 = g->stride_a1;
    a.sB0 = g->stride_b0; a.sB1 = g->stride_b1;
    a.sC0 = g->stride_c0; a.sC1 = g->stride_c1;
    a.M = g->m; a.N = g->n; a.K = g->k;
    a.batch1 = g->batch1;
    // float4 / DMA staging needs 16-B aligned rows in both operands.
    auto strides_ok = [](long s0, long s1) { return (s0 % 4 == 0) && (s1 % 4 == 0); };
    bool vec = aligned16(g->a) && aligned16(g->b) && g->lda % 4 == 0 && g->ldb % 4 == 0 &&
               strides_ok(g->stride_a0, g->stride_a1) && strides_ok(g->stride_b0, g->stride_b1);
    vec = vec && (a_kmaj ? g->k % 4 == 0 : g->m % 4 == 0);
    vec = vec && (b_kmaj ? g->k % 4 == 0 : g->n % 4 == 0);
    const bool want_colsum = g->colsum != nullptr;
    const bool want_bsum = g->bsum != nullptr, want_asum = g->asum != nullptr;
    NPM_ARG(!want_bsum || (!b_kmaj && g->batch0 == 1 && g->batch1 == 1 && !want_colsum));   // B stored [K, N]
    NPM_ARG(!want_asum || (!a_kmaj && g->batch0 == 1 && g->batch1 == 1 && !want_colsum && !want_bsum));   // A stored [K, M]
    const bool want_ksum = want_bsum || want_asum;
    const int ksum_len = want_bsum ? g->n : g->m;
    float *ksum_out = want_bsum ? g->bsum : g->asum;
    // 128 x 256 tile: only where it divides N and the LDS-DMA kernel will run (spans re-checked below)
    const int tile_n = ((g_wide_tile == 1 || (g_wide_tile == 2 && a_kmaj) || (g_wide_tile == 3 && a_kmaj && b_kmaj)) && g_pipe == 2 && vec && g->k % GK == 0 && g->n % 256 == 0 && !want_colsum && !want_ksum &&
                        (long)256 * g->ldb * 4 < (1L << 30)) ? 256 : BN;
    a.wide = tile_n == 256;
    a.tiles_m = (g->m + BM - 1) / BM;
    a.tiles_n = (g->n + tile_n - 1) / tile_n;
    a.e.alpha = g->alpha;
    a.e.flags = epi;
    a.e.bias = g->bias;
    a.e.R = (epi & NPM_EPI_RESIDUAL) ? g->residual : nullptr;
    a.e.ldr = g->ldr;
    a.e.aux = (epi & (NPM_EPI_RELU_SAVE | NPM_EPI_RELU_MASK | NPM_EPI_SOFTMAX_BWD | NPM_EPI_ROWDOT)) ? g->aux : nullptr;
    a.e.rowdot = rowdot ? g->rowdot : nullptr;
    a.e.rowdot_scale = g->rowdot_scale;
    a.e.rowdot_m = g->m;
    a.e.rowvec = (epi & NPM_EPI_SOFTMAX_BWD) ? g->rowvec : nullptr;
    a.e.ldaux = g->ldaux;
    {
        // the buffer epilogue addresses one block (<= 256 rows) at a time: its row pitches must keep that below 2^31
        auto fits = [&](long ld) { return (256L * ld + g->n) * 4 < (1L << 31); };
        a.e.buf_ok = g_buf_epilogue && fits(g->ldc) && (!(epi & NPM_EPI_RESIDUAL) || fits(g->ldr)) &&
                     (!(epi & (NPM_EPI_RELU_SAVE | NPM_EPI_RELU_MASK | NPM_EPI_SOFTMAX_BWD | NPM_EPI_ROWDOT)) || fits(g->ldaux)) && fits(g->n);
    }
    a.group_m = g_group_m;
    a.ablate = g_ablate;
    a.e.prio = g_wave_prio;
    a.math = g_math == 3 ? 2 : g_math;          // NPM_MATH_F16X2 where its kernel does not apply: the bf16 split
    a.trace = g_trace;

    const long batch = (long)g->batch0 * g->batch1;
    const long tiles = (long)a.tiles_m * a.tiles_n * batch;

    // Split-K: only for the linear epilogues, when the grid cannot fill 256 CUs x 2-3 blocks.
    int splits = 1;
    const int nkt = (g->k + BK - 1) / BK;
    const bool linear_epi = !(epi & (NPM_EPI_RELU_SAVE | NPM_EPI_RELU_MASK | NPM_EPI_RELU | NPM_EPI_SOFTMAX_BWD | NPM_EPI_ROWDOT));
    if (g->split_k > 1) {
        splits = g->split_k;
    } else if (g->split_k == 0 && linear_epi && tiles < 2L * npm::ctx().num_cus && nkt >= 16) {
        // (several generations of shorter K ranges: measured to pay for the A-heavy products that also sum A's columns -- the packed
        //  q/k/v weight gradient [3F, F] -- and to cost 0.3 ms per step on the FFN weight gradients, which sum B's: profiles/r04_splitk_sweep.log)
        const bool more_generations = g_math == 0 && g_split_gens && want_asum && a.tiles_m > a.tiles_n;
        splits = pick_splits(tiles, nkt, npm::ctx().num_cus, 0, g_math >= 2 ? 2 : g_math == 1 ? 3 : 4, more_generations ? 1 : 0);
    }
    if (!linear_epi) splits = 1;
    if (splits > nkt) splits = nkt > 0 ? nkt : 1;
    if (splits < 1) splits = 1;
    int kt_per_split = (nkt + splits - 1) / splits;
    if (kt_per_split < 1) kt_per_split = 1;
    splits = nkt > 0 ? (nkt + kt_per_split - 1) / kt_per_split : 1;
    a.splits = splits;
    a.k_per_split = splits > 1 ? kt_per_split * BK : (g->k > 0 ? nkt * BK : BK);

    if (want_colsum) splits = a.splits = 1, a.k_per_split = g->k > 0 ? nkt * BK : BK;
    npm::Scratch ws;
    if (splits > 1) {
        a.slab = batch * (long)g->m * g->n;
        int rc = ws.alloc(sizeof(float) * (size_t)a.slab * splits);
        if (rc) return rc;
        a.e.ws = (float *)ws.ptr;
    }

    const long grid = tiles * splits;
    NPM_ARG(grid < (1L << 31));
    hipStream_t stream = npm::ctx().stream;
    // Column sums of the stored C: taken in the epilogue of the LDS-DMA kernel when it is eligible,
    // otherwise by a separate pass over C (small / unaligned shapes).
    const long a_span = (a_kmaj ? (long)BM * g->lda + g->k : (long)a.k_per_split * g->lda + BM) * 4;
    const long b_span = (b_kmaj ? (long)tile_n * g->ldb + g->k : (long)a.k_per_split * g->ldb + tile_n) * 4;
    const bool dma = g_pipe == 2 && vec && g->k % GK == 0 && a.k_per_split % GK == 0 &&
                     a_span < (1L << 31) && b_span < (1L << 31);
    if (a.wide && !(dma && a.e.buf_ok))
        return npm::fail(NPM_E_UNSUPPORTED, "npm_sgemm: operand spans too large for the 128x256 tile (disable NPM_TUNE_GEMM_WIDE_TILE)");
    const bool dma_path = dma && a.e.buf_ok;
    if (rowdot && !(dma_path && a_kmaj && !b_kmaj && !a.wide && batch == 1 && splits == 1 && g->n % 128 == 0 && g_math == 0 && !a.ablate && !a.trace))
        return npm::fail(NPM_E_UNSUPPORTED, "npm_sgemm: NPM_EPI_ROWDOT needs the 128 x 128 LDS-DMA kernel in exact fp32 (A [M, K], B [K, N], "
                                            "aligned operands, k %% 16 == 0, n %% 128 == 0, no batch, no split-K)");
    // Column sums of B beside a TN product: in the kernel when the LDS-DMA path runs, else a pass over B.
    npm::Scratch bs_part;
    // NPM_MATH_F16X2: one product (no batch), the LDS-DMA path's alignment rules, no epilogue column sums
    const bool f16x2 = g_math == 3 && dma_path && batch == 1 && !a.wide && !want_colsum && g->k > 0 && !a.ablate && !a.trace;
    const bool ksum_fused = want_ksum && dma_path && !a_kmaj && !b_kmaj && g->k > 0 && !f16x2;
    const long ksum_rows = (long)splits * std::min(want_bsum ? a.tiles_m : a.tiles_n, GK);
    if (ksum_fused) {
        a.ksum_op = want_bsum ? 1 : 2;
        int rc = bs_part.alloc(sizeof(float) * (size_t)ksum_rows * ksum_len);
        if (rc) return rc;
        a.ksum = (float *)bs_part.ptr;
    }
    npm::Scratch cs_part;
    const long cs_rows = (long)g->batch0 * a.tiles_m * 2, cs_cols = (long)g->batch1 * g->n;
    if (want_colsum && dma_path) {
        int rc = cs_part.alloc(sizeof(float) * (size_t)cs_rows * cs_cols);
        if (rc) return rc;
        a.e.cs = (float *)cs_part.ptr;
    }
    npm::Scratch scales;
    if (f16x2) {
        // maxima along K of both operands -> power-of-two scales (one pass over each), then the product
        const size_t mn = (size_t)g->m + (size_t)g->n;
        int rc = scales.alloc(3 * sizeof(float) * mn);
        if (rc) return rc;
        float *sa = (float *)scales.ptr, *sb = sa + g->m, *ia = sb + g->n, *ib = ia + g->m;
        unsigned *ua = (unsigned *)(ib + g->n), *ub = ua + g->m;
        // the bias gradient beside a weight gradient (asum / bsum: column sums of an MN-major operand) rides the same pass
        rc = npm_tile::f16x2_scales(g->a, g->lda, a_kmaj, g->m, g->k, ua, sa, ia, stream, want_asum ? g->asum : nullptr);
        if (rc) return rc;
        rc = npm_tile::f16x2_scales(g->b, g->ldb, b_kmaj, g->n, g->k, ub, sb, ib, stream, want_bsum ? g->bsum : nullptr);
        if (rc) return rc;
        npm_tile::F16x2Args f{};
        f.A = g->a; f.B = g->b; f.lda = g->lda; f.ldb = g->ldb;
        f.M = g->m; f.N = g->n; f.K = g->k;
        f.a_kmaj = a_kmaj; f.b_kmaj = b_kmaj;
        f.tiles_m = a.tiles_m; f.tiles_n = a.tiles_n; f.group_m = a.group_m;
        f.splits = a.splits; f.k_per_split = a.k_per_split; f.slab = a.slab;
        f.sa = sa; f.sb = sb; f.inv_sa = ia; f.inv_sb = ib;
        f.e = a.e;
        npm::note_math(3);
        rc = npm_tile::launch_f16x2(f, stream);
        if (rc) return rc;
    } else {
        if (a_kmaj && b_kmaj) launch<true, true>(a, vec, dma, (int)grid, stream);
        else if (a_kmaj && !b_kmaj) launch<true, false>(a, vec, dma, (int)grid, stream);
        else launch<false, false>(a, vec, dma, (int)grid, stream);
        NPM_CHECK_LAUNCH();
    }

    if (want_ksum && !f16x2) {
        int rc = NPM_OK;
        if (!ksum_fused) rc = want_bsum ? npm::colsum_launch(g->b, g->bsum, g->k, g->n, g->ldb)
                                        : npm::colsum_launch(g->a, g->asum, g->k, g->m, g->lda);
        else rc = npm::colsum_launch(a.ksum, ksum_out, ksum_rows, ksum_len, ksum_len);
        if (rc) return rc;
    }
    if (want_colsum) {
        if (a.e.cs) return npm::colsum_launch(a.e.cs, g->colsum, cs_rows, cs_cols, cs_cols);
        // fallback: one strided pass per z1 over the stored C (batches must be row-contiguous)
        NPM_ARG(g->batch0 == 1 || g->stride_c0 == (int64_t)g->m * g->ldc);
        for (int z1 = 0; z1 < g->batch1; ++z1) {
            int rc = npm::colsum_launch(g->c + z1 * g->stride_c1, g->colsum + (long)z1 * g->n,
                                        (long)g->m * g->batch0, g->n, g->ldc);
            if (rc) return rc;
        }
        return NPM_OK;
    }
    if (splits > 1) {
        ReduceArgs r{};
        r.ws = a.e.ws; r.slab = a.slab; r.splits = splits;
        r.M = a.M; r.N = a.N; r.batch1 = a.batch1; r.sC0 = a.sC0; r.sC1 = a.sC1;
        r.e = a.e;
        return launch_splitk_reduce(r, stream);
    }
    return NPM_OK;
}
