// Fused scaled-dot-product attention core on the exact-fp32 MFMA (v_mfma_f32_32x32x2_f32), forward and backward.
//
// Replaces, for one call each, the chain of reference layers/attentions.py:103-112 (QK^T einsum, 1/sqrt(Dk),
// optional mask, Softmax.forward, PV einsum) and attentions.py:146-162 (dscores, dv, Softmax.backward, 1/sqrt(Dk),
// dq, dk).  The [B,H,Sq,Skv] probability tensor never goes to memory: the forward keeps a running row maximum and
// row sum (the blockwise online softmax the reference derives in layers/attentions_test.py:158-265) and saves one
// log-sum-exp per query row; the backward recomputes P = exp(scale * q.k - LSE) tile by tile.
//
// Layout: q/k/v/ctx and their gradients stay [B, S, H, D] (row pitch given: H*D, or 3*H*D inside a packed qkv
// buffer); a head is the column slice [h*D, (h+1)*D) of a row.  D in {16, 32, 64, 128}.
//
// MFMA orientation (one f32 per lane per operand: lane l supplies A[i = l & 31][k = l >> 5], B[k = l >> 5][j = l & 31];
// the result has its column j on the lane and rows (r & 3) + 8 (r >> 2) + 4 (l >> 5) in registers r = 0..15):
//   forward  S^T[kv, q] = K Q^T : the query is on the lane, so the softmax statistics of a row are lane-local (one
//            cross-half exchange per reduction) and P^T is, register for register, the B operand of
//            O^T[d, q] += V^T[d, kv] P^T[kv, q] -- no data movement between the two products.
//   backward S[q, kv] and dP[q, kv] with the KEY on the lane: P and dS are then the B operands of
//            dV^T[d, kv] += dO^T[d, q] P[q, kv] and dK^T[d, kv] += Q^T[d, q] dS[q, kv]; only dS crosses LDS, once,
//            for dQ[q, d] += dS[q, kv] K[kv, d].
// Operand tiles arrive by LDS-DMA (buffer_load_dwordx4 ... lds) with the bank swizzle applied on the SOURCE address.
#include <algorithm>
#include <cmath>

#include "npm_mfma_tile.h"

namespace {

using npm_tile::f32x16;
using npm_tile::xcd_remap;

constexpr float LOG2E = 1.44269504088896340736f;
constexpr float LN2 = 0.69314718055994530942f;
constexpr int OOB = 0x7FFFFFFF;

struct MhaArgs {
    const float *q, *k, *v;
    long q_pitch, k_pitch, v_pitch;
    float *ctx;
    long ctx_pitch;
    float *lse;                       // [B, H, Sq]
    float *delta;                     // backward scratch [B, H, Sq]: rowsum(dctx * ctx); for mha_bwd16_kernel / mha_bwd8_kernel: MINUS scale *
                                      // that, row (b, h, i) at delta[b delta_sb + h delta_sh + i], delta_len valid floats per (b, h)
    long delta_sb, delta_sh;          //   (own scratch: [B, H, sq_pad]; the caller's npm_mha_core.neg_delta: its strides)
    int delta_len;
    float *lse2;                      // mha_bwd8_kernel: [B, H, sq_pad] log2(e) * LSE, zeros in the padding
    int sq_pad;                       // seq_q rounded up to whole 32-query tiles
    const unsigned char *skip;        // optional tile summary (npm_mha_mask_summary): byte (qt, kb) at skip[b sb + h sh + qt nkb + kb]
    long skip_sb, skip_sh;
    int skip_nkb;
    long skip_all;                    // byte offset from an "any" byte to the "all" byte of the same tile (0: no "all" bits)
    const float *dctx;
    long dctx_pitch;
    float *dq, *dk, *dv;
    long dq_pitch, dk_pitch, dv_pitch;
    const unsigned char *mask;        // optional, element (b, h, i, j) at mask[b sb + h sh + i sq + j]; 0 = masked out
    long mask_sb, mask_sh, mask_sq;
    float *scores;                    // optional [B, H, Sq, Skv]: raw (unscaled, masked) scores kept for the backward
    int batch, heads, seq_q, seq_kv;
    float scale;
    int q_tiles;                      // forward: 128-query tiles per (b, h)
    int q_pair;                       // forward: 1 = a block takes the two tiles u and q_tiles - 1 - u (balanced work under a mask)
    int stagger;                      // forward: s_sleep(127) units one of the two blocks of a CU waits at its start
    long long *trace;                 // diagnostics (npm_debug_attn_trace): 16 s_memtime stamps per block, or null
};

// [rows][D] fp32 tile in LDS whose 16-byte chunk c of row r sits at chunk position c ^ sw(r).  sw is a LINEAR function of the
// row's low four bits, chosen per head size so that every access pattern of the kernels below is free of bank conflicts
// (tools/lds_bank_model.py models them against the bank rules of MI355X_MICROARCH.md, section LDS):
//   * rows read with one ds_read_b128 per lane (lane = row: the K-major MFMA operand), 16 x 16 x 4 and 32 x 32 x 2 layouts;
//   * COLUMN vectors (a lane reads VW adjacent columns of row 4 kk + j: the A operand of dV^T / dK^T / O^T) -- with the plain
//     sw(r) = r & 15 of rounds 2-4 rows j and 4 + j of one ds_read_b128 lane group landed on the same 16 chunk positions: every
//     such read was a 2-way conflict, 27.6 % of the LDS cycles of mha_bwd16_kernel and 33 % of mha_fwd8_kernel's
//     (profiles/r04_pmc_attn_sq.log; the model reproduces both figures).  Position bit 3 = row bit 3 XOR row bit 2 separates them;
//   * ds_read_b32 / ds_write_b32 along a row.
// Rows shorter than a 256-byte bank line (D = 32, 16) put 2 / 4 rows on one line: there the rows 4 kk + j of a column read (kk =
// 0, 1 in one ds_read_b32 / b64 lane group) start on the same banks whatever the chunk swizzle does inside a row, so the ROW itself
// moves too -- row r sits at row position r ^ (bit 2 of r) (neighbours swap places in every second group of four), and the chunk
// position comes from row bits 1 .. 3 (D = 32) / bit 3 (D = 16): every pattern conflict-free in the model; measured before (round 5
// counters, backward): 19 % (D = 32) and 34 % (D = 16) of the LDS cycles were conflicts.
template <int D>
struct Tile {
    static constexpr int CPR = D / 4;                           // chunks per row
    static constexpr int SWZ = (CPR < 16 ? CPR : 16) - 1;
    __host__ __device__ static constexpr int sw(int row) {
        return D >= 64 ? ((row & 15) ^ ((row & 4) << 1))
             : D == 32 ? (((row >> 1) & 1) | (((row >> 3) & 1) << 1) | (((row >> 2) & 1) << 2))
                       : (((row >> 3) & 1) << 1);
    }
    __host__ __device__ static constexpr int prow(int row) { return D <= 32 ? row ^ ((row >> 2) & 1) : row; }   // row position (an involution)
    static constexpr int PIECE_ROWS = 256 / D;                  // rows per 1 KiB DMA piece
    static constexpr int NG = D / 8;                            // k groups of 8 along a row
    static constexpr int NB = NG < 8 ? NG : 8;                  // lane-dependent bases of the row reads
    static_assert(D == 16 || D == 32 || D == 64 || D == 128 || D == 256, "head dim (256: the dS tile of a 256-key block)");
    __device__ static __forceinline__ int chunk(int row, int c) { return prow(row) * D + ((c ^ sw(row)) << 2); }
    __device__ static __forceinline__ int elem(int row, int col) { return prow(row) * D + ((((col >> 2) ^ sw(row)) << 2) | (col & 3)); }
    // byte offset (inside a [rows][pitch] global matrix) that lane `lane` of DMA piece `piece` copies from
    __device__ static __forceinline__ unsigned src(int lane, int piece, long pitch) {
        const int row = prow(piece * PIECE_ROWS + lane / CPR);                 // the row whose position this lane's 16 bytes are
        const int c = (lane % CPR) ^ sw(row);
        return (unsigned)((row * pitch + c * 4) * 4);
    }
    // Addresses as (lane-dependent base register) + (compile-time immediate), so that a fully unrolled phase costs a
    // handful of address registers instead of one per read (hipcc hoists every distinct address out of the loop).
    // Row read (ds_read_b128) of row l32 (+ a multiple of 16), group g: rb[g & 7] + row_imm(g).
    __device__ static __forceinline__ void row_bases(int l32, int half, int (&rb)[NB]) {
#pragma unroll
        for (int j = 0; j < NB; ++j) rb[j] = chunk(l32, 2 * j + half);
    }
    __device__ static constexpr int row_imm(int g) { return (g >> 3) * 64; }      // chunk bit 4 is above the swizzle
    // Column reads.  A lane reads VEC = D / 32 adjacent columns (one ds_read_b128 / b64 / b32) of row
    // 4 half + (r & 3) + 8 (r >> 2), starting at column VEC * lane: vb[(r >> 2) & 1][r & 3] + vec_imm(r).  Element t
    // of the vector belongs to MFMA tile t, whose row (or column) i = lane is head dimension VEC * lane + t.
    static constexpr int VEC = D >= 32 ? D / 32 : 1;
    __device__ static __forceinline__ void vec_bases(int half, int l32, int (&vb)[2][4]) {
        const int col = VEC * (D < 32 ? (l32 & (D - 1)) : l32);
#pragma unroll
        for (int par = 0; par < 2; ++par)
#pragma unroll
            for (int k = 0; k < 4; ++k) vb[par][k] = elem(4 * half + k + 8 * par, col);
    }
    __device__ static constexpr int vec_imm(int r) { return 16 * (r >> 3) * D; }
};

__device__ __forceinline__ float4 ld4(const float *p) { return *reinterpret_cast<const float4 *>(p); }
// VEC adjacent floats from LDS into v[0 .. VEC)
template <int VEC>
__device__ __forceinline__ void ldv(const float *p, float (&v)[4]) {
    if (VEC == 4) { const float4 x = ld4(p); v[0] = x.x; v[1] = x.y; v[2] = x.z; v[3] = x.w; }
    else if (VEC == 2) { const float2 x = *reinterpret_cast<const float2 *>(p); v[0] = x.x; v[1] = x.y; }
    else v[0] = p[0];
}
// The saved scores (2.1 GB at C4 / C5) are read ONCE by the backward: nontemporal loads, so that they do not displace
// the K / Q / dO tiles in L2 (backward 5.23 -> 5.19 ms).  The forward's stores keep the default policy: each lane
// writes 16 bytes of a different row and L2 merges the four pieces of a line; nontemporal stores cost the forward 3 %.
__device__ __forceinline__ float ld_stream(const float *p) { return __builtin_nontemporal_load(p); }
__device__ __forceinline__ float xhalf(float x) { return __shfl_xor(x, 32, 64); }
__device__ __forceinline__ float fast_exp2(float x) { return __builtin_amdgcn_exp2f(x); }

typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));

__device__ __forceinline__ float4 buf_load4(__amdgpu_buffer_rsrc_t r, int voff) {
    const u32x4_t v = __builtin_amdgcn_raw_buffer_load_b128(r, voff, 0, 0);
    return make_float4(__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w));
}
__device__ __forceinline__ void buf_store4(__amdgpu_buffer_rsrc_t r, int voff, float a, float b, float c, float d) {
    u32x4_t v = {__float_as_uint(a), __float_as_uint(b), __float_as_uint(c), __float_as_uint(d)};
    __builtin_amdgcn_raw_buffer_store_b128(v, r, voff, 0, 0);
}
__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void *base, long bytes) {
    return __builtin_amdgcn_make_buffer_rsrc((void *)base, 0, (int)std::min<long>(std::max<long>(bytes, 0), 0x7FFFFFF0L), 0x00020000);
}

using npm_tile::i32x4_t;
using npm_tile::make_desc;
using npm_tile::lds_offset;
using npm_tile::dma_group;

#define MFMA(a, b, c) __builtin_amdgcn_mfma_f32_32x32x2f32((a), (b), (c), 0, 0, 0)

// Instruction-order hints for one phase of `steps` steps, each `reads` LDS reads feeding `mfmas` MFMAs: the reads
// of step i + 1 are issued before the MFMAs of step i (one step of prefetch), nothing is hoisted further.  Without
// them hipcc clusters every LDS read of a fully unrolled phase in front of its first MFMA (hundreds of live
// registers, spills).  Masks: 0x100 DS read, 0x008 MFMA.
template <int STEPS, int READS, int MFMAS>
__device__ __forceinline__ void sched_pipeline() {
    __builtin_amdgcn_sched_group_barrier(0x100, READS, 0);
#pragma unroll
    for (int i = 0; i + 1 < STEPS; ++i) {
        __builtin_amdgcn_sched_group_barrier(0x100, READS, 0);
        __builtin_amdgcn_sched_group_barrier(0x008, MFMAS, 0);
    }
    __builtin_amdgcn_sched_group_barrier(0x008, MFMAS, 0);
}
#define PHASE_END() __builtin_amdgcn_sched_barrier(0)
#define FENCE() __builtin_amdgcn_sched_barrier(0)
// diagnostics: stamp `slot` of this block's trace record (wave 0, one chosen tile); costs nothing when trace == null
#define STAMP(slot) do { if (tr) { FENCE(); tr[slot] = __builtin_amdgcn_s_memtime(); FENCE(); } } while (0)

__device__ __forceinline__ void zero16(f32x16 &x) {
#pragma unroll
    for (int e = 0; e < 16; ++e) x[e] = 0.f;
}

// ---------------------------------------------------------------------------------------------------------------
// What bounds these kernels (tools/microbench/mfma_f32_chain.hip, profiles/r02_mfma_f32_chain.log): v_mfma_f32_32x32x2
// holds the SIMD's issue for its whole 64 cycles -- with one wave per SIMD every other instruction of that wave adds
// its full issue time (about 7 cycles) to the tile, with two waves about half of that.  So the design rule is the
// INSTRUCTION COUNT per MFMA: operands are read 16 bytes at a time wherever the layout allows (one LDS read per 4
// MFMAs), addresses are register + immediate, global accesses use scalar offsets, and nothing is computed twice.
// ---------------------------------------------------------------------------------------------------------------
// Forward.  Block = 4 wavefronts = 128 query rows of one (batch, head); wave w owns rows 32 w .. 32 w + 31 with
// its Q fragment (D / 2 registers) and O^T accumulators (D / 2 registers) resident.  K and V tiles of 32 keys
// stream through two LDS stages; one barrier per tile; two blocks per CU.
// ---------------------------------------------------------------------------------------------------------------
template <int D, bool MASK, bool SAVE, bool TRACE>
__global__ void __launch_bounds__(256, 2)
mha_fwd_kernel(const MhaArgs p) {
    using T = Tile<D>;
    constexpr int NG = D / 8, DT = (D + 31) / 32, VEC = T::VEC, TILE = 32 * D, PIECES = D / 8, PPW = (PIECES + 3) / 4;
    __shared__ __attribute__((aligned(16))) float smem[4 * TILE];            // K stages 0/1, V stages 0/1

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    long long *const blk_tr = (TRACE && p.trace && tid == 0) ? p.trace + (long)blockIdx.x * 16 : nullptr;   // block-level stamps 9..13
    if (blk_tr) blk_tr[9] = __builtin_amdgcn_s_memtime();
    const int l32 = lane & 31, half = lane >> 5;

    const int logical = xcd_remap(blockIdx.x, gridDim.x);
    // With a tile summary a block takes TWO query tiles, u and q_tiles - 1 - u, one after the other (p.q_pair): under a causal
    // mask -- under any mask whose rows see more and more keys -- the work of a query tile grows with its index, blocks of 4 / 8 /
    // 12 / 16 key tiles at C4's shape, and the dispatcher hands blocks to CUs in order: the light ones queue behind the heavy
    // ones (2.07 ms, against 1.45 ms for a mask with the SAME number of visited tiles spread evenly).  Paired, every block has
    // the same work.
    const bool pairing = MASK && p.q_pair;                                   // (summaries come with masks: no loop in the unmasked instances)
    const int units = pairing ? (p.q_tiles + 1) / 2 : p.q_tiles;
    const int unit = logical % units, bh = logical / units;
    const int b = bh / p.heads, h = bh - b * p.heads;
  for (int rep = 0; rep < (pairing ? 2 : 1); ++rep) {
    const int qt = rep == 0 ? unit : p.q_tiles - 1 - unit;
    if (rep == 1) {
        if (qt == unit) break;                                               // odd count: the middle tile has no partner
        __syncthreads();                                                     // every wave is done with the LDS stages of the first tile
    }
    const int qrow = qt * 128 + wave * 32 + l32;                             // this lane's query
    const bool qok = qrow < p.seq_q;

    const auto descK = make_desc(p.k + (long)b * p.seq_kv * p.k_pitch + h * D, ((long)(p.seq_kv - 1) * p.k_pitch + D) * 4);
    const auto descV = make_desc(p.v + (long)b * p.seq_kv * p.v_pitch + h * D, ((long)(p.seq_kv - 1) * p.v_pitch + D) * 4);
    const auto rsrcQ = make_rsrc(p.q + (long)b * p.seq_q * p.q_pitch + h * D, ((long)(p.seq_q - 1) * p.q_pitch + D) * 4);

    // Q fragment: element s of group g is Q[qrow][8 g + 4 half + s]
    float4 qf[NG];
#pragma unroll
    for (int g = 0; g < NG; ++g) qf[g] = buf_load4(rsrcQ, qok ? (int)((qrow * p.q_pitch + 8 * g + 4 * half) * 4) : OOB);

    unsigned vk[PPW], vv[PPW];
#pragma unroll
    for (int i = 0; i < PPW; ++i) {
        vk[i] = T::src(lane, wave * PPW + i, p.k_pitch);
        vv[i] = T::src(lane, wave * PPW + i, p.v_pitch);
    }
    const unsigned kstep = (unsigned)(32 * p.k_pitch * 4), vstep = (unsigned)(32 * p.v_pitch * 4);
    const unsigned lds_k = lds_offset(smem + wave * PPW * 256), lds_v = lds_k + 2 * TILE * 4;      // this wave's pieces, stage 0
    auto issue = [&](int t, int stage) {
        if (PIECES % 4 == 0 || wave * PPW < PIECES) {     // D = 16: two pieces per tile, waves 0 and 1 (PPW = 1)
            const unsigned tu = (unsigned)__builtin_amdgcn_readfirstlane(t);  // (called from the tile lambda the compiler no longer sees that t is uniform)
            dma_group<PPW>(descK, lds_k + stage * TILE * 4, tu * kstep, vk);  // the tile's first row = the scalar offset: range-checked
            dma_group<PPW>(descV, lds_v + stage * TILE * 4, tu * vstep, vv);
        }
    };

    int rb[T::NB], vb[2][4];
    T::row_bases(l32, half, rb);
    T::vec_bases(half, l32, vb);

    f32x16 O[DT];
#pragma unroll
    for (int t = 0; t < DT; ++t) zero16(O[t]);
    float m = -INFINITY, l = 0.f;
    const float c = p.scale * LOG2E;
    const int nt = (p.seq_kv + 31) / 32;
    // mask bytes of this (b, h) behind a descriptor: lane part (its query row, its 4 half keys) in one register, the tile's
    // first key in the scalar offset, the key inside the tile in the immediate; a key or query beyond the end reads 0 = masked
    const auto rsrcM = make_rsrc(MASK ? p.mask + b * p.mask_sb + h * p.mask_sh : nullptr, MASK ? (long)(p.seq_q - 1) * p.mask_sq + p.seq_kv : 0);
    const int mvoff = qok ? (int)(qrow * p.mask_sq + 4 * half) : OOB;
    const bool mask_dw = MASK && (p.mask_sq & 3) == 0 && (p.seq_kv & 3) == 0 && ((unsigned long)(p.mask + b * p.mask_sb + h * p.mask_sh) & 3) == 0;
    // Saved scores of this (b, h): [seq_q][seq_kv] behind one descriptor.  A lane's part of the address (its query row,
    // its 4 half keys) is ONE register, the tile's first key a scalar offset, the register group an immediate: a full
    // tile is four bare buffer_store_dwordx4, rows beyond seq_q carry an out-of-range offset.
    const auto rsrcS = make_rsrc(SAVE ? p.scores + (long)bh * p.seq_q * p.seq_kv : nullptr, SAVE ? (long)p.seq_q * p.seq_kv * 4 : 0);
    const int svoff = qok ? (int)(((long)qrow * p.seq_kv + 4 * half) * 4) : OOB;
    const bool rows16 = (p.seq_kv & 3) == 0;             // every score row starts 16-byte aligned

    // Tile skipping (masks): bit t of `act` = some query of this block may look at key tile t, bit t of `mine` = some query
    // of this wave may (p.skip: one byte per (query tile of 32, key block of 128), bit w = keys 16 w .. 16 w + 15 of the
    // block; npm_mha_mask_summary).  Tiles outside `act` are not visited (no pieces, no barrier, nothing stored for them);
    // in a visited tile a wave outside `mine` only takes part in the pieces and the barrier.
    // `plain`: bit t = EVERY position of this wave's 32 queries x the 32 keys of tile t is allowed (the summary's "all" bits):
    // such a tile needs no mask bytes and no compares -- under a causal mask all but one visited tile per wave.
    unsigned long act = ~0ul, mine = ~0ul, plain = 0ul;
    const bool skipping = MASK && p.skip != nullptr;
    if (skipping) {
        const unsigned char *sk = p.skip + b * p.skip_sb + h * p.skip_sh;
        unsigned any = 0, own = 0, full = 0;
        if (lane < nt) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int qtile = 4 * qt + i;
                const unsigned bits = qtile * 32 < p.seq_q ? (sk[(long)qtile * p.skip_nkb + (lane >> 2)] >> (2 * (lane & 3))) & 3u : 0u;
                any |= bits;
                if (i == wave) own = bits;
            }
            const int mytile = 4 * qt + wave;
            if (p.skip_all && mytile * 32 < p.seq_q)
                full = (sk[p.skip_all + (long)mytile * p.skip_nkb + (lane >> 2)] >> (2 * (lane & 3))) & 3u;
        }
        act = __builtin_amdgcn_ballot_w64(any != 0);
        mine = __builtin_amdgcn_ballot_w64(own != 0);
        plain = __builtin_amdgcn_ballot_w64(full == 3u);
    }
    const int t_first = skipping ? (act ? __builtin_ctzl(act) : -1) : 0;
    if (t_first >= 0) issue(t_first, 0);
    // The two blocks of a CU run the same program; started together they reach their softmax (no MFMA) together and
    // the matrix pipe idles.  Blocks b and b + 256 share a CU on a first dispatch: delay one of them by about half a
    // tile, later generations inherit the offset.  (Placement is not guaranteed: this is for speed only.)
    if (p.stagger && rep == 0 && ((blockIdx.x >> 8) & 1)) {
        for (int i = 0; i < p.stagger; ++i) __builtin_amdgcn_s_sleep(127);
    }
    if (blk_tr) blk_tr[10] = __builtin_amdgcn_s_memtime();
    // One key tile out of LDS stage STG.  The loop below is unrolled by two so that the stage is a compile-time constant: every
    // LDS address of the tile is a loop-invariant register plus an immediate (32 vector-ALU adds per tile fewer -- each costs the
    // matrix pipe its issue cycles, tools/microbench/mfma_f32_16x16.hip).
    auto tile = [&](auto stage_c, const int t, const int t_next) __attribute__((always_inline)) {
        constexpr int STG = decltype(stage_c)::value;
        npm_tile::dma_barrier();                        // tile t has landed (every wave's pieces); nobody still reads the stage refilled next
        long long *tr = (TRACE && p.trace && tid == 0 && t == (nt > 5 ? 5 : 0)) ? p.trace + (long)blockIdx.x * 16 : nullptr;
        if (TRACE && p.trace && tid == 0 && t == (nt > 5 ? 6 : 1)) { FENCE(); p.trace[(long)blockIdx.x * 16 + 4] = __builtin_amdgcn_s_memtime(); FENCE(); }
        STAMP(0);
        if (t_next >= 0) issue(t_next, STG ^ 1);
        if (skipping && !((mine >> t) & 1)) return;     // no query of this wave may look at these 32 keys
        const float *sK = smem + STG * TILE, *sV = smem + (2 + STG) * TILE;
        // Mask bytes of this tile: requested now, used behind the 64 MFMAs of the score product.  The four keys a register group
        // holds are four ADJACENT bytes: one dword load per group (4 loads, 4 registers) when every row of the mask starts on a
        // 4-byte boundary and no dword straddles the end; byte by byte (16 loads, issued late: the register file has no room for
        // 16 more live values across the score product at D = 128) otherwise.
        unsigned mkw[4];
        const bool masked_tile = MASK && !((plain >> t) & 1);     // (plain is 0 without a summary: every tile reads its mask)
        if (masked_tile && mask_dw) {
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4) mkw[g4] = __builtin_amdgcn_raw_buffer_load_b32(rsrcM, mvoff + 8 * g4, 32 * t, 0);
        }

        // ---- S^T[kv, q] = K Q^T: NG steps of (1 row read, 4 MFMAs), every read one step ahead of its use
        f32x16 S;
        float4 fk[2];
        float ev[2][4];
        fk[0] = ld4(sK + rb[0]);
#pragma unroll
        for (int g = 0; g < NG; ++g) {
            if (g + 1 < NG) fk[(g + 1) & 1] = ld4(sK + rb[(g + 1) & 7] + T::row_imm(g + 1));
            else ldv<VEC>(sV + vb[0][0], ev[0]);                                 // first vector of the PV phase
            FENCE();
            if (g == 0) {                                                        // the first MFMA starts from the constant 0: nothing zeroes S
                const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
                S = MFMA(fk[0].x, qf[0].x, zero);
            } else {
                S = MFMA(fk[g & 1].x, qf[g].x, S);
            }
            S = MFMA(fk[g & 1].y, qf[g].y, S);
            S = MFMA(fk[g & 1].z, qf[g].z, S);
            S = MFMA(fk[g & 1].w, qf[g].w, S);
            FENCE();
        }
        STAMP(1);
        const int kv0 = 32 * t + 4 * half;              // register r holds key kv0 + (r & 3) + 8 (r >> 2)
        if (32 * t + 32 > p.seq_kv) {
#pragma unroll
            for (int r = 0; r < 16; ++r)
                if (kv0 + (r & 3) + 8 * (r >> 2) >= p.seq_kv) S[r] = -INFINITY;
        }
        if (masked_tile) {
            if (!mask_dw) {
#pragma unroll
                for (int g4 = 0; g4 < 4; ++g4) {
                    mkw[g4] = 0;
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        mkw[g4] |= (unsigned)__builtin_amdgcn_raw_buffer_load_b8(rsrcM, mvoff + e + 8 * g4, 32 * t, 0) << (8 * e);
                }
            }
#pragma unroll
            for (int r = 0; r < 16; ++r)
                if (((mkw[r >> 2] >> (8 * (r & 3))) & 0xffu) == 0) S[r] = -INFINITY;
        }
        if (SAVE) {
            const int stile = 32 * t * 4;                                     // scalar: byte offset of the tile's first key
            if (rows16 && 32 * t + 32 <= p.seq_kv) {
#pragma unroll
                for (int g4 = 0; g4 < 4; ++g4) {
                    const u32x4_t v = {__float_as_uint(S[4 * g4]), __float_as_uint(S[4 * g4 + 1]), __float_as_uint(S[4 * g4 + 2]), __float_as_uint(S[4 * g4 + 3])};
                    __builtin_amdgcn_raw_buffer_store_b128(v, rsrcS, svoff + 32 * g4, stile, 0);
                }
            } else {                                                          // ragged last tile / unaligned rows: element by element
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int e = (r & 3) + 8 * (r >> 2);
                    __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(S[r]), rsrcS, (qok && kv0 + e < p.seq_kv) ? svoff + 4 * e : OOB, stile, 0);
                }
            }
        }
        // ---- online softmax in the exp2 domain: the statistics of query `qrow` live on its two lanes.  The running
        //      maximum is only a reference point: it moves (and the accumulators are rescaled) when some row of the wave
        //      would otherwise exceed it by more than 2^RESCALE -- a few times per block instead of almost every tile
        //      (the rescale is 32 packed multiplies that issue beside nothing).  Probabilities stay below 2^RESCALE.
        constexpr float RESCALE = 10.f;
        float tmax = S[0];
#pragma unroll
        for (int r = 1; r < 16; ++r) tmax = fmaxf(tmax, S[r]);
        tmax = fmaxf(tmax, xhalf(tmax)) * c;
        float m_new = m;
        if (__builtin_amdgcn_ballot_w64(tmax > m + RESCALE) != 0) {       // first tile: m = -inf
            m_new = fmaxf(m, tmax);
            // A row whose keys were ALL masked so far (m = -inf: l = 0, O = 0) keeps alpha finite: exp2(-inf - (-inf)) is
            // NaN and would poison a row that still has keys to come (banded, left-padded, block-diagonal masks)
            const float alpha = (MASK && m == -INFINITY) ? 0.f : fast_exp2(m - m_new);
            l *= alpha;
#pragma unroll
            for (int t2 = 0; t2 < DT; ++t2)
#pragma unroll
                for (int e = 0; e < 16; ++e) O[t2][e] *= alpha;
        }
        // ... and takes its exponents against 0 while its reference point is still -inf: masked entries give
        // exp2(-inf) = 0, not fma(-inf, c, +inf) = NaN.  Only a row with NO key left ends with l = 0 -> NaN, as
        // np.where(mask, s, -inf) followed by the softmax does (attentions.py:105-109).
        const float m_use = (MASK && m_new == -INFINITY) ? 0.f : m_new;
        float psum = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            S[r] = fast_exp2(fmaf(S[r], c, -m_use));
            psum += S[r];
        }
        l += psum + xhalf(psum);
        m = m_new;
        FENCE();
        STAMP(2);
        // ---- O^T[d, q] += V^T[d, kv] P^T[kv, q]: 16 steps (one key row each) of (1 vector read, DT MFMAs);
        //      tile t2 row `lane` is head dimension VEC lane + t2
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            if (r + 1 < 16) ldv<VEC>(sV + vb[((r + 1) >> 2) & 1][(r + 1) & 3] + T::vec_imm(r + 1), ev[(r + 1) & 1]);
            FENCE();
#pragma unroll
            for (int t2 = 0; t2 < DT; ++t2) O[t2] = MFMA(ev[r & 1][t2], S[r], O[t2]);
            FENCE();
        }
        STAMP(3);
    };
    auto next_tile = [&](int t) -> int {               // the tile visited after tile t, or -1
        if (!skipping) return t + 1 < nt ? t + 1 : -1;
        const unsigned long m = t < 63 ? act >> (t + 1) : 0ul;
        return m ? t + 1 + __builtin_ctzl(m) : -1;
    };
    for (int t = t_first; t >= 0;) {
        const int t1 = next_tile(t);
        tile(std::integral_constant<int, 0>{}, t, t1);
        if (t1 < 0) break;
        const int t2 = next_tile(t1);
        tile(std::integral_constant<int, 1>{}, t1, t2);
        t = t2;
    }
    if (blk_tr) blk_tr[12] = __builtin_amdgcn_s_memtime();

    const float inv = 1.f / l;
    if (half == 0 && qok) p.lse[(long)bh * p.seq_q + qrow] = (m + __builtin_amdgcn_logf(l)) * LN2;       // v_log_f32 is log2
    const auto rsrcC = make_rsrc(p.ctx + (long)b * p.seq_q * p.ctx_pitch + h * D, ((long)(p.seq_q - 1) * p.ctx_pitch + D) * 4);
    // register r of the DT tiles together is VEC adjacent head dimensions, from VEC * (4 half + (r & 3) + 8 (r >> 2))
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int d0 = VEC * (4 * half + (r & 3) + 8 * (r >> 2));
        if (VEC * (8 * (r >> 2) + (r & 3)) < D) {                                // compile-time part of the bound (D = 16)
            const int off = (qok && d0 < D) ? (int)((qrow * p.ctx_pitch + d0) * 4) : OOB;
            if (VEC == 4) buf_store4(rsrcC, off, O[0][r] * inv, O[DT > 1 ? 1 : 0][r] * inv, O[DT > 2 ? 2 : 0][r] * inv, O[DT > 3 ? 3 : 0][r] * inv);
            else
#pragma unroll
                for (int t2 = 0; t2 < DT; ++t2)
                    __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(O[t2][r] * inv), rsrcC, off == OOB ? OOB : off + 4 * t2, 0, 0);
        }
    }
    if (blk_tr) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        blk_tr[13] = __builtin_amdgcn_s_memtime();
    }
  }   // rep
}

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
// maxima as the instructions are: fmaxf() canonicalises each input first (a v_max_f32 x, x per operand)
__device__ __forceinline__ float max3(float a, float b, float c) {
    float r;
    asm("v_max3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}
__device__ __forceinline__ float fmax_raw(float a, float b) {
    float r;
    asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
// Reductions over the four lanes l16 + 16 k (k = 0 .. 3) that share a query in the 16 x 16 x 4 result layout, every lane getting
// the result: gfx950's v_permlane16_swap / v_permlane32_swap exchange 16-lane rows / wave halves between two registers, so with
// both operands = x the two results hold "this row" and "the partner row" in every lane -- a copy, the swap and the operation
// per step, where __shfl_xor costs four address instructions, a ds_bpermute and an LDS round trip.
__device__ __forceinline__ float quad_max(float x) {
    const auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(x), __float_as_uint(x), false, false);
    const float y = fmax_raw(__uint_as_float(r[0]), __uint_as_float(r[1]));
    const auto q = __builtin_amdgcn_permlane32_swap(__float_as_uint(y), __float_as_uint(y), false, false);
    return fmax_raw(__uint_as_float(q[0]), __uint_as_float(q[1]));
}
__device__ __forceinline__ float quad_sum(float x) {
    const auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(x), __float_as_uint(x), false, false);
    const float y = __uint_as_float(r[0]) + __uint_as_float(r[1]);
    const auto q = __builtin_amdgcn_permlane32_swap(__float_as_uint(y), __float_as_uint(y), false, false);
    return __uint_as_float(q[0]) + __uint_as_float(q[1]);
}
#define MFMA16(a, b, c) __builtin_amdgcn_mfma_f32_16x16x4f32((a), (b), (c), 0, 0, 0)

// ---------------------------------------------------------------------------------------------------------------
// Forward on v_mfma_f32_16x16x4_f32, EIGHT wavefronts per block (round 4).  The 32 x 32 kernel above gives a wave 32
// queries; its O^T accumulators have the HEAD DIMENSION on the rows of 32 x 32 tiles, so head size 16 multiplies half of
// its P V tiles by padding, and at 200+ registers only two waves share a SIMD.  Here a wave owns 16 queries of the block's
// 128: the Q fragment (D / 4 registers), the O^T accumulators (D / 4) and two 16 x 16 score tiles fit 128 registers, FOUR
// waves share a SIMD (two blocks of eight per CU), and every head size fills its tiles.
//   S^T[kv, q] = K Q^T   A = row kv = 16 t + l16 of the K tile, 16 bytes at d = 16 c + 4 kk (one LDS read per 4 MFMAs),
//                        B = the Q fragment; the result has its query on the lane (column l16) and keys 16 t + 4 kk + r in
//                        registers: the softmax statistics of a query live on its four lanes (two cross-lane steps each),
//   O^T[d, q] += V^T[d, kv] P^T[kv, q]   B = P[t][r], register for register the result of the exp; A = VW = min(4, D / 16)
//                        adjacent head dimensions of V row kv = 16 t + 4 kk + r (element e belongs to d-tile e).
// Same K / V tiles of 32 keys through two LDS stages, same online softmax, masks, saved scores, tile skipping and pairing
// as mha_fwd_kernel.
// ---------------------------------------------------------------------------------------------------------------
template <int D, bool MASK, bool SAVE>
__global__ void __launch_bounds__(512, 4)
mha_fwd8_kernel(const MhaArgs p) {
    using T = Tile<D>;
    constexpr int NC = D / 16, TILE = 32 * D, ROWS16 = 16 * D;
    constexpr int PIECES = D / 8, PPW = PIECES >= 8 ? PIECES / 8 : 1;             // 1 KiB DMA pieces of a K (or V) tile; per wave
    constexpr int VW = D >= 64 ? 4 : D / 16, NV = NC / VW, NB = NC < 4 ? NC : 4;
    __shared__ __attribute__((aligned(16))) float smem[4 * TILE];                // K stages 0 / 1, V stages 0 / 1

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);                  // 0 .. 7
    const int l16 = lane & 15, kk = lane >> 4;
    const int logical = xcd_remap(blockIdx.x, gridDim.x);
    const bool pairing = MASK && p.q_pair;                                       // see mha_fwd_kernel
    const int units = pairing ? (p.q_tiles + 1) / 2 : p.q_tiles;
    const int unit = logical % units, bh = logical / units;
    const int b = bh / p.heads, h = bh - b * p.heads;
    const auto descK = make_desc(p.k + (long)b * p.seq_kv * p.k_pitch + h * D, ((long)(p.seq_kv - 1) * p.k_pitch + D) * 4);
    const auto descV = make_desc(p.v + (long)b * p.seq_kv * p.v_pitch + h * D, ((long)(p.seq_kv - 1) * p.v_pitch + D) * 4);
    const auto rsrcQ = make_rsrc(p.q + (long)b * p.seq_q * p.q_pitch + h * D, ((long)(p.seq_q - 1) * p.q_pitch + D) * 4);
    const auto rsrcC = make_rsrc(p.ctx + (long)b * p.seq_q * p.ctx_pitch + h * D, ((long)(p.seq_q - 1) * p.ctx_pitch + D) * 4);
    const auto rsrcM = make_rsrc(MASK ? p.mask + b * p.mask_sb + h * p.mask_sh : nullptr, MASK ? (long)(p.seq_q - 1) * p.mask_sq + p.seq_kv : 0);
    const auto rsrcS = make_rsrc(SAVE ? p.scores + (long)bh * p.seq_q * p.seq_kv : nullptr, SAVE ? (long)p.seq_q * p.seq_kv * 4 : 0);
    const bool mask_dw = MASK && (p.mask_sq & 3) == 0 && (p.seq_kv & 3) == 0 && ((unsigned long)(p.mask + b * p.mask_sb + h * p.mask_sh) & 3) == 0;
    const bool rows16 = (p.seq_kv & 3) == 0;                                     // every score row starts 16-byte aligned
    const float c = p.scale * LOG2E;
    const int nt = (p.seq_kv + 31) / 32;

    unsigned vk[PPW], vv[PPW];
#pragma unroll
    for (int i = 0; i < PPW; ++i) {
        vk[i] = T::src(lane, wave * PPW + i, p.k_pitch);
        vv[i] = T::src(lane, wave * PPW + i, p.v_pitch);
    }
    const bool owner = PIECES >= 8 || wave < PIECES;                             // D < 64: fewer pieces than waves
    const unsigned kstep = (unsigned)(32 * p.k_pitch * 4), vstep = (unsigned)(32 * p.v_pitch * 4);
    const unsigned lds_k = lds_offset(smem + wave * PPW * 256), lds_v = lds_k + 2 * TILE * 4;
    auto issue = [&](int t, int stage) {
        if (owner) {
            const unsigned tu = (unsigned)__builtin_amdgcn_readfirstlane(t);
            dma_group<PPW>(descK, lds_k + stage * TILE * 4, tu * kstep, vk);
            dma_group<PPW>(descV, lds_v + stage * TILE * 4, tu * vstep, vv);
        }
    };
    int rb[NB], vb[4];
#pragma unroll
    for (int j = 0; j < NB; ++j) rb[j] = T::chunk(l16, 4 * j + kk);             // K rows 16 t + l16, chunk 4 c + kk: + t ROWS16 + (c >> 2) 64
#pragma unroll
    for (int j = 0; j < 4; ++j) vb[j] = T::elem(4 * kk + j, VW * l16);          // V rows 16 t + 4 kk + r, columns VW l16 ..: + t ROWS16 (+ 64)

  for (int rep = 0; rep < (pairing ? 2 : 1); ++rep) {
    const int qt = rep == 0 ? unit : p.q_tiles - 1 - unit;
    if (rep == 1) {
        if (qt == unit) break;
        __syncthreads();
    }
    // The lane index is made afresh per query tile (v_mbcnt inside an opaque asm): hipcc hoists everything it can derive from
    // threadIdx out of the rep loop, and in the masked instances at D = 128 -- which sit at the 128-register limit of four waves
    // per SIMD -- what it kept alive across the key-tile loop for the second query tile (this lane's query row, the byte
    // addresses and shifts of the summary look-up below: two 64-bit pointers and three words) went to scratch.
    int ln;
    asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(ln));
    const int qrow = qt * 128 + wave * 16 + (ln & 15);                           // this lane's query (shared by its four lanes kk)
    const bool qok = qrow < p.seq_q;
    const int mvoff = qok ? (int)(qrow * p.mask_sq + 4 * kk) : OOB;              // mask bytes: this query's row, keys 4 kk ..
    const int svoff = qok ? (int)(((long)qrow * p.seq_kv + 4 * kk) * 4) : OOB;   // saved scores, the same way

    // tiles this block / this wave visits, tiles without an excluded position (see mha_fwd_kernel)
    unsigned long act = ~0ul, mine = ~0ul, plain = 0ul;
    const bool skipping = MASK && p.skip != nullptr;
    if (skipping) {
        const unsigned char *sk = p.skip + b * p.skip_sb + h * p.skip_sh;
        unsigned any = 0, own = 0, full = 0;
        const int mytile = 4 * qt + (wave >> 1);                                 // the summary row (32 queries) of this wave's 16
        if (ln < nt) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int qtile = 4 * qt + i;
                const unsigned bits = qtile * 32 < p.seq_q ? (sk[(long)qtile * p.skip_nkb + (ln >> 2)] >> (2 * (ln & 3))) & 3u : 0u;
                any |= bits;
                if (qtile == mytile) own = bits;
            }
            if (p.skip_all && mytile * 32 < p.seq_q)
                full = (sk[p.skip_all + (long)mytile * p.skip_nkb + (ln >> 2)] >> (2 * (ln & 3))) & 3u;
        }
        act = __builtin_amdgcn_ballot_w64(any != 0);
        mine = __builtin_amdgcn_ballot_w64(own != 0);
        plain = __builtin_amdgcn_ballot_w64(full == 3u);
    }
    const int t_first = skipping ? (act ? __builtin_ctzl(act) : -1) : 0;
    if (t_first >= 0) issue(t_first, 0);
    float4 qf[NC];                                                               // Q[qrow][16 c + 4 kk + s] (loaded behind the summary's
#pragma unroll                                                                   //  temporaries: the register file is full at D = 128)
    for (int cc = 0; cc < NC; ++cc) qf[cc] = buf_load4(rsrcQ, qok ? (int)((qrow * p.q_pitch + 16 * cc + 4 * kk) * 4) : OOB);

    f32x4 O[NC];
#pragma unroll
    for (int x = 0; x < NC; ++x) O[x] = f32x4{0.f, 0.f, 0.f, 0.f};
    float m = -INFINITY, l = 0.f;

    auto tile = [&](auto stage_c, const int t, const int t_next) __attribute__((always_inline)) {
        constexpr int STG = decltype(stage_c)::value;
        npm_tile::dma_barrier();                        // tile t has landed (every wave's pieces); nobody still reads the stage refilled next
        if (t_next >= 0) issue(t_next, STG ^ 1);
        if (skipping && !((mine >> t) & 1)) return;
        const float *sK = smem + STG * TILE, *sV = smem + (2 + STG) * TILE;
        const bool masked_tile = MASK && !((plain >> t) & 1);
        unsigned mkw[2];                                // the 4 keys of a register group are 4 adjacent mask bytes
        if (masked_tile && mask_dw) {
#pragma unroll
            for (int g2 = 0; g2 < 2; ++g2) mkw[g2] = __builtin_amdgcn_raw_buffer_load_b32(rsrcM, mvoff + 16 * g2, 32 * t, 0);
        }
        // ---- S^T[kv, q] = K Q^T: NC steps of (2 row reads, 8 MFMAs), reads one step ahead
        f32x4 S[2];
        constexpr int AHEAD = (MASK && D == 128) ? 0 : 1;   // K rows one step ahead -- not in the instances at the register limit
        float4 fk[2][2];
        if (AHEAD) {
#pragma unroll
            for (int g2 = 0; g2 < 2; ++g2) fk[0][g2] = ld4(sK + rb[0] + g2 * ROWS16);
        }
#pragma unroll
        for (int cc = 0; cc < NC; ++cc) {
            if (AHEAD && cc + 1 < NC) {
#pragma unroll
                for (int g2 = 0; g2 < 2; ++g2) fk[(cc + 1) & 1][g2] = ld4(sK + rb[(cc + 1) & 3] + ((cc + 1) >> 2) * 64 + g2 * ROWS16);
            }
            if (!AHEAD) {
#pragma unroll
                for (int g2 = 0; g2 < 2; ++g2) fk[cc & 1][g2] = ld4(sK + rb[cc & 3] + (cc >> 2) * 64 + g2 * ROWS16);
            }
            FENCE();
#pragma unroll
            for (int g2 = 0; g2 < 2; ++g2) {
                if (cc == 0) {                          // the first MFMA starts from the constant 0: nothing zeroes S
                    const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
                    S[g2] = MFMA16(fk[0][g2].x, qf[0].x, zero);
                } else {
                    S[g2] = MFMA16(fk[cc & 1][g2].x, qf[cc].x, S[g2]);
                }
                S[g2] = MFMA16(fk[cc & 1][g2].y, qf[cc].y, S[g2]);
                S[g2] = MFMA16(fk[cc & 1][g2].z, qf[cc].z, S[g2]);
                S[g2] = MFMA16(fk[cc & 1][g2].w, qf[cc].w, S[g2]);
            }
            FENCE();
        }
        // first vectors of the P V product: requested under the softmax -- except in the masked instances at D = 128, which
        // sit at the 128-register limit of four waves per SIMD (they spilled 6-8 registers): there behind it
        constexpr bool EARLY_V = !(MASK && D == 128);
        float ev[2][NV][4];
        if (EARLY_V) {
#pragma unroll
            for (int v = 0; v < NV; ++v) ldv<VW>(sV + vb[0] + 64 * v, ev[0][v]);
        }
        const int kv0 = 32 * t + 4 * kk;                // register r of group g2 holds key kv0 + 16 g2 + r
        if (32 * t + 32 > p.seq_kv) {
#pragma unroll
            for (int g2 = 0; g2 < 2; ++g2)
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    if (kv0 + 16 * g2 + r >= p.seq_kv) S[g2][r] = -INFINITY;
        }
        if (masked_tile) {
            if (!mask_dw) {
#pragma unroll
                for (int g2 = 0; g2 < 2; ++g2) {
                    mkw[g2] = 0;
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        mkw[g2] |= (unsigned)__builtin_amdgcn_raw_buffer_load_b8(rsrcM, mvoff + e + 16 * g2, 32 * t, 0) << (8 * e);
                }
            }
#pragma unroll
            for (int g2 = 0; g2 < 2; ++g2)
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    if (((mkw[g2] >> (8 * r)) & 0xffu) == 0) S[g2][r] = -INFINITY;
        }
        if (SAVE) {
            const int stile = 32 * t * 4;
            if (rows16 && 32 * t + 32 <= p.seq_kv) {
#pragma unroll
                for (int g2 = 0; g2 < 2; ++g2) {
                    const u32x4_t v = {__float_as_uint(S[g2][0]), __float_as_uint(S[g2][1]), __float_as_uint(S[g2][2]), __float_as_uint(S[g2][3])};
                    __builtin_amdgcn_raw_buffer_store_b128(v, rsrcS, svoff + 64 * g2, stile, 0);
                }
            } else {
#pragma unroll
                for (int g2 = 0; g2 < 2; ++g2)
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(S[g2][r]), rsrcS,
                                                              (qok && kv0 + 16 * g2 + r < p.seq_kv) ? svoff + 64 * g2 + 4 * r : OOB, stile, 0);
            }
        }
        // ---- online softmax in the exp2 domain (mha_fwd_kernel's rules: lazy reference point, rows that start masked)
        // (vector-ALU diet: four waves share a SIMD here and every vector instruction takes issue cycles from the other
        //  three's MFMAs -- three-input maxima without the canonicalising self-maxima fmaxf brings, packed fma / add)
        constexpr float RESCALE = 10.f;
        float tmax = max3(max3(max3(S[0][0], S[0][1], S[0][2]), S[0][3], S[1][0]), S[1][1], S[1][2]);
        tmax = quad_max(fmax_raw(tmax, S[1][3])) * c;
        float m_new = m;
        if (__builtin_amdgcn_ballot_w64(tmax > m + RESCALE) != 0) {
            m_new = fmaxf(m, tmax);
            const float alpha = (MASK && m == -INFINITY) ? 0.f : fast_exp2(m - m_new);
            l *= alpha;
#pragma unroll
            for (int x = 0; x < NC; ++x)
#pragma unroll
                for (int e = 0; e < 4; ++e) O[x][e] *= alpha;
        }
        const float m_use = (MASK && m_new == -INFINITY) ? 0.f : m_new;
        const f32x2 c2 = {c, c}, m2 = {-m_use, -m_use};
        f32x2 ps = {0.f, 0.f};
#pragma unroll
        for (int g2 = 0; g2 < 2; ++g2)
#pragma unroll
            for (int r = 0; r < 4; r += 2) {
                const f32x2 a = __builtin_elementwise_fma(f32x2{S[g2][r], S[g2][r + 1]}, c2, m2);
                const f32x2 e = {fast_exp2(a.x), fast_exp2(a.y)};
                S[g2][r] = e.x;
                S[g2][r + 1] = e.y;
                ps += e;
            }
        l += quad_sum(ps.x + ps.y);
        m = m_new;
        FENCE();
        // ---- O^T[d, q] += V^T[d, kv] P^T[kv, q]: 8 steps (one key row per lane quarter each) of (NV vector reads, NC MFMAs)
        if (!EARLY_V) {
#pragma unroll
            for (int v = 0; v < NV; ++v) ldv<VW>(sV + vb[0] + 64 * v, ev[0][v]);
        }
#pragma unroll
        for (int st = 0; st < 8; ++st) {
            const int g2 = st >> 2, r = st & 3;
            if (st + 1 < 8) {
#pragma unroll
                for (int v = 0; v < NV; ++v) ldv<VW>(sV + vb[(st + 1) & 3] + ((st + 1) >> 2) * ROWS16 + 64 * v, ev[(st + 1) & 1][v]);
            }
            FENCE();
#pragma unroll
            for (int x = 0; x < NC; ++x) O[x] = MFMA16(ev[st & 1][x / VW][x % VW], S[g2][r], O[x]);
            FENCE();
        }
    };
    auto next_tile = [&](int t) -> int {
        if (!skipping) return t + 1 < nt ? t + 1 : -1;
        const unsigned long mm = t < 63 ? act >> (t + 1) : 0ul;
        return mm ? t + 1 + __builtin_ctzl(mm) : -1;
    };
    for (int t = t_first; t >= 0;) {
        const int t1 = next_tile(t);
        tile(std::integral_constant<int, 0>{}, t, t1);
        if (t1 < 0) break;
        const int t2 = next_tile(t1);
        tile(std::integral_constant<int, 1>{}, t1, t2);
        t = t2;
    }

    const float inv = 1.f / l;
    if (kk == 0 && qok) p.lse[(long)bh * p.seq_q + qrow] = (m + __builtin_amdgcn_logf(l)) * LN2;       // v_log_f32 is log2
    // register r of the d-tiles of one vector read together: VW adjacent head dimensions from VW (4 kk + r); r = 0 .. 3 continue them
#pragma unroll
    for (int x4 = 0; x4 < (NC + 3) / 4; ++x4)
#pragma unroll
        for (int r0 = 0; r0 < 4; r0 += 4 / VW) {
            const int d0 = 64 * x4 + VW * (4 * kk + r0);
            const int off = qok ? (int)((qrow * p.ctx_pitch + d0) * 4) : OOB;
            float o4[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int x = 4 * x4 + e % VW, r = r0 + e / VW;
                o4[e] = O[x < NC ? x : 0][r & 3] * inv;
            }
            buf_store4(rsrcC, off, o4[0], o4[1], o4[2], o4[3]);
        }
  }   // rep
}

// ---------------------------------------------------------------------------------------------------------------
// Backward.  One block = 4 wavefronts = one (batch, head); one block per CU (145 KB of LDS at D = 128, the whole
// register file per wave).  Outer loop: key blocks of 128 (wave w owns keys 32 w .. 32 w + 31 of the block, its dK^T
// and dV^T accumulators and its V fragment in registers, the K block in LDS); inner loop: query tiles of 32 (Q and
// dO tiles by LDS-DMA, two stages).  dQ is summed over the key blocks by the SAME wave in program order (a plain
// read-modify-write of its own 32 x 32 slice): no atomics, bitwise reproducible.
// The row terms LSE (from the forward) and delta = rowsum(dO * O) (mha_delta_kernel, 8 B/element of [B,S,H,D]) are
// read one tile ahead, one value per lane, and turned from "query on the lane" into "query in the registers" through LDS.
// ---------------------------------------------------------------------------------------------------------------
template <int D, bool MASK, bool SAVED, bool TRACE>
__global__ void __launch_bounds__(256, 1)
mha_bwd_kernel(const MhaArgs p) {
    using T = Tile<D>;
    using TS = Tile<128>;
    constexpr int NG = D / 8, DT = (D + 31) / 32, VEC = T::VEC, QTILE = 32 * D, KBLK = 128 * D;
    constexpr int QPIECES = D / 8, QPPW = (QPIECES + 3) / 4, KPPW = D / 8;       // K block: D / 2 pieces, D / 8 per wave
    // ONE __shared__ object: with several, hipcc's wait-count pass learns which of them an LDS-DMA piece writes and
    // puts `s_waitcnt vmcnt(0)` in front of the next ds_read of that object -- here the read of the CURRENT Q / dO stage
    // right after the NEXT stage's pieces were issued: the whole prefetch drained once per tile.  The pipeline's
    // waits are explicit (counted vmcnt + barrier at the top of the tile).
    __shared__ __attribute__((aligned(16))) float smem[KBLK + 4 * QTILE + 32 * 128 + 4 * 64];
    float *const sKB = smem, *const sQ = sKB + KBLK, *const sDO = sQ + 2 * QTILE, *const sDS = sDO + 2 * QTILE, *const sRow = sDS + 32 * 128;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    long long *const blk_tr = (TRACE && p.trace && tid == 0) ? p.trace + (long)blockIdx.x * 16 : nullptr;   // block-level stamps 9..13
    if (blk_tr) blk_tr[9] = __builtin_amdgcn_s_memtime();
    const int l32 = lane & 31, half = lane >> 5;
    const int dcol = D < 32 ? (l32 & (D - 1)) : l32;
    const int bh = blockIdx.x;
    const int b = bh / p.heads, h = bh - b * p.heads;
    float *xs = sRow + wave * 64;

    const auto descK = make_desc(p.k + (long)b * p.seq_kv * p.k_pitch + h * D, ((long)(p.seq_kv - 1) * p.k_pitch + D) * 4);
    const auto rsrcV = make_rsrc(p.v + (long)b * p.seq_kv * p.v_pitch + h * D, ((long)(p.seq_kv - 1) * p.v_pitch + D) * 4);
    const auto descQ = make_desc(p.q + (long)b * p.seq_q * p.q_pitch + h * D, ((long)(p.seq_q - 1) * p.q_pitch + D) * 4);
    const auto descDO = make_desc(p.dctx + (long)b * p.seq_q * p.dctx_pitch + h * D, ((long)(p.seq_q - 1) * p.dctx_pitch + D) * 4);
    const auto rsrcDQ = make_rsrc(p.dq + (long)b * p.seq_q * p.dq_pitch + h * D, ((long)(p.seq_q - 1) * p.dq_pitch + D) * 4);
    const auto rsrcDK = make_rsrc(p.dk + (long)b * p.seq_kv * p.dk_pitch + h * D, ((long)(p.seq_kv - 1) * p.dk_pitch + D) * 4);
    const auto rsrcDV = make_rsrc(p.dv + (long)b * p.seq_kv * p.dv_pitch + h * D, ((long)(p.seq_kv - 1) * p.dv_pitch + D) * 4);
    // an empty descriptor: every access through it is out of range (loads give 0, stores are dropped)
    const auto rsrcNone = make_rsrc(p.dq, 0);
    // Row terms and saved scores of this (b, h) behind descriptors: a row beyond seq_q (a tile past the end, the
    // prefetch of the last tile) is out of range and reads 0 -- no branch, no exec masking around any of these loads.
    const auto rsrcL = make_rsrc(p.lse + (long)bh * p.seq_q, (long)p.seq_q * 4);
    const auto rsrcDl = make_rsrc(p.delta + (long)bh * p.seq_q, (long)p.seq_q * 4);
    const auto rsrcS = make_rsrc(SAVED ? p.scores + (long)bh * p.seq_q * p.seq_kv : nullptr, SAVED ? (long)p.seq_q * p.seq_kv * 4 : 0);
    const int srow_bytes = p.seq_kv * 4;
    const auto rsrcM = make_rsrc(MASK ? p.mask + b * p.mask_sb + h * p.mask_sh : nullptr, MASK ? (long)(p.seq_q - 1) * p.mask_sq + p.seq_kv : 0);

    unsigned vq[QPPW], vdo[QPPW], vkb[KPPW];
#pragma unroll
    for (int i = 0; i < QPPW; ++i) {
        vq[i] = T::src(lane, wave * QPPW + i, p.q_pitch);
        vdo[i] = T::src(lane, wave * QPPW + i, p.dctx_pitch);
    }
#pragma unroll
    for (int i = 0; i < KPPW; ++i) vkb[i] = T::src(lane, wave * KPPW + i, p.k_pitch);
    const unsigned qstep = (unsigned)(32 * p.q_pitch * 4), dostep = (unsigned)(32 * p.dctx_pitch * 4);
    // The Q pieces (which = 0) or the dO pieces (which = 1) of this wave for tile qt; the tile's first row is the
    // scalar offset of the group.
    const unsigned lds_q = lds_offset(sQ + wave * QPPW * 256), lds_do = lds_offset(sDO + wave * QPPW * 256);
    auto issue_half = [&](int qt, int stage, int which) {
        if (QPIECES % 4 == 0 || wave * QPPW < QPIECES) {     // D = 16: two pieces per tile, waves 0 and 1
            if (which == 0) dma_group<QPPW>(descQ, lds_q + stage * QTILE * 4, qt * qstep, vq);
            else dma_group<QPPW>(descDO, lds_do + stage * QTILE * 4, qt * dostep, vdo);
        }
    };

    const float c = p.scale * LOG2E;
    const int nqt = (p.seq_q + 31) / 32, nkb = (p.seq_kv + 127) / 128;
    const int kvl = 32 * wave + l32;                                             // this lane's key inside the block

    // LDS addresses: lane-dependent bases + immediates (Tile<D>::row_bases / vec_bases / elem_bases)
    int rb[T::NB], vb[2][4], rbs[TS::NB], ebk[2][4], ebs[2][4];
    T::row_bases(l32, half, rb);                         // rows of the Q / dO tiles, and of this wave's K rows
    T::vec_bases(half, l32, vb);                         // column vectors of the Q / dO tiles
    TS::row_bases(l32, half, rbs);                       // rows of the dS tile
    {
#pragma unroll
        for (int par = 0; par < 2; ++par)
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                // row 4 half + k + 8 par (the 8 par rows are added as an immediate at the read), columns 32 wave + ..
                ebk[par][k] = T::elem(4 * half + k + 8 * par, 32 * wave + dcol) - 8 * par * D;       // K block
                ebs[par][k] = TS::elem(4 * half + k + 8 * par, 32 * wave + l32) - 8 * par * 128;     // dS tile
            }
    }
    const float *sKW = sKB + 32 * wave * D;              // this wave's 32 keys of the K block

    // dQ slice of this wave, held TRANSPOSED (dQ^T[d, q] = K^T dS^T): the query q0 + l32 is on the lane and register
    // group r >> 2 is the 4 adjacent head dimensions 32 wave + 8 (r >> 2) + 4 half ..: the read-modify-write of the
    // slice is 4 + 4 sixteen-byte accesses per tile instead of 16 + 16 dwords.  The lane part of the address is one
    // register, the tile's first row a scalar offset (range-checked with it, as in the GEMM epilogue).
    const bool dq_wave = wave < DT;
    const int dq_voff = (int)((l32 * p.dq_pitch + 32 * wave + 4 * half) * 4);
    const auto rsrcDQw = dq_wave ? rsrcDQ : rsrcNone;     // waves without a slice: every access out of range
    // Key-block seams.  Nothing of the next key block waits at the seam: its K block and V fragment are requested in
    // the LAST query tile of the current one (behind a barrier that says every wave is done with the K block, in
    // front of that tile's 64 dK MFMAs), its first Q / dO tile -- tile 0 again -- is the regular "next tile" prefetch
    // of that last tile, and this block's dK / dV stores stay in flight behind a counted wait.
    const unsigned lds_kb = lds_offset(sKB + wave * KPPW * 256);
    auto issue_kblock = [&](int kb) { dma_group<KPPW>(descK, lds_kb, (unsigned)(kb * 128 * p.k_pitch * 4), vkb); };
    float4 vf[NG];                                                               // V[key][8 g + 4 half + s] of the current block
    auto load_vfrag = [&](int kb) {
        const int row = kb * 128 + kvl;
#pragma unroll
        for (int g = 0; g < NG; ++g) {
            vf[g] = buf_load4(rsrcV, row < p.seq_kv ? (int)((row * p.v_pitch + 8 * g + 4 * half) * 4) : OOB);
            // the 1 / sqrt(Dk) of dS = scale P (dP - delta) rides the operand of the dP product (and delta, mha_delta_kernel):
            // NG multiplies per key block instead of 16 per tile
            vf[g].x *= p.scale; vf[g].y *= p.scale; vf[g].z *= p.scale; vf[g].w *= p.scale;
        }
    };
    constexpr int SEAM_STORES = VEC == 4 ? 32 + 4 : 0;   // D = 128: 32 dK/dV stores + the last tile's 4 dQ stores stay in flight
    issue_kblock(0);
    load_vfrag(0);
    issue_half(0, 0, 0);
    issue_half(0, 0, 1);
    int it = 0;                                          // tiles done so far: stage parity
    // Row terms of the next tile (one query per lane), fetched one tile ahead -- across the seams too
    // (put into the wave's 256 bytes of LDS at the END of the tile before -- here for the first one -- so that a
    // tile starts with the reads alone: "query on the lane" becomes "query in the registers")
    float lse_n = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rsrcL, l32 * 4, 0, 0));
    float dlt_n = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rsrcDl, l32 * 4, 0, 0));
    if (half == 0) { xs[l32] = lse_n * LOG2E; xs[32 + l32] = dlt_n; }

    if (blk_tr) blk_tr[10] = __builtin_amdgcn_s_memtime();
    for (int kb = 0; kb < nkb; ++kb) {
        const int kvrow = kb * 128 + kvl;
        const bool kvok = kvrow < p.seq_kv;
        f32x16 dK[DT], dV[DT];
#pragma unroll
        for (int t = 0; t < DT; ++t) { zero16(dK[t]); zero16(dV[t]); }
        const auto rsrcOld = kb > 0 ? rsrcDQw : rsrcNone;                        // first key block: "old" dQ reads as 0

        int in_flight = kb > 0 ? SEAM_STORES : 0;        // youngest vector-memory operations that may stay outstanding

        for (int qt = 0; qt < nqt; ++qt) {
            // Tile qt (and the K block) has landed for every wave; the other stage and sDS are free.  The dQ stores
            // of the previous tile are the youngest 4 vector-memory operations of this wave: they stay in flight.
            if (in_flight == 4) asm volatile("s_waitcnt vmcnt(4) ; npm:wait" ::: "memory");
            else if (in_flight == 36) asm volatile("s_waitcnt vmcnt(36) ; npm:wait" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            const bool seam = qt + 1 == nqt && kb + 1 < nkb;      // last tile of a key block that has a successor
            const int nq = qt + 1 < nqt ? qt + 1 : 0;             // the next tile to prefetch: wraps to the next block's tile 0
            long long *tr = (TRACE && p.trace && tid == 0 && kb == (nkb > 1 ? 1 : 0) && qt == (nqt > 5 ? 5 : 0)) ? p.trace + (long)blockIdx.x * 16 : nullptr;
            if (TRACE && p.trace && tid == 0 && kb == (nkb > 1 ? 1 : 0) && qt == (nqt > 5 ? 6 : 1)) { FENCE(); p.trace[(long)blockIdx.x * 16 + 8] = __builtin_amdgcn_s_memtime(); FENCE(); }
            STAMP(0);
            const int cur = it & 1, nxt = cur ^ 1;
            const float *tQ = sQ + cur * QTILE, *tDO = sDO + cur * QTILE;
            const int q0 = 32 * qt;
            // Row terms, from "query on the lane" to "query in the registers", through 256 bytes of LDS private to the
            // wave; read back at once (a read inside a later phase would wait behind that phase's operand prefetch).
            float Lr[16], Dr[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                Lr[r] = xs[4 * half + (r & 3) + 8 * (r >> 2)];
                Dr[r] = xs[32 + 4 * half + (r & 3) + 8 * (r >> 2)];
            }
            lse_n = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rsrcL, l32 * 4, 32 * nq * 4, 0));
            dlt_n = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rsrcDl, l32 * 4, 32 * nq * 4, 0));
            const int dq_tile = q0 * (int)p.dq_pitch * 4;                         // scalar: byte offset of the tile's first row
            // Old dQ^T values of this wave's slice (head dimensions 8 g + 4 half .. of query q0 + l32; zeros in the first key
            // block): requested now, they become the INITIAL VALUE of the dQ accumulator -- the sum over the key blocks
            // costs no add, and the last of this tile's compiler-visible loads is consumed before the next tile's
            // LDS-DMA pieces go out (a later use would make hipcc's counted wait drain those too).
            f32x16 acc;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                if (8 * g < D) {
                    const u32x4_t o = __builtin_amdgcn_raw_buffer_load_b128(rsrcOld, dq_voff + 32 * g, dq_tile, 0);
                    acc[4 * g] = __uint_as_float(o.x); acc[4 * g + 1] = __uint_as_float(o.y);
                    acc[4 * g + 2] = __uint_as_float(o.z); acc[4 * g + 3] = __uint_as_float(o.w);
                } else {
                    acc[4 * g] = acc[4 * g + 1] = acc[4 * g + 2] = acc[4 * g + 3] = 0.f;
                }
            }
            f32x16 S, P, dP, dS;
            float4 fa[2], fk[2];                       // row fragments, one step ahead
            float ea[2][4];                            // column vectors, one step ahead
            if (SAVED) {
                // raw scores of this tile: lane part (its key, its 4 half rows) in one register, the row a scalar offset;
                // rows beyond seq_q fall out of the descriptor's range and read 0.  Nontemporal (aux = 2): read once.
                const int svoff = kvok ? (kvrow + 4 * half * p.seq_kv) * 4 : OOB;
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    S[r] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rsrcS, svoff, (q0 + (r & 3) + 8 * (r >> 2)) * srow_bytes, 2));
                fa[0] = ld4(tDO + rb[0]);
            } else {
                // ---- S[q, kv] = Q K^T: NG steps of (2 row reads, 4 MFMAs), every read one step ahead of its use;
                //      the next tile's DMA pieces are issued along the way
                zero16(S);
                fa[0] = ld4(tQ + rb[0]);
                fk[0] = ld4(sKW + rb[0]);
#pragma unroll
                for (int g = 0; g < NG; ++g) {
                    if (g + 1 < NG) {
                        fa[(g + 1) & 1] = ld4(tQ + rb[(g + 1) & 7] + T::row_imm(g + 1));
                        fk[(g + 1) & 1] = ld4(sKW + rb[(g + 1) & 7] + T::row_imm(g + 1));
                    } else {
                        fa[(g + 1) & 1] = ld4(tDO + rb[0]);                       // first fragment of the next phase
                    }
                    FENCE();
                    S = MFMA(fa[g & 1].x, fk[g & 1].x, S);
                    S = MFMA(fa[g & 1].y, fk[g & 1].y, S);
                    S = MFMA(fa[g & 1].z, fk[g & 1].z, S);
                    S = MFMA(fa[g & 1].w, fk[g & 1].w, S);
                    FENCE();
                }
            }
            STAMP(1);
            // ---- dP[q, kv] = dO V^T (V fragment in registers); the old dQ values are fetched along the way
            constexpr int F0 = SAVED ? 0 : (NG & 1);   // parity slot that holds dO fragment 0
            zero16(dP);
#pragma unroll
            for (int g = 0; g < NG; ++g) {
                if (g + 1 < NG) fa[(F0 + g + 1) & 1] = ld4(tDO + rb[(g + 1) & 7] + T::row_imm(g + 1));
                else ldv<VEC>(tDO + vb[0][0], ea[0]);                                 // first vector of the next phase
                FENCE();
                dP = MFMA(fa[(F0 + g) & 1].x, vf[g].x, dP);
                dP = MFMA(fa[(F0 + g) & 1].y, vf[g].y, dP);
                dP = MFMA(fa[(F0 + g) & 1].z, vf[g].z, dP);
                dP = MFMA(fa[(F0 + g) & 1].w, vf[g].w, dP);
                FENCE();
            }
            // ---- P = exp(scale S - LSE); dS = scale P (dP - delta), also into LDS for the dQ product
            unsigned char mk[16];
            if (MASK) {                                // mask bytes of this tile: one byte load each, row = scalar offset, no branch
                const int mvoff = kvok ? kvrow + 4 * half * (int)p.mask_sq : OOB;
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    mk[r] = __builtin_amdgcn_raw_buffer_load_b8(rsrcM, mvoff, (q0 + (r & 3) + 8 * (r >> 2)) * (int)p.mask_sq, 0);
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                // an excluded position has the score -inf (np.where(mask, scaled, -inf)): P = 0 -- except in a row with NO allowed
                // key, whose LSE is -inf too: P = NaN there, as NumPy's softmax of a row of -inf, and as the saved scores (which
                // carry the -inf) give it.  (a key or query beyond the end reads 0 too: its P is never used)
                const float pr = fast_exp2(fmaf((MASK && mk[r] == 0) ? -INFINITY : S[r], c, -Lr[r]));
                P[r] = pr;
                dS[r] = pr * (dP[r] - Dr[r]);                 // dP and delta carry the 1 / sqrt(Dk) already
                sDS[ebs[(r >> 2) & 1][r & 3] + 8 * (r >> 2) * 128] = dS[r];
            }
            STAMP(2);
            // ---- dV^T[d, kv] += dO^T[d, q] P[q, kv]: 16 steps (one query row each) of (1 vector read, DT MFMAs);
            //      tile t row `lane` is head dimension VEC lane + t
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                if (r + 1 < 16) ldv<VEC>(tDO + vb[((r + 1) >> 2) & 1][(r + 1) & 3] + T::vec_imm(r + 1), ea[(r + 1) & 1]);
                FENCE();
#pragma unroll
                for (int t = 0; t < DT; ++t) dV[t] = MFMA(ea[r & 1][t], P[r], dV[t]);
                FENCE();
            }
            STAMP(3);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();                                      // dS of all four key groups is in LDS
            asm volatile("" ::: "memory");
            STAMP(4);

            // ---- dQ[q, d] (+)= dS[q, kv] K[kv, d] over the 128 keys of the block; wave w takes the columns 32 w .. 32 w + 31
            if (dq_wave) {
                float4 da[2];
                float dk4[2][4];
                da[0] = ld4(sDS + rbs[0]);
#pragma unroll
                for (int k = 0; k < 4; ++k) dk4[0][k] = sKB[ebk[0][k]];
#pragma unroll
                for (int g = 0; g < 16; ++g) {
                    if (g + 1 < 16) {
                        da[(g + 1) & 1] = ld4(sDS + rbs[(g + 1) & 7] + TS::row_imm(g + 1));
#pragma unroll
                        for (int k = 0; k < 4; ++k) dk4[(g + 1) & 1][k] = sKB[8 * (g + 1) * D + ebk[(g + 1) & 1][k]];
                    } else {
                        ldv<VEC>(tQ + vb[0][0], ea[0]);                               // first vector of the next phase
                    }
                    // The next tile's Q / dO pieces go out HERE: behind every load whose result this tile still waits for
                    // (hipcc's counted waits do not see them), half a tile ahead of their use.
                    if (g == 1) issue_half(nq, nxt, 0);
                    if (g == 9) issue_half(nq, nxt, 1);
                    FENCE();
                    acc = MFMA(dk4[g & 1][0], da[g & 1].x, acc);      // transposed: rows = head dimension, column = query
                    acc = MFMA(dk4[g & 1][1], da[g & 1].y, acc);
                    acc = MFMA(dk4[g & 1][2], da[g & 1].z, acc);
                    acc = MFMA(dk4[g & 1][3], da[g & 1].w, acc);
                    FENCE();
                }
            } else {
                issue_half(nq, nxt, 0);
                issue_half(nq, nxt, 1);
                ldv<VEC>(tQ + vb[0][0], ea[0]);
            }
            STAMP(5);
            if (seam) {                                // every wave is done with this K block: request the next one
                __builtin_amdgcn_s_barrier();
                asm volatile("" ::: "memory");
                issue_kblock(kb + 1);
                load_vfrag(kb + 1);
            }
            // ---- dK^T[d, kv] += Q^T[d, q] dS[q, kv]; the dQ stores ride along, one per step
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                if (r + 1 < 16) ldv<VEC>(tQ + vb[((r + 1) >> 2) & 1][(r + 1) & 3] + T::vec_imm(r + 1), ea[(r + 1) & 1]);
                if ((r & 3) == 0 && 8 * (r >> 2) < D) {
                    const int g4 = r >> 2;
                    const u32x4_t v = {__float_as_uint(acc[r]), __float_as_uint(acc[r + 1]), __float_as_uint(acc[r + 2]), __float_as_uint(acc[r + 3])};
                    __builtin_amdgcn_raw_buffer_store_b128(v, rsrcDQw, dq_voff + 32 * g4, dq_tile, 0);
                }
                FENCE();
#pragma unroll
                for (int t = 0; t < DT; ++t) dK[t] = MFMA(ea[r & 1][t], dS[r], dK[t]);
                FENCE();
            }
            // the next tile's row terms (requested at the top of this one) go to LDS now: the reads at the top of this
            // tile are long done, and the next tile starts without a write -> read round trip
            if (half == 0) { xs[l32] = lse_n * LOG2E; xs[32 + l32] = dlt_n; }
            in_flight = 4;
            ++it;
            STAMP(6);
            STAMP(7);
            FENCE();
        }
        if (blk_tr && kb == 0) blk_tr[11] = __builtin_amdgcn_s_memtime();
        if (SEAM_STORES == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        // this block's dK and dV rows: lane = key; register r of the DT tiles together is VEC adjacent head dimensions
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int d0 = VEC * (4 * half + (r & 3) + 8 * (r >> 2));
            if (VEC * (8 * (r >> 2) + (r & 3)) < D) {
                const bool ok = kvok && d0 < D;
                const int offk = ok ? (int)((kvrow * p.dk_pitch + d0) * 4) : OOB, offv = ok ? (int)((kvrow * p.dv_pitch + d0) * 4) : OOB;
                if (VEC == 4) {
                    buf_store4(rsrcDK, offk, dK[0][r], dK[DT > 1 ? 1 : 0][r], dK[DT > 2 ? 2 : 0][r], dK[DT > 3 ? 3 : 0][r]);
                    buf_store4(rsrcDV, offv, dV[0][r], dV[DT > 1 ? 1 : 0][r], dV[DT > 2 ? 2 : 0][r], dV[DT > 3 ? 3 : 0][r]);
                } else {
#pragma unroll
                    for (int t = 0; t < DT; ++t) {
                        __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(dK[t][r]), rsrcDK, ok ? offk + 4 * t : OOB, 0, 0);
                        __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(dV[t][r]), rsrcDV, ok ? offv + 4 * t : OOB, 0, 0);
                    }
                }
            }
        }
    }
    if (blk_tr) {
        blk_tr[12] = __builtin_amdgcn_s_memtime();
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        blk_tr[13] = __builtin_amdgcn_s_memtime();
    }
}

// ---------------------------------------------------------------------------------------------------------------
// Backward, head size 128, saved scores: EIGHT wavefronts per block on v_mfma_f32_16x16x4_f32.
// The 32 x 32 kernel above keeps 32 keys per wave and with them the whole register file: one wave per SIMD, and every
// instruction that is not an MFMA (operand reads, waits, the exp / dS arithmetic, address bookkeeping) is time the
// matrix pipe stands still -- 75 % of the cycles of a tile were MFMA cycles (profiles/r03_attn_core.log).  Here a wave
// owns 16 keys: dK^T and dV^T accumulators (128 x 16 each), the V fragment and the S / P / dP / dS tiles of 32 queries
// fit 256 registers, two waves share a SIMD and fill each other's gaps.  Same LDS budget (K block, two Q / dO stages,
// the dS tile), same products, same order of summation over the key blocks for dQ (bitwise reproducible).
//
// 16 x 16 x 4 operand layout: lane l supplies A[i = l & 15][k = l >> 4] and B[k = l >> 4][j = l & 15]; the result has
// its column j = l & 15 on the lane and rows 4 (l >> 4) + r in registers r = 0..3.  With kk = l >> 4:
//   dP[q, kv]   = dO[q, d] V^T[d, kv]   A = row q = 16 t + l16 of the dO tile, 16 bytes at d = 16 c + 4 kk (one read per
//                                        4 MFMAs), B = the V fragment in registers (same d order)
//   dV^T[d, kv] += dO^T[d, q] P[q, kv]  A = 16 bytes of row q = 16 t + 4 kk + r at d = 4 l16 (+ 64): element e is row l16
//                                        of d-tile e (+ 4); B = P[t][r]: register for register the result of the exp
//   dK^T[d, kv] += Q^T[d, q] dS[q, kv]  likewise with the Q tile and dS
//   dQ^T[d, q]  += K^T[d, kv] dS^T[kv, q] over the 128 keys of the block, wave w taking d = 16 w .. 16 w + 15:
//                                        A = K[kv = 16 c + 4 kk + s][d] (4-byte reads of the K block), B = row q of the dS
//                                        tile in LDS, 16 bytes at kv = 16 c + 4 kk
// ---------------------------------------------------------------------------------------------------------------

template <bool TRACE>
__global__ void __launch_bounds__(512, 1)
mha_bwd16_kernel(const MhaArgs p) {
    constexpr int D = 128;
    using T = Tile<D>;
    constexpr int QTILE = 32 * D, KBLK = 128 * D, ROWS16 = 16 * D;         // floats
    // one __shared__ object, pieces issued from inline assembly (see mha_bwd_kernel)
    __shared__ __attribute__((aligned(16))) float smem[KBLK + 4 * QTILE + 32 * 128];
    float *const sKB = smem, *const sQ = sKB + KBLK, *const sDO = sQ + 2 * QTILE, *const sDS = sDO + 2 * QTILE;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);          // 0 .. 7
    const int l16 = lane & 15, kk = lane >> 4;
    const int bh = blockIdx.x;
    const int b = bh / p.heads, h = bh - b * p.heads;
    // diagnostics (npm_debug_attn_trace, tools/attn_trace.py --bwd16): s_memtime stamps of thread 0 at the phase boundaries of one
    // tile (key block 1, query tile 5) and at the block's start / end
    long long *const blk_tr = (TRACE && p.trace && tid == 0) ? p.trace + (long)blockIdx.x * 16 : nullptr;
    if (blk_tr) blk_tr[9] = __builtin_amdgcn_s_memtime();

    const auto descK = make_desc(p.k + (long)b * p.seq_kv * p.k_pitch + h * D, ((long)(p.seq_kv - 1) * p.k_pitch + D) * 4);
    const auto rsrcV = make_rsrc(p.v + (long)b * p.seq_kv * p.v_pitch + h * D, ((long)(p.seq_kv - 1) * p.v_pitch + D) * 4);
    const auto descQ = make_desc(p.q + (long)b * p.seq_q * p.q_pitch + h * D, ((long)(p.seq_q - 1) * p.q_pitch + D) * 4);
    const auto descDO = make_desc(p.dctx + (long)b * p.seq_q * p.dctx_pitch + h * D, ((long)(p.seq_q - 1) * p.dctx_pitch + D) * 4);
    const auto rsrcDQ = make_rsrc(p.dq + (long)b * p.seq_q * p.dq_pitch + h * D, ((long)(p.seq_q - 1) * p.dq_pitch + D) * 4);
    const auto rsrcDK = make_rsrc(p.dk + (long)b * p.seq_kv * p.dk_pitch + h * D, ((long)(p.seq_kv - 1) * p.dk_pitch + D) * 4);
    const auto rsrcDV = make_rsrc(p.dv + (long)b * p.seq_kv * p.dv_pitch + h * D, ((long)(p.seq_kv - 1) * p.dv_pitch + D) * 4);
    const auto rsrcNone = make_rsrc(p.dq, 0);
    // row terms (mha_rowterms_kernel): log2(e) LSE and MINUS delta, padded to whole tiles and finite in the padding, so that a
    // lane's four consecutive queries are one 16-byte load each -- no staging through LDS, no multiply, no exec-masked writes
    const auto rsrcL = make_rsrc(p.lse2 + (long)bh * p.sq_pad, (long)p.sq_pad * 4);
    const auto rsrcDl = make_rsrc(p.delta + b * p.delta_sb + h * p.delta_sh, (long)p.delta_len * 4);
    const auto rsrcS = make_rsrc(p.scores + (long)bh * p.seq_q * p.seq_kv, (long)p.seq_q * p.seq_kv * 4);
    const int srow_bytes = p.seq_kv * 4;

    // ---- LDS-DMA: the Q / dO tiles are 16 pieces each (2 per wave), the K block 64 (8 per wave)
    unsigned vq[2], vdo[2], vkb[8];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        vq[i] = T::src(lane, wave * 2 + i, p.q_pitch);
        vdo[i] = T::src(lane, wave * 2 + i, p.dctx_pitch);
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) vkb[i] = T::src(lane, wave * 8 + i, p.k_pitch);
    const unsigned qstep = (unsigned)(32 * p.q_pitch * 4), dostep = (unsigned)(32 * p.dctx_pitch * 4);
    const unsigned lds_q = lds_offset(sQ + wave * 2 * 256), lds_do = lds_offset(sDO + wave * 2 * 256), lds_kb = lds_offset(sKB + wave * 8 * 256);

    // ---- LDS addresses (floats): lane-dependent bases, everything else is an immediate
    int rb[4], vb[4], ws[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        rb[j] = T::chunk(l16, 4 * j + kk);                 // row reads: row 16 t + l16, chunk 4 c + kk; + t * ROWS16 + (c >> 2) * 64
        vb[j] = T::elem(4 * kk + j, 4 * l16);              // column vectors: row 16 t + 4 kk + r, columns 4 l16 ..; + t * ROWS16 (+ 64)
        ws[j] = T::elem(4 * kk + j, 16 * wave + l16);      // single elements: row 16 x + 4 kk + r, column 16 wave + l16; + x * ROWS16
    }

    const float c = p.scale * LOG2E;
    const int nqt = (p.seq_q + 31) / 32, nkb = (p.seq_kv + 127) / 128;
    const int kvl = 16 * wave + l16;                                            // this lane's key inside the block
    // dQ slice of this wave: head dimensions 16 wave + 4 kk + r of query q0 + 16 t + l16
    const int dq_voff = (int)((l16 * p.dq_pitch + 16 * wave + 4 * kk) * 4);
    const int dq_sub = 16 * (int)p.dq_pitch * 4;                                // byte step between the two query sub-tiles

    float4 vf[8];                                                               // V[key][16 c + 4 kk + s] * scale
    int it = 0;
    dma_group<8>(descK, lds_kb, 0u, vkb);
    {
        const int row = kvl;
#pragma unroll
        for (int cc = 0; cc < 8; ++cc) {
            vf[cc] = buf_load4(rsrcV, row < p.seq_kv ? (int)((row * p.v_pitch + 16 * cc + 4 * kk) * 4) : OOB);
            vf[cc].x *= p.scale; vf[cc].y *= p.scale; vf[cc].z *= p.scale; vf[cc].w *= p.scale;
        }
    }
    dma_group<2>(descQ, lds_q, 0u, vq);
    dma_group<2>(descDO, lds_do, 0u, vdo);

    bool scale_pending = false;
    f32x4 Lr_n[2], Dr_n[2];
    auto load_rows = [&](int qfirst) {
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const u32x4_t l4 = __builtin_amdgcn_raw_buffer_load_b128(rsrcL, 16 * kk, (qfirst + 16 * t) * 4, 0);
            const u32x4_t d4 = __builtin_amdgcn_raw_buffer_load_b128(rsrcDl, 16 * kk, (qfirst + 16 * t) * 4, 0);
            Lr_n[t] = f32x4{__uint_as_float(l4.x), __uint_as_float(l4.y), __uint_as_float(l4.z), __uint_as_float(l4.w)};
            Dr_n[t] = f32x4{__uint_as_float(d4.x), __uint_as_float(d4.y), __uint_as_float(d4.z), __uint_as_float(d4.w)};
        }
    };
    load_rows(0);
    if (blk_tr) blk_tr[10] = __builtin_amdgcn_s_memtime();
    for (int kb = 0; kb < nkb; ++kb) {
        const int kvrow = kb * 128 + kvl;
        const bool kvok = kvrow < p.seq_kv;
        f32x4 dK[8], dV[8];
#pragma unroll
        for (int t = 0; t < 8; ++t) { dK[t] = f32x4{0.f, 0.f, 0.f, 0.f}; dV[t] = f32x4{0.f, 0.f, 0.f, 0.f}; }
        const auto rsrcOld = kb > 0 ? rsrcDQ : rsrcNone;                         // first key block: "old" dQ reads as 0
        const int svoff = kvok ? (kvrow + 4 * kk * p.seq_kv) * 4 : OOB;          // scores: this lane's key, its 4 kk rows
        int in_flight = kb > 0 ? 18 : 0;                 // youngest vector-memory operations that may stay outstanding

        for (int qt = 0; qt < nqt; ++qt) {
            long long *const tr = (TRACE && blk_tr && kb == (nkb > 1 ? 1 : 0) && qt == (nqt > 5 ? 5 : 0)) ? blk_tr : nullptr;
            STAMP(0);
            if (in_flight == 2) asm volatile("s_waitcnt vmcnt(2) ; npm:wait" ::: "memory");
            else if (in_flight == 18) asm volatile("s_waitcnt vmcnt(18) ; npm:wait" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            if (scale_pending) {                         // the V fragment of a new key block: landed (the counted wait above)
#pragma unroll
                for (int cc = 0; cc < 8; ++cc) { vf[cc].x *= p.scale; vf[cc].y *= p.scale; vf[cc].z *= p.scale; vf[cc].w *= p.scale; }
                scale_pending = false;
            }
            STAMP(1);
            const bool seam = qt + 1 == nqt && kb + 1 < nkb;
            const int nq = qt + 1 < nqt ? qt + 1 : 0;
            const int cur = it & 1, nxt = cur ^ 1;
            const float *tQ = sQ + cur * QTILE, *tDO = sDO + cur * QTILE;
            const int q0 = 32 * qt;
            // row terms of this tile (queries 16 t + 4 kk + r in the registers): requested a tile ago; the next tile's now
            f32x4 Lr[2], Dr[2];
#pragma unroll
            for (int t = 0; t < 2; ++t) { Lr[t] = Lr_n[t]; Dr[t] = Dr_n[t]; }
            load_rows(32 * nq);
            // raw scores of this tile (nontemporal: read once) and the old dQ values (the dQ accumulators start from them)
            // (requesting the scores a whole tile ahead, like the row terms, was measured in round 4: 4.56 -> 4.58-4.62 ms --
            //  their trip from HBM is covered by the dP product)
            f32x4 S[2], acc[2];
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    S[t][r] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rsrcS, svoff, (q0 + 16 * t + r) * srow_bytes, 2));
            const int dq_tile = q0 * (int)p.dq_pitch * 4;
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                const u32x4_t o = __builtin_amdgcn_raw_buffer_load_b128(rsrcOld, dq_voff, dq_tile + t * dq_sub, 0);
                acc[t] = f32x4{__uint_as_float(o.x), __uint_as_float(o.y), __uint_as_float(o.z), __uint_as_float(o.w)};
            }

            // ---- dP[q, kv] = dO V^T: 8 steps of (2 row reads, 8 MFMAs), reads one step ahead
            // (the accumulators start from -delta, which has their layout: no zeroing, no subtraction afterwards)
            f32x4 dP[2] = {Dr[0], Dr[1]};
            float4 fa[2][2];
            fa[0][0] = ld4(tDO + rb[0]);
            fa[0][1] = ld4(tDO + rb[0] + ROWS16);
#pragma unroll
            for (int cc = 0; cc < 8; ++cc) {
                if (cc + 1 < 8) {
                    fa[(cc + 1) & 1][0] = ld4(tDO + rb[(cc + 1) & 3] + ((cc + 1) >> 2) * 64);
                    fa[(cc + 1) & 1][1] = ld4(tDO + rb[(cc + 1) & 3] + ((cc + 1) >> 2) * 64 + ROWS16);
                }
                FENCE();
#pragma unroll
                for (int t = 0; t < 2; ++t) {
                    dP[t] = MFMA16(fa[cc & 1][t].x, vf[cc].x, dP[t]);
                    dP[t] = MFMA16(fa[cc & 1][t].y, vf[cc].y, dP[t]);
                    dP[t] = MFMA16(fa[cc & 1][t].z, vf[cc].z, dP[t]);
                    dP[t] = MFMA16(fa[cc & 1][t].w, vf[cc].w, dP[t]);
                }
                FENCE();
            }
            STAMP(2);
            // ---- P = exp(scale S - LSE); dS = P (dP - delta) (both carry the 1 / sqrt(Dk) already), also into LDS
            f32x4 P[2], dS[2];
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float pr = fast_exp2(fmaf(S[t][r], c, -Lr[t][r]));
                    P[t][r] = pr;
                    dS[t][r] = pr * dP[t][r];
                    sDS[ws[r] + t * ROWS16] = dS[t][r];
                }
            STAMP(3);
            // ---- dV^T[d, kv] += dO^T[d, q] P[q, kv]: 8 steps (4 queries each) of (2 vector reads, 8 MFMAs)
            float4 ea[2][2];
            ea[0][0] = ld4(tDO + vb[0]);
            ea[0][1] = ld4(tDO + vb[0] + 64);
#pragma unroll
            for (int st = 0; st < 8; ++st) {
                const int t = st >> 2, r = st & 3;
                if (st + 1 < 8) {
                    ea[(st + 1) & 1][0] = ld4(tDO + vb[(st + 1) & 3] + ((st + 1) >> 2) * ROWS16);
                    ea[(st + 1) & 1][1] = ld4(tDO + vb[(st + 1) & 3] + ((st + 1) >> 2) * ROWS16 + 64);
                }
                FENCE();
                const float4 e0 = ea[st & 1][0], e1 = ea[st & 1][1];
                dV[0] = MFMA16(e0.x, P[t][r], dV[0]); dV[1] = MFMA16(e0.y, P[t][r], dV[1]);
                dV[2] = MFMA16(e0.z, P[t][r], dV[2]); dV[3] = MFMA16(e0.w, P[t][r], dV[3]);
                dV[4] = MFMA16(e1.x, P[t][r], dV[4]); dV[5] = MFMA16(e1.y, P[t][r], dV[5]);
                dV[6] = MFMA16(e1.z, P[t][r], dV[6]); dV[7] = MFMA16(e1.w, P[t][r], dV[7]);
                FENCE();
            }
            STAMP(4);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();                                      // dS of all eight key groups is in LDS
            asm volatile("" ::: "memory");
            STAMP(5);

            // ---- dQ^T[d, q] (+)= K^T[d, kv] dS^T[kv, q] over the 128 keys: 8 steps of (4 element reads, 2 row reads, 8 MFMAs)
            float ak[2][4];
            float4 da[2][2];
#pragma unroll
            for (int s = 0; s < 4; ++s) ak[0][s] = sKB[ws[s]];
            da[0][0] = ld4(sDS + rb[0]);
            da[0][1] = ld4(sDS + rb[0] + ROWS16);
#pragma unroll
            for (int cc = 0; cc < 8; ++cc) {
                if (cc + 1 < 8) {
#pragma unroll
                    for (int s = 0; s < 4; ++s) ak[(cc + 1) & 1][s] = sKB[ws[s] + (cc + 1) * ROWS16];
                    da[(cc + 1) & 1][0] = ld4(sDS + rb[(cc + 1) & 3] + ((cc + 1) >> 2) * 64);
                    da[(cc + 1) & 1][1] = ld4(sDS + rb[(cc + 1) & 3] + ((cc + 1) >> 2) * 64 + ROWS16);
                }
                // the next tile's Q / dO pieces go out here: behind every load this tile still waits for
                if (cc == 1) dma_group<2>(descQ, lds_q + nxt * QTILE * 4, nq * qstep, vq);
                if (cc == 5) dma_group<2>(descDO, lds_do + nxt * QTILE * 4, nq * dostep, vdo);
                FENCE();
#pragma unroll
                for (int t = 0; t < 2; ++t) {
                    acc[t] = MFMA16(ak[cc & 1][0], da[cc & 1][t].x, acc[t]);
                    acc[t] = MFMA16(ak[cc & 1][1], da[cc & 1][t].y, acc[t]);
                    acc[t] = MFMA16(ak[cc & 1][2], da[cc & 1][t].z, acc[t]);
                    acc[t] = MFMA16(ak[cc & 1][3], da[cc & 1][t].w, acc[t]);
                }
                FENCE();
            }
            STAMP(6);
            if (seam) {                                // every wave is done with this K block: request the next one
                __builtin_amdgcn_s_barrier();
                asm volatile("" ::: "memory");
                dma_group<8>(descK, lds_kb, (unsigned)((kb + 1) * 128 * p.k_pitch * 4), vkb);
                const int row = (kb + 1) * 128 + kvl;
#pragma unroll
                for (int cc = 0; cc < 8; ++cc) vf[cc] = buf_load4(rsrcV, row < p.seq_kv ? (int)((row * p.v_pitch + 16 * cc + 4 * kk) * 4) : OOB);
                scale_pending = true;                  // (scaled behind the next tile's barrier: a wait HERE would drain the K block's pieces)
            }
            // ---- dK^T[d, kv] += Q^T[d, q] dS[q, kv]; the dQ stores ride along
            ea[0][0] = ld4(tQ + vb[0]);
            ea[0][1] = ld4(tQ + vb[0] + 64);
#pragma unroll
            for (int st = 0; st < 8; ++st) {
                const int t = st >> 2, r = st & 3;
                if (st + 1 < 8) {
                    ea[(st + 1) & 1][0] = ld4(tQ + vb[(st + 1) & 3] + ((st + 1) >> 2) * ROWS16);
                    ea[(st + 1) & 1][1] = ld4(tQ + vb[(st + 1) & 3] + ((st + 1) >> 2) * ROWS16 + 64);
                }
                if (st == 0 || st == 4) {
                    const int tt = st >> 2;
                    const u32x4_t v = {__float_as_uint(acc[tt][0]), __float_as_uint(acc[tt][1]), __float_as_uint(acc[tt][2]), __float_as_uint(acc[tt][3])};
                    __builtin_amdgcn_raw_buffer_store_b128(v, rsrcDQ, dq_voff, dq_tile + tt * dq_sub, 0);
                }
                FENCE();
                const float4 e0 = ea[st & 1][0], e1 = ea[st & 1][1];
                dK[0] = MFMA16(e0.x, dS[t][r], dK[0]); dK[1] = MFMA16(e0.y, dS[t][r], dK[1]);
                dK[2] = MFMA16(e0.z, dS[t][r], dK[2]); dK[3] = MFMA16(e0.w, dS[t][r], dK[3]);
                dK[4] = MFMA16(e1.x, dS[t][r], dK[4]); dK[5] = MFMA16(e1.y, dS[t][r], dK[5]);
                dK[6] = MFMA16(e1.z, dS[t][r], dK[6]); dK[7] = MFMA16(e1.w, dS[t][r], dK[7]);
                FENCE();
            }
            in_flight = 2;
            ++it;
            FENCE();
            STAMP(7);
        }
        // this block's dK and dV rows: lane = key; registers r of tiles 4 x + 0..3 are 4 adjacent head dimensions
        // 64 x + 16 kk + 4 r + e
#pragma unroll
        for (int x = 0; x < 2; ++x)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int d0 = 64 * x + 16 * kk + 4 * r;
                const int offk = kvok ? (int)((kvrow * p.dk_pitch + d0) * 4) : OOB, offv = kvok ? (int)((kvrow * p.dv_pitch + d0) * 4) : OOB;
                buf_store4(rsrcDK, offk, dK[4 * x][r], dK[4 * x + 1][r], dK[4 * x + 2][r], dK[4 * x + 3][r]);
                buf_store4(rsrcDV, offv, dV[4 * x][r], dV[4 * x + 1][r], dV[4 * x + 2][r], dV[4 * x + 3][r]);
            }
    }
    if (blk_tr) {
        blk_tr[12] = __builtin_amdgcn_s_memtime();
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        blk_tr[13] = __builtin_amdgcn_s_memtime();
    }
}

// ---------------------------------------------------------------------------------------------------------------
// Backward, round 4: EIGHT wavefronts on v_mfma_f32_16x16x4_f32 for EVERY head size and both score modes, ONE workgroup
// barrier per tile, tiles without an allowed position skipped.
//
// mha_bwd16_kernel above (head size 128, saved scores) spends 17.6 % of its wave cycles in s_waitcnt / s_barrier and keeps
// the matrix pipe 82 % busy (profiles/r04_pmc_attn_sq.log): two barriers per tile -- "the tile has landed" at the top and "dS of
// all eight key groups is in LDS" in front of the dQ product -- and at each of them both waves of a SIMD stand still together.
// Here the dQ product of tile j runs at the START of tile j + 1 (software pipelining across tiles: two dS buffers), so the one
// barrier at the top of a tile says both things.  LDS at D = 128: K block 64 KB + two Q and two dO stages 64 KB + two dS tiles
// 32 KB = 160 KB exactly; the row terms no longer pass through LDS (mha_rowterms_kernel stores log2(e) LSE and delta padded to
// whole tiles, and a lane's four consecutive queries are ONE 16-byte buffer load each).
//
// Operand layouts as in mha_bwd16_kernel (lane l: l16 = l & 15, kk = l >> 4), generalised over NC = D / 16:
//   S[q, kv]    = Q K^T (recomputing mode): A = row q of the Q tile, 16 bytes at d = 16 c + 4 kk; B = the wave's own key row
//                 of the K block at the same d
//   dP[q, kv]   = dO V^T: A = row q of the dO tile; B = the V fragment in registers (scaled by 1 / sqrt(Dk))
//   dV^T / dK^T : A = VW = min(4, D / 16) adjacent head dimensions of row q = 16 t + 4 kk + r of the dO / Q tile (element e
//                 belongs to d-tile e), B = P[t][r] / dS[t][r]
//   dQ^T[d, q]  : wave w takes d-slice w % NC (and, for NC < 8, query half w / NC; waves >= 2 NC have no slice)
// Last tile of a key block: its dQ product cannot wait for the next tile (the K block is replaced), so that tile keeps the
// two-barrier form and the next K block / V fragment are requested in front of its dK product, as before.
//
// Tile skipping (masks): p.skip holds one byte per (query tile of 32, key block of 128) whose bit w says "some position of
// the 32 x 16 sub-tile of wave w is allowed" (npm_mha_mask_summary).  A tile whose byte is 0 is not visited at all, a wave
// whose bit is 0 skips its S / dP / dV / dK products (its dS columns are zeros).  Saved scores of skipped tiles are never
// read (the forward does not write them).  dQ rows of tiles no key block visits are zero-filled at the end.
//
// Hand-counted waits (checked against the disassembly by tools/waitcnt_check.py): the LDS-DMA pieces are issued from
// inline assembly and are invisible to the compiler; the wait at the top of a tile is `s_waitcnt vmcnt(N)` with N = the
// number of vector-memory instructions this wave issued AFTER its last piece: the 2 NC dK / dV stores of a key-block seam
// (the K block's pieces and the V fragment's loads are older than those), else 0.  The Q / dO pieces of the next tile are
// issued in the dV product, BEHIND every compiler-visible load of the tile (row terms, scores, mask bytes, old dQ).
// ---------------------------------------------------------------------------------------------------------------
template <int D, bool MASK, bool SAVED>
__global__ void __launch_bounds__(512, D <= 32 ? 4 : 2)      // (waves per SIMD) D <= 32: 64 KB of LDS, two blocks per CU: 128 registers per wave
mha_bwd8_kernel(const MhaArgs p) {
    using T = Tile<D>;
    // Head size 64: a key block is TWO groups of 128 keys (KG = 2) -- wave w owns keys 16 w .. of each group.  With one group a tile
    // of this head size is 128 MFMAs per wave under the same barrier, waits, row-term loads and dQ read-modify-write as the 256 of head
    // size 128 (0.67 against 0.78 of the peak); with two the block has exactly the footprint of head size 128 (64 KB of K, 96
    // accumulator / V-fragment registers, 160 KB of LDS), 256 MFMAs per tile, and half the key-block passes over Q, dO and dQ.
    constexpr int KG = D == 64 ? 2 : 1;
    constexpr int SP = 128 * KG;                                                 // keys of a block = floats of a dS tile row
    using TS = Tile<SP>;
    constexpr int NC = D / 16;
    constexpr int QTILE = 32 * D, KBLK = SP * D, ROWS16 = 16 * D, SROWS16 = 16 * SP, DSBUF = 32 * SP;       // floats
    constexpr int QP = D / 8, QPPW = QP >= 8 ? QP / 8 : 1, KPPW = D / 16 * KG;   // 1 KiB DMA pieces: Q / dO tile, per wave; K block per wave
    constexpr int VW = D >= 64 ? 4 : D / 16, NV = NC / VW;                       // column-vector reads: width, reads per step
    constexpr int DQT = NC == 8 ? 2 : 1;                                         // query halves a wave's dQ slice covers
    constexpr int NB = NC < 4 ? NC : 4;
    constexpr bool CTP = D < 64;                                                 // compile-time stage / buffer parity (see `visit`)
    // Head size 16: the dQ product has only two 16 x 16 output tiles -- two of the eight waves carried it (32 MFMAs beside the 32
    // of their other four products; their SIMDs were the kernel's critical path).  SPREAD: every wave multiplies its OWN 16 keys'
    // dS (its own columns of the dS tile: no other wave reads them) into a partial dQ^T tile, 8 MFMAs, writes it to LDS, and
    // behind the next barrier all 512 threads add the eight partials of one (query, d) element each, in wave order.
    // (Head size 32 the same way: 6.6 -> 8.5 ms -- its four dQ waves sit on four different SIMDs already, and the partial tiles are
    // twice as large: 16 reads per thread behind the barrier.)
    constexpr bool SPREAD = NC == 1;
    constexpr int PQBUF = 8 * 32 * D;                                            // SPREAD: eight partial [32 q][D] tiles (floats)
    static_assert(!(MASK && SAVED), "saved scores carry the mask");
    __shared__ __attribute__((aligned(16))) float smem[KBLK + 4 * QTILE + (SPREAD ? DSBUF + 2 * PQBUF : 2 * DSBUF)];
    float *const sKB = smem, *const sQ = sKB + KBLK, *const sDO = sQ + 2 * QTILE, *const sDS = sDO + 2 * QTILE;
    float *const sPQ = sDS + DSBUF;                                              // SPREAD: [2][8][32][D]

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);          // 0 .. 7
    const int l16 = lane & 15, kk = lane >> 4;
    const int bh = blockIdx.x;
    const int b = bh / p.heads, h = bh - b * p.heads;

    const auto descK = make_desc(p.k + (long)b * p.seq_kv * p.k_pitch + h * D, ((long)(p.seq_kv - 1) * p.k_pitch + D) * 4);
    const auto rsrcV = make_rsrc(p.v + (long)b * p.seq_kv * p.v_pitch + h * D, ((long)(p.seq_kv - 1) * p.v_pitch + D) * 4);
    const auto descQ = make_desc(p.q + (long)b * p.seq_q * p.q_pitch + h * D, ((long)(p.seq_q - 1) * p.q_pitch + D) * 4);
    const auto descDO = make_desc(p.dctx + (long)b * p.seq_q * p.dctx_pitch + h * D, ((long)(p.seq_q - 1) * p.dctx_pitch + D) * 4);
    const auto rsrcDQ = make_rsrc(p.dq + (long)b * p.seq_q * p.dq_pitch + h * D, ((long)(p.seq_q - 1) * p.dq_pitch + D) * 4);
    const auto rsrcDK = make_rsrc(p.dk + (long)b * p.seq_kv * p.dk_pitch + h * D, ((long)(p.seq_kv - 1) * p.dk_pitch + D) * 4);
    const auto rsrcDV = make_rsrc(p.dv + (long)b * p.seq_kv * p.dv_pitch + h * D, ((long)(p.seq_kv - 1) * p.dv_pitch + D) * 4);
    const auto rsrcNone = make_rsrc(p.dq, 0);
    // row terms, padded to whole tiles and finite in the padding (mha_rowterms_kernel): no straddling 16-byte load
    const auto rsrcL = make_rsrc(p.lse2 + (long)bh * p.sq_pad, (long)p.sq_pad * 4);
    const auto rsrcDl = make_rsrc(p.delta + b * p.delta_sb + h * p.delta_sh, (long)p.delta_len * 4);
    const auto rsrcS = make_rsrc(SAVED ? p.scores + (long)bh * p.seq_q * p.seq_kv : nullptr, SAVED ? (long)p.seq_q * p.seq_kv * 4 : 0);
    const int srow_bytes = p.seq_kv * 4;
    const auto rsrcM = make_rsrc(MASK ? p.mask + b * p.mask_sb + h * p.mask_sh : nullptr, MASK ? (long)(p.seq_q - 1) * p.mask_sq + p.seq_kv : 0);

    // ---- LDS-DMA lane offsets
    unsigned vq[QPPW], vdo[QPPW], vkb[KPPW];
#pragma unroll
    for (int i = 0; i < QPPW; ++i) {
        vq[i] = T::src(lane, wave * QPPW + i, p.q_pitch);
        vdo[i] = T::src(lane, wave * QPPW + i, p.dctx_pitch);
    }
#pragma unroll
    for (int i = 0; i < KPPW; ++i) vkb[i] = T::src(lane, wave * KPPW + i, p.k_pitch);
    const bool q_owner = QP >= 8 || wave < QP;                                  // D < 64: the tile has fewer pieces than waves
    const unsigned qstep = (unsigned)(32 * p.q_pitch * 4), dostep = (unsigned)(32 * p.dctx_pitch * 4);
    const unsigned lds_q = lds_offset(sQ + wave * QPPW * 256), lds_do = lds_offset(sDO + wave * QPPW * 256), lds_kb = lds_offset(sKB + wave * KPPW * 256);
    auto issue_q = [&](int qt, int stage) { if (q_owner) dma_group<QPPW>(descQ, lds_q + stage * QTILE * 4, (unsigned)qt * qstep, vq); };
    auto issue_do = [&](int qt, int stage) { if (q_owner) dma_group<QPPW>(descDO, lds_do + stage * QTILE * 4, (unsigned)qt * dostep, vdo); };

    // ---- LDS addresses (floats): lane-dependent bases, everything else an immediate
    int rb[NB], vb[4], rbS[4], wsS[4], wsK[4];
    const int dqs = NC == 8 ? wave : wave % NC;                                 // d-slice of this wave's dQ rows
    const int dqh = NC == 8 ? 0 : wave / NC;                                    // ... and (NC < 8) its query half
    const bool dq_wave = NC == 8 || wave < 2 * NC;
#pragma unroll
    for (int j = 0; j < NB; ++j) rb[j] = T::chunk(l16, 4 * j + kk);             // rows 16 t + l16, chunk 4 c + kk: + t ROWS16 + (c >> 2) 64
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        vb[j] = T::elem(4 * kk + j, VW * l16);                                  // column vectors: row 16 t + 4 kk + r; + t ROWS16 (+ 64)
        rbS[j] = TS::chunk(l16, 4 * j + kk);                                    // dS tile rows 16 t + l16; + t SROWS16 + (c >> 2) 64
        wsS[j] = TS::elem(4 * kk + j, 16 * wave + l16);                         // dS elements: row 16 t + 4 kk + r, this lane's key (+ 128 kg)
        wsK[j] = T::elem(4 * kk + j, 16 * dqs + l16);                           // K block: row 16 c + 4 kk + s, column of the dQ slice
    }
    const float *const sKW = sKB + 16 * wave * D;                               // this wave's 16 key rows (of group kg: + 128 kg D)

    const float c = p.scale * LOG2E;
    const int nqt = (p.seq_q + 31) / 32, nkb = (p.seq_kv + SP - 1) / SP;
    const int kvl = 16 * wave + l16;
    const int dq_voff = SPREAD ? (int)(((tid >> 4) * p.dq_pitch + (tid & 15)) * 4)          // SPREAD: thread = (query tid / 16, d tid % 16)
                               : (int)((l16 * p.dq_pitch + 16 * dqs + 4 * kk) * 4);
    const int dq_sub = 16 * (int)p.dq_pitch * 4;
    // SPREAD: this wave's own keys -- dS rows l16 (+ 16 t) at the chunk of keys 16 wave + 4 kk, K rows 16 wave + 4 kk + s at column l16,
    // the lane's slot of the wave's partial tile
    // (the partial tile's 16-byte chunks are swizzled by the query, chunk position c ^ ((q >> 1) & 3): ds_write_b128 is served in groups
    // of eight neighbouring lanes over 32 banks, and eight queries' chunk c straight would be a 4-way conflict -- 48 extra LDS cycles
    // per wave and tile, measured)
    static_assert(!SPREAD || D == 16, "the partial tile's swizzle is written for 16-float rows");
    const int own_ds = TS::chunk(l16, 4 * wave + kk), own_pq = wave * 32 * D + l16 * D + ((kk ^ ((l16 >> 1) & 3)) << 2);
    int own_k[4];
#pragma unroll
    for (int s = 0; s < 4; ++s) own_k[s] = T::elem(4 * kk + s, l16) + 16 * wave * D;

    // ---- which tiles exist
    const unsigned char *const sk = p.skip ? p.skip + b * p.skip_sb + h * p.skip_sh : nullptr;
    auto load_tiles = [&](int kb, int &bytes) -> unsigned long {                // bit qt: tile (qt, kb) is visited
        if (!sk) return ~0ul;
        int ln = lane;
        asm volatile("" : "+v"(ln));                                            // (once per key block: not worth a register pair held, or spilled, across the tile loop)
        bytes = 0;                                                              // per key group kg: bits 16 kg .. "any", 16 kg + 8 .. "all"
#pragma unroll
        for (int kg = 0; kg < KG; ++kg) {
            const int k128 = KG * kb + kg;                                      // the summary's key blocks are 128 keys
            if (ln < nqt && k128 < p.skip_nkb) {
                int bb = (int)sk[(long)ln * p.skip_nkb + k128];
                if (MASK && p.skip_all) bb |= (int)sk[p.skip_all + (long)ln * p.skip_nkb + k128] << 8;
                bytes |= bb << (16 * kg);
            }
        }
        const unsigned long any = __builtin_amdgcn_ballot_w64((bytes & 0x00ff00ff) != 0);
        return any ? any : 1ul;                                                 // an empty key block still visits tile 0 (all waves idle in it)
    };
    auto first_tile = [&](unsigned long act) -> int { return sk ? __builtin_ctzl(act) : 0; };
    auto next_tile = [&](unsigned long act, int from) -> int {
        if (!sk) return from < nqt ? from : -1;
        if (from >= 64) return -1;
        const unsigned long m = act >> from;
        return m ? from + __builtin_ctzl(m) : -1;
    };

    float4 vf[KG][NC];                                                          // V[key][16 c + 4 kk + s] * scale of the current key block
    auto load_vfrag = [&](int kb) {
#pragma unroll
        for (int kg = 0; kg < KG; ++kg) {
            const int row = kb * SP + 128 * kg + kvl;
#pragma unroll
            for (int cc = 0; cc < NC; ++cc) {
                vf[kg][cc] = buf_load4(rsrcV, row < p.seq_kv ? (int)((row * p.v_pitch + 16 * cc + 4 * kk) * 4) : OOB);
                vf[kg][cc].x *= p.scale; vf[kg][cc].y *= p.scale; vf[kg][cc].z *= p.scale; vf[kg][cc].w *= p.scale;
            }
        }
    };

    int cur_bytes = 0x00ff00ff, nxt_bytes = 0x00ff00ff;
    unsigned long act = load_tiles(0, cur_bytes);
    unsigned long written = 0;                                                  // bit qt: some key block has stored dQ rows of tile qt
    dma_group<KPPW>(descK, lds_kb, 0u, vkb);
    load_vfrag(0);
    {
        const int q_first = first_tile(act);
        issue_q(q_first, 0);
        issue_do(q_first, 0);
    }
    int j = 0;                                                                  // tiles visited: stage and dS-buffer parity
    int n_inflight = 0;                                                         // vector-memory instructions issued after the last DMA piece

    f32x4 dK[KG][NC], dV[KG][NC], P[2], dS[KG][2];
    // ---- dQ^T[d, q] (+)= K^T[d, kv] dS^T[kv, q] of tile `tile` over the 128 keys of the block, from dS buffer `buf`;
    //      `oldq` = what earlier key blocks left there; the DMA pieces of tile `nq` (stage `nstage`) go out inside
    auto dq_phase = [&](int tile, int buf, const f32x4 (&oldq)[DQT]) __attribute__((always_inline)) {
        if constexpr (SPREAD) {
            const float *pp = sPQ + buf * PQBUF + ((tid & ~15) | ((((tid >> 2) ^ (tid >> 5)) & 3) << 2) | (tid & 3));
            float part[8];
#pragma unroll
            for (int w = 0; w < 8; ++w) part[w] = pp[w * 32 * D];
            float sum = oldq[0][0];
#pragma unroll
            for (int w = 0; w < 8; ++w) sum += part[w];
            __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(sum), rsrcDQ, dq_voff, tile * 32 * (int)p.dq_pitch * 4, 0);
        } else if (dq_wave) {
            const float *tS = sDS + buf * DSBUF + (NC == 8 ? 0 : dqh * SROWS16);
            f32x4 acc[DQT];
#pragma unroll
            for (int t = 0; t < DQT; ++t) acc[t] = oldq[t];
            float ak[2][4];
            float4 da[2][DQT];
#pragma unroll
            for (int s = 0; s < 4; ++s) ak[0][s] = sKB[wsK[s]];
#pragma unroll
            for (int t = 0; t < DQT; ++t) da[0][t] = ld4(tS + rbS[0] + t * SROWS16);
#pragma unroll
            for (int cc = 0; cc < 8 * KG; ++cc) {
                if (cc + 1 < 8 * KG) {
#pragma unroll
                    for (int s = 0; s < 4; ++s) ak[(cc + 1) & 1][s] = sKB[wsK[s] + (cc + 1) * ROWS16];
#pragma unroll
                    for (int t = 0; t < DQT; ++t) da[(cc + 1) & 1][t] = ld4(tS + rbS[(cc + 1) & 3] + ((cc + 1) >> 2) * 64 + t * SROWS16);
                }
                FENCE();
#pragma unroll
                for (int t = 0; t < DQT; ++t) {
                    acc[t] = MFMA16(ak[cc & 1][0], da[cc & 1][t].x, acc[t]);
                    acc[t] = MFMA16(ak[cc & 1][1], da[cc & 1][t].y, acc[t]);
                    acc[t] = MFMA16(ak[cc & 1][2], da[cc & 1][t].z, acc[t]);
                    acc[t] = MFMA16(ak[cc & 1][3], da[cc & 1][t].w, acc[t]);
                }
                FENCE();
            }
            const int dq_tile = tile * 32 * (int)p.dq_pitch * 4 + (NC == 8 ? 0 : dqh * dq_sub);
#pragma unroll
            for (int t = 0; t < DQT; ++t) {
                const u32x4_t v = {__float_as_uint(acc[t][0]), __float_as_uint(acc[t][1]), __float_as_uint(acc[t][2]), __float_as_uint(acc[t][3])};
                __builtin_amdgcn_raw_buffer_store_b128(v, rsrcDQ, dq_voff, dq_tile + t * dq_sub, 0);
            }
        }
    };

    for (int kb = 0; kb < nkb; ++kb) {
        unsigned long act_n = 0;
        if (kb + 1 < nkb) act_n = load_tiles(kb + 1, nxt_bytes);
        int kvrow[KG], svoff[KG], mvoff[KG], kvoob[KG];                         // kvoob: 0, or OOB for a key beyond the sequence (or-ed into store offsets)
        bool kvok[KG];
        int kkm = kk;
        if (MASK) asm volatile("" : "+v"(kkm));                                  // (the product below once per key block, not a register held across the tile loop)
#pragma unroll
        for (int kg = 0; kg < KG; ++kg) {
            kvrow[kg] = kb * SP + 128 * kg + kvl;
            kvok[kg] = kvrow[kg] < p.seq_kv;
            kvoob[kg] = kvok[kg] ? 0 : OOB;
            if (KG > 1) asm volatile("" : "+v"(kvoob[kg]));                      // (KG = 2: a value, not a condition -- with two conditions hipcc turned the stores below
                                                                                 //  into exec-masked branches, and the seam's wait counts store instructions)
#pragma unroll
            for (int t = 0; t < NC; ++t) { dK[kg][t] = f32x4{0.f, 0.f, 0.f, 0.f}; dV[kg][t] = f32x4{0.f, 0.f, 0.f, 0.f}; }
            svoff[kg] = kvok[kg] ? (kvrow[kg] + 4 * kk * p.seq_kv) * 4 : OOB;    // saved scores: this lane's key, its rows 4 kk ..
            mvoff[kg] = kvok[kg] ? kvrow[kg] + 4 * kkm * (int)p.mask_sq : OOB;   // mask bytes, the same way
        }

        int pend = -1;                                                           // tile whose dQ product is still to come
        f32x4 pend_old[DQT];
#pragma unroll
        for (int t = 0; t < DQT; ++t) pend_old[t] = f32x4{0.f, 0.f, 0.f, 0.f};
        int qt = first_tile(act);
        // One visited tile.  Head sizes below 128: the parity of the visit count (Q / dO stage, dS buffer) is a compile-time
        // constant -- two copies of the body -- so that every LDS address is a lane-dependent register plus an immediate (the
        // run-time parity costs 21 vector-ALU adds per tile, each of which takes the matrix pipe's issue slot for its cycles).
        // At head size 128 the register file has no room for the second copy's hoisted bases (it spilled 100 registers):
        // run-time parity there.
        auto visit = [&](auto parity) __attribute__((always_inline)) {
            const int cur = CTP ? (int)decltype(parity)::value : (j & 1);
            const int nxt = next_tile(act, qt + 1);
            const bool last = nxt < 0;                                           // last tile of this key block
            const int nq = !last ? nxt : (kb + 1 < nkb ? first_tile(act_n) : -1);   // the tile whose Q / dO pieces go out in this one
            // ---- the tile's pieces (and, at a seam, the K block) have landed for every wave; dS of the pending tile is in LDS
            if (n_inflight == 2 * NC * KG) asm volatile("s_waitcnt vmcnt(%0) ; npm:wait" :: "n"(2 * NC * KG) : "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            n_inflight = 0;
            const float *tQ = sQ + cur * QTILE, *tDO = sDO + cur * QTILE;
            float *const tDS = sDS + (SPREAD ? 0 : cur * DSBUF);
            const int q0 = 32 * qt;
            const int tile_bits = sk ? __builtin_amdgcn_readlane(cur_bytes, qt & 63) : 0x00ff00ff;
            bool on[KG], masked_sub[KG];
#pragma unroll
            for (int kg = 0; kg < KG; ++kg) {
                on[kg] = !sk || ((tile_bits >> (16 * kg + wave)) & 1);               // this wave's 32 x 16 sub-tile of key group kg has work
                //            (written with the `!sk ||`: without it hipcc's allocation of this kernel at D = 128 ends 18 registers
                //             higher and spills -- the register file is full here, any change to this kernel needs the metadata test)
                masked_sub[kg] = MASK && !((tile_bits >> (16 * kg + 8 + wave)) & 1);  // ... and excluded positions (else: no mask bytes)
            }
            const bool any_on = KG == 1 ? on[0] : (on[0] || on[KG - 1]);

            // ---- requests of this tile: row terms, raw scores / mask bytes, what earlier key blocks left in its dQ rows
            f32x4 Lr[2], Dr[2], S[KG][2];
            unsigned char mk[KG][2][4];
            if (any_on) {
#pragma unroll
                for (int t = 0; t < 2; ++t) {
                    const u32x4_t l4 = __builtin_amdgcn_raw_buffer_load_b128(rsrcL, 16 * kk, (q0 + 16 * t) * 4, 0);
                    const u32x4_t d4 = __builtin_amdgcn_raw_buffer_load_b128(rsrcDl, 16 * kk, (q0 + 16 * t) * 4, 0);
                    Lr[t] = f32x4{__uint_as_float(l4.x), __uint_as_float(l4.y), __uint_as_float(l4.z), __uint_as_float(l4.w)};
                    Dr[t] = f32x4{__uint_as_float(d4.x), __uint_as_float(d4.y), __uint_as_float(d4.z), __uint_as_float(d4.w)};
                }
            }
#pragma unroll
            for (int kg = 0; kg < KG; ++kg) {
                if (!on[kg]) continue;
                if (SAVED) {
#pragma unroll
                    for (int t = 0; t < 2; ++t)
#pragma unroll
                        for (int r = 0; r < 4; ++r)
                            S[kg][t][r] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rsrcS, svoff[kg], (q0 + 16 * t + r) * srow_bytes, 2));
                }
                if (masked_sub[kg]) {
#pragma unroll
                    for (int t = 0; t < 2; ++t)
#pragma unroll
                        for (int r = 0; r < 4; ++r)
                            mk[kg][t][r] = __builtin_amdgcn_raw_buffer_load_b8(rsrcM, mvoff[kg], (q0 + 16 * t + r) * (int)p.mask_sq, 0);
                }
            }

            // ---- the pending tile's dQ product
            if (pend >= 0) dq_phase(pend, cur ^ 1, pend_old);          // the tile visited before this one: the other dS buffer
            // (what earlier key blocks left in THIS tile's dQ rows is requested only now: the pending tile's values are consumed,
            // one set of registers serves both; the values are needed a tile from now, or behind this tile's dV product)
            f32x4 own_old[DQT];
#pragma unroll
            for (int t = 0; t < DQT; ++t) own_old[t] = f32x4{0.f, 0.f, 0.f, 0.f};
            if constexpr (SPREAD) {
                const auto rsrcOld = ((written >> (qt & 63)) & 1) || (!sk && kb > 0) ? rsrcDQ : rsrcNone;
                own_old[0][0] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rsrcOld, dq_voff, q0 * (int)p.dq_pitch * 4, 0));
            } else if (dq_wave) {
                const auto rsrcOld = ((written >> (qt & 63)) & 1) || (!sk && kb > 0) ? rsrcDQ : rsrcNone;     // nothing stored yet: reads 0
                const int dq_tile = q0 * (int)p.dq_pitch * 4 + (NC == 8 ? 0 : dqh * dq_sub);
#pragma unroll
                for (int t = 0; t < DQT; ++t) {
                    const u32x4_t o = __builtin_amdgcn_raw_buffer_load_b128(rsrcOld, dq_voff, dq_tile + t * dq_sub, 0);
                    own_old[t] = f32x4{__uint_as_float(o.x), __uint_as_float(o.y), __uint_as_float(o.z), __uint_as_float(o.w)};
                }
            }
            if (sk) written |= 1ul << (qt & 63);

#pragma unroll
          for (int kg = 0; kg < KG; ++kg) {                                  // (KG = 2: the phases below once per key group; the pieces go out in the last)
            const bool pieces = kg == KG - 1;
            if (on[kg]) {
                float4 fa[2][2];
                if (!SAVED) {
                    // ---- S[q, kv] = Q K^T: NC steps of (3 row reads, 8 MFMAs), reads one step ahead
                    S[kg][0] = f32x4{0.f, 0.f, 0.f, 0.f}; S[kg][1] = f32x4{0.f, 0.f, 0.f, 0.f};
                    float4 fk[2];
                    fa[0][0] = ld4(tQ + rb[0]);
                    fa[0][1] = ld4(tQ + rb[0] + ROWS16);
                    fk[0] = ld4(sKW + 128 * kg * D + rb[0]);
#pragma unroll
                    for (int cc = 0; cc < NC; ++cc) {
                        if (cc + 1 < NC) {
                            fa[(cc + 1) & 1][0] = ld4(tQ + rb[(cc + 1) & 3] + ((cc + 1) >> 2) * 64);
                            fa[(cc + 1) & 1][1] = ld4(tQ + rb[(cc + 1) & 3] + ((cc + 1) >> 2) * 64 + ROWS16);
                            fk[(cc + 1) & 1] = ld4(sKW + 128 * kg * D + rb[(cc + 1) & 3] + ((cc + 1) >> 2) * 64);
                        }
                        FENCE();
#pragma unroll
                        for (int t = 0; t < 2; ++t) {
                            S[kg][t] = MFMA16(fa[cc & 1][t].x, fk[cc & 1].x, S[kg][t]);
                            S[kg][t] = MFMA16(fa[cc & 1][t].y, fk[cc & 1].y, S[kg][t]);
                            S[kg][t] = MFMA16(fa[cc & 1][t].z, fk[cc & 1].z, S[kg][t]);
                            S[kg][t] = MFMA16(fa[cc & 1][t].w, fk[cc & 1].w, S[kg][t]);
                        }
                        FENCE();
                    }
                }
                // ---- dP[q, kv] - delta[q] = dO V^T - delta: NC steps of (2 row reads, 8 MFMAs), reads one step ahead.  The
                //      accumulators START from -delta (the row term has the accumulator's layout: rows 4 kk + r): no zeroing,
                //      no subtraction afterwards
                f32x4 dP[2] = {Dr[0], Dr[1]};
                fa[0][0] = ld4(tDO + rb[0]);
                fa[0][1] = ld4(tDO + rb[0] + ROWS16);
#pragma unroll
                for (int cc = 0; cc < NC; ++cc) {
                    if (cc + 1 < NC) {
                        fa[(cc + 1) & 1][0] = ld4(tDO + rb[(cc + 1) & 3] + ((cc + 1) >> 2) * 64);
                        fa[(cc + 1) & 1][1] = ld4(tDO + rb[(cc + 1) & 3] + ((cc + 1) >> 2) * 64 + ROWS16);
                    }
                    FENCE();
#pragma unroll
                    for (int t = 0; t < 2; ++t) {
                        dP[t] = MFMA16(fa[cc & 1][t].x, vf[kg][cc].x, dP[t]);
                        dP[t] = MFMA16(fa[cc & 1][t].y, vf[kg][cc].y, dP[t]);
                        dP[t] = MFMA16(fa[cc & 1][t].z, vf[kg][cc].z, dP[t]);
                        dP[t] = MFMA16(fa[cc & 1][t].w, vf[kg][cc].w, dP[t]);
                    }
                    FENCE();
                }
                // ---- P = exp2(c S - log2(e) LSE); dS = P (dP - delta) (both carry the 1 / sqrt(Dk) already), also into LDS
#pragma unroll
                for (int t = 0; t < 2; ++t)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        // (an excluded position: score -inf, so P = 0 -- and NaN in a row without any allowed key, see mha_bwd_kernel)
                        const float pr = fast_exp2(fmaf((masked_sub[kg] && mk[kg][t][r] == 0) ? -INFINITY : S[kg][t][r], c, -Lr[t][r]));
                        P[t][r] = pr;
                        dS[kg][t][r] = pr * dP[t][r];
                        tDS[wsS[r] + t * SROWS16 + 128 * kg] = dS[kg][t][r];
                    }
                // ---- dV^T[d, kv] += dO^T[d, q] P[q, kv]: 8 steps (4 queries each) of (NV vector reads, NC MFMAs)
                float ea[2][NV][4];
                float akq[4];
                float4 daq[2];
                if constexpr (SPREAD) {                                          // own dS rows back (transposed), own K rows: used behind dV
#pragma unroll
                    for (int s = 0; s < 4; ++s) akq[s] = sKB[own_k[s]];
#pragma unroll
                    for (int t = 0; t < 2; ++t) daq[t] = ld4(tDS + own_ds + t * SROWS16);
                }
#pragma unroll
                for (int v = 0; v < NV; ++v) ldv<VW>(tDO + vb[0] + 64 * v, ea[0][v]);
#pragma unroll
                for (int st = 0; st < 8; ++st) {
                    const int t = st >> 2, r = st & 3;
                    if (st + 1 < 8) {
#pragma unroll
                        for (int v = 0; v < NV; ++v) ldv<VW>(tDO + vb[(st + 1) & 3] + ((st + 1) >> 2) * ROWS16 + 64 * v, ea[(st + 1) & 1][v]);
                    }
                    // The next tile's Q / dO pieces go out HERE: behind every load whose result this tile still waits for (the
                    // compiler's counted waits do not see them: a wait placed after them would drain them too), a phase and a
                    // half ahead of their use.
                    if (pieces && st == 1 && nq >= 0) issue_q(nq, cur ^ 1);
                    if (pieces && st == 5 && nq >= 0) issue_do(nq, cur ^ 1);
                    FENCE();
#pragma unroll
                    for (int x = 0; x < NC; ++x) dV[kg][x] = MFMA16(ea[st & 1][x / VW][x % VW], P[t][r], dV[kg][x]);
                    FENCE();
                }
                if constexpr (SPREAD) {
                    // ---- partial dQ^T[d, q] = K^T[d, own keys] dS^T[own keys, q]: rows d = 4 kk + r, column q = 16 t + l16
                    f32x4 pq[2];
#pragma unroll
                    for (int t = 0; t < 2; ++t) {
                        pq[t] = f32x4{0.f, 0.f, 0.f, 0.f};
                        pq[t] = MFMA16(akq[0], daq[t].x, pq[t]);
                        pq[t] = MFMA16(akq[1], daq[t].y, pq[t]);
                        pq[t] = MFMA16(akq[2], daq[t].z, pq[t]);
                        pq[t] = MFMA16(akq[3], daq[t].w, pq[t]);
                    }
                    FENCE();
#pragma unroll
                    for (int t = 0; t < 2; ++t)
                        *reinterpret_cast<float4 *>(sPQ + cur * PQBUF + own_pq + t * ROWS16) = make_float4(pq[t][0], pq[t][1], pq[t][2], pq[t][3]);
                }
            } else {
                if constexpr (SPREAD) {
#pragma unroll
                    for (int t = 0; t < 2; ++t)
                        *reinterpret_cast<float4 *>(sPQ + cur * PQBUF + own_pq + t * ROWS16) = make_float4(0.f, 0.f, 0.f, 0.f);
                } else {
#pragma unroll
                for (int t = 0; t < 2; ++t)
#pragma unroll
                    for (int r = 0; r < 4; ++r) tDS[wsS[r] + t * SROWS16 + 128 * kg] = 0.f;
                }
                if (pieces && nq >= 0) {
                    issue_q(nq, cur ^ 1);
                    issue_do(nq, cur ^ 1);
                }
            }
          }

            if (last) {
                // ---- this tile's own dQ product now (the K block is replaced next): dS of all eight key groups must be in LDS
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();
                asm volatile("" ::: "memory");
                dq_phase(qt, cur, own_old);
                if (kb + 1 < nkb) {                    // every wave is done with this K block: request the next one
                    __builtin_amdgcn_s_barrier();
                    asm volatile("" ::: "memory");
                    dma_group<KPPW>(descK, lds_kb, (unsigned)((kb + 1) * SP * p.k_pitch * 4), vkb);
                    load_vfrag(kb + 1);
                }
                pend = -1;
            } else {
                pend = qt;
#pragma unroll
                for (int t = 0; t < DQT; ++t) pend_old[t] = own_old[t];
            }

#pragma unroll
          for (int kg = 0; kg < KG; ++kg)
            if (on[kg]) {
                // ---- dK^T[d, kv] += Q^T[d, q] dS[q, kv]
                float ea[2][NV][4];
#pragma unroll
                for (int v = 0; v < NV; ++v) ldv<VW>(tQ + vb[0] + 64 * v, ea[0][v]);
#pragma unroll
                for (int st = 0; st < 8; ++st) {
                    const int t = st >> 2, r = st & 3;
                    if (st + 1 < 8) {
#pragma unroll
                        for (int v = 0; v < NV; ++v) ldv<VW>(tQ + vb[(st + 1) & 3] + ((st + 1) >> 2) * ROWS16 + 64 * v, ea[(st + 1) & 1][v]);
                    }
                    FENCE();
#pragma unroll
                    for (int x = 0; x < NC; ++x) dK[kg][x] = MFMA16(ea[st & 1][x / VW][x % VW], dS[kg][t][r], dK[kg][x]);
                    FENCE();
                }
            }
            ++j;
            qt = nxt;
            FENCE();
        };
        while (qt >= 0) {
            if (CTP && (j & 1)) visit(std::integral_constant<int, 1>{});
            else visit(std::integral_constant<int, 0>{});
        }
        // ---- this block's dK and dV rows: lane = key; for a fixed register r the d-tiles of one vector read are VW adjacent
        //      head dimensions, and r = 0 .. 3 continue them: 16 bytes per store
#pragma unroll
      for (int kg = 0; kg < KG; ++kg)
#pragma unroll
        for (int x4 = 0; x4 < (NC + 3) / 4; ++x4)
#pragma unroll
            for (int r0 = 0; r0 < 4; r0 += 4 / VW) {
                // d0 = first head dimension of this store: tiles x = 4 x4 .. (VW per register), registers r0 .. r0 + 4 / VW - 1
                const int d0 = 64 * x4 + VW * (4 * kk + r0);
                const int offk = KG > 1 ? ((int)((kvrow[kg] * p.dk_pitch + d0) * 4) | kvoob[kg]) : kvok[kg] ? (int)((kvrow[kg] * p.dk_pitch + d0) * 4) : OOB;
                const int offv = KG > 1 ? ((int)((kvrow[kg] * p.dv_pitch + d0) * 4) | kvoob[kg]) : kvok[kg] ? (int)((kvrow[kg] * p.dv_pitch + d0) * 4) : OOB;
                float k4[4], v4[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int x = 4 * x4 + e % VW, r = r0 + e / VW;
                    k4[e] = dK[kg][x < NC ? x : 0][r & 3];
                    v4[e] = dV[kg][x < NC ? x : 0][r & 3];
                }
                buf_store4(rsrcDK, offk, k4[0], k4[1], k4[2], k4[3]);
                buf_store4(rsrcDV, offv, v4[0], v4[1], v4[2], v4[3]);
            }
        n_inflight = 2 * NC * KG;
        act = act_n;
        cur_bytes = nxt_bytes;
    }
    // ---- query tiles no key block visited (every position masked): their dQ rows are zero
    if (SPREAD && sk) {
        for (int qt = 0; qt < nqt; ++qt)
            if (!((written >> (qt & 63)) & 1)) __builtin_amdgcn_raw_buffer_store_b32(0u, rsrcDQ, dq_voff, qt * 32 * (int)p.dq_pitch * 4, 0);
    } else if (sk && dq_wave) {
        for (int qt = 0; qt < nqt; ++qt) {
            if ((written >> (qt & 63)) & 1) continue;
            const int dq_tile = qt * 32 * (int)p.dq_pitch * 4 + (NC == 8 ? 0 : dqh * dq_sub);
            const u32x4_t zero = {0u, 0u, 0u, 0u};
#pragma unroll
            for (int t = 0; t < DQT; ++t) __builtin_amdgcn_raw_buffer_store_b128(zero, rsrcDQ, dq_voff, dq_tile + t * dq_sub, 0);
        }
    }
}

long long *g_attn_trace = nullptr;
int g_attn_stagger = 1;
int g_attn_pair = 1;           // forward with a tile summary: two query tiles per block (see mha_fwd_kernel); NPM_TUNE_ATTN_STAGGER's bit 8 clears it
int g_attn_fwd8 = 2;           // NPM_TUNE_ATTN_FWD8: 2 mha_fwd8_kernel (8 waves, 16x16x4 MFMA) for every head size, 1 below head size 128 only,
                               // 0 the 4-wave 32x32x2 mha_fwd_kernel always
int g_attn_bwd16 = 2;          // NPM_TUNE_ATTN_BWD16: 2 (default) mha_bwd8_kernel, except head size 128 with saved scores and no tile
                               // summary, which stays on mha_bwd16_kernel (4.59 against 4.86 ms at C4: fewer vector-ALU instructions per
                               // tile); 3 mha_bwd8_kernel for everything; 1 round 3's choice (mha_bwd16_kernel for head size 128 with
                               // saved scores, the 4-wave kernel otherwise); 0 the 4-wave kernel always
char g_attn_last[96] = "";     // npm_last_attn_kernel: what the most recent npm_mha_core_* call launched

void note_kernel(const char *name, int d, bool mask, bool saved) {
    snprintf(g_attn_last, sizeof g_attn_last, "%s D=%d mask=%d scores=%d", name, d, (int)mask, (int)saved);
}

// delta[b, h, s] = scale * sum_d dO[b, s, h, d] * O[b, s, h, d]: half a wavefront (32 lanes x float4) per (b, s, h) row.
__global__ void __launch_bounds__(256)
mha_delta_kernel(const float *__restrict__ dctx, long dctx_pitch, const float *__restrict__ ctx, long ctx_pitch,
                 float *__restrict__ delta, long batch, long seq, int heads, int dim, float scale) {
    const long rows = batch * seq * heads;
    const int sub = threadIdx.x & 31;
    const long row = ((long)blockIdx.x * blockDim.x + threadIdx.x) >> 5;
    if (row >= rows) return;
    const long bs = row / heads;
    const int h = (int)(row - bs * heads);
    const float *pa = dctx + bs * dctx_pitch + (long)h * dim, *pb = ctx + bs * ctx_pitch + (long)h * dim;
    float acc = 0.f;
    for (int c0 = sub * 4; c0 < dim; c0 += 128) {
        const float4 x = ld4(pa + c0), y = ld4(pb + c0);
        acc += (x.x * y.x + x.y * y.y) + (x.z * y.z + x.w * y.w);
    }
#pragma unroll
    for (int off = 16; off > 0; off >>= 1) acc += __shfl_xor(acc, off, 64);
    if (sub == 0) {
        const long s_ = bs % seq, b_ = bs / seq;
        delta[(b_ * heads + h) * seq + s_] = acc * scale;
    }
}

// Row terms for mha_bwd8_kernel: MINUS delta (delta as above) and lse2 = log2(e) * LSE, both [B, H, sq_pad] with zeros in the padding
// (sq_pad = seq rounded up to whole 32-query tiles), so that the kernel's 16-byte row-term loads never straddle the end.
__global__ void __launch_bounds__(256)
mha_rowterms_kernel(const float *__restrict__ dctx, long dctx_pitch, const float *__restrict__ ctx, long ctx_pitch,
                    const float *__restrict__ lse, float *__restrict__ delta, float *__restrict__ lse2,
                    long batch, long seq, long sq_pad, int heads, int dim, float scale) {
    const long rows = batch * sq_pad * heads;
    const int sub = threadIdx.x & 31;
    const long row = ((long)blockIdx.x * blockDim.x + threadIdx.x) >> 5;
    if (row >= rows) return;
    const long bs = row / heads;
    const int h = (int)(row - bs * heads);
    const long s_ = bs % sq_pad, b_ = bs / sq_pad;
    const long out = (b_ * heads + h) * sq_pad + s_;
    if (s_ >= seq) {
        if (sub == 0) { delta[out] = 0.f; lse2[out] = 0.f; }
        return;
    }
    const long src = b_ * seq + s_;
    const float *pa = dctx + src * dctx_pitch + (long)h * dim, *pb = ctx + src * ctx_pitch + (long)h * dim;
    float acc = 0.f;
    for (int c0 = sub * 4; c0 < dim; c0 += 128) {
        const float4 x = ld4(pa + c0), y = ld4(pb + c0);
        acc += (x.x * y.x + x.y * y.y) + (x.z * y.z + x.w * y.w);
    }
#pragma unroll
    for (int off = 16; off > 0; off >>= 1) acc += __shfl_xor(acc, off, 64);
    if (sub == 0) {
        delta[out] = -acc * scale;                    // NEGATED: the kernel starts its dP accumulators from it (dP - delta for free)
        lse2[out] = lse[(b_ * heads + h) * seq + s_] * LOG2E;
    }
}

// lse2 alone (the caller brought the delta row terms: npm_mha_core.neg_delta): [B, H, sq_pad] = log2(e) * LSE, zeros in the padding
__global__ void __launch_bounds__(256)
mha_lse2_kernel(const float *__restrict__ lse, float *__restrict__ lse2, long planes, long seq, long sq_pad) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= planes * sq_pad) return;
    const long bh = i / sq_pad, s_ = i - bh * sq_pad;
    lse2[i] = s_ < seq ? lse[bh * seq + s_] * LOG2E : 0.f;
}

// Tile summary of a mask: byte (qt, kb) of plane (b, h) has bit w set when some position of queries 32 qt .. 32 qt + 31 x keys
// 128 kb + 16 w .. + 15 is allowed; a second array of the same shape behind it (`total` bytes later) has the bit set when EVERY
// position of the sub-tile that lies inside the tensors is.  One block per (plane, qt, kb): thread = (query row, 16-key group).
__global__ void __launch_bounds__(256)
mha_mask_summary_kernel(const unsigned char *__restrict__ mask, long sb, long sh, long sq, int nh, int seq_q, int seq_kv,
                        int nqt, int nkb, long total, unsigned char *__restrict__ out) {
    __shared__ unsigned bits;
    const long blk = blockIdx.x;
    const int kb = (int)(blk % nkb), qt = (int)((blk / nkb) % nqt);
    const long plane = blk / ((long)nkb * nqt);
    const int hh = (int)(plane % nh);
    const long bb = plane / nh;
    if (threadIdx.x == 0) bits = 0;
    __syncthreads();
    const int row = qt * 32 + (threadIdx.x >> 3), w = threadIdx.x & 7;
    unsigned any = 0, hole = 0, inside = 0;
    if (row < seq_q) {
        const unsigned char *src = mask + bb * sb + hh * sh + (long)row * sq;
        const int k0 = kb * 128 + w * 16;
        for (int i = 0; i < 16; ++i)
            if (k0 + i < seq_kv) {
                any |= src[k0 + i];
                hole |= src[k0 + i] == 0;
                inside = 1;
            }
    }
    // bits 0..7: some position allowed; 8..15: some position excluded; 16..23: the sub-tile has positions inside the tensors
    if (any) atomicOr(&bits, 1u << w);
    if (hole) atomicOr(&bits, 0x100u << w);
    if (inside) atomicOr(&bits, 0x10000u << w);
    __syncthreads();
    if (threadIdx.x == 0) {
        out[blk] = (unsigned char)bits;
        out[total + blk] = (unsigned char)((bits >> 16) & ~(bits >> 8));      // "all": in range and no excluded position
    }
}

// Rows without ANY allowed key.  np.where(mask, scaled, -inf) followed by the softmax (attentions.py:105-110) makes such a row NaN,
// and through P = NaN its dq row and EVERY dk / dv row of its (batch, head) -- which is what the kernels compute when they visit
// every tile.  Skipping must not change results (a skipped tile would leave that row's dq zero and the keys of skipped blocks
// clean), so a plane that has such a row is not skipped at all: every "some position allowed" byte of the plane is set to 0xFF
// (the "every position allowed" bytes stay as computed).  One block per plane, after mha_mask_summary_kernel in stream order.
__global__ void __launch_bounds__(256)
mha_mask_nokey_rows_kernel(const unsigned char *__restrict__ mask, long sb, long sh, long sq, int nh, int seq_q, int seq_kv,
                           int tiles_per_plane, unsigned char *__restrict__ out) {
    __shared__ int empty_row;
    const long plane = blockIdx.x;
    const int hh = (int)(plane % nh);
    const long bb = plane / nh;
    if (threadIdx.x == 0) empty_row = 0;
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const unsigned char *base = mask + bb * sb + hh * sh;
    for (int row = wave; row < seq_q; row += 4) {
        const unsigned char *src = base + (long)row * sq;
        unsigned any = 0;
        for (int k = lane; k < seq_kv; k += 64) any |= src[k];
        if (__builtin_amdgcn_ballot_w64(any != 0) == 0 && lane == 0) empty_row = 1;
    }
    __syncthreads();
    if (empty_row)
        for (int i = threadIdx.x; i < tiles_per_plane; i += 256) out[plane * tiles_per_plane + i] = 0xFF;
}

// The s_memtime stamps of npm_debug_attn_trace live in instances of their own (D = 128, no mask): the product
// instances carry no trace code at all (each stamp is an exec-masked branch in a loop where every instruction counts).
template <int D, bool MASK, bool SAVE>
void launch_fwd_instance(const MhaArgs &a, int grid, hipStream_t s) {
    if (D == 128 && !MASK && a.trace) hipLaunchKernelGGL((mha_fwd_kernel<D, MASK, SAVE, D == 128 && !MASK>), dim3(grid), dim3(256), 0, s, a);
    else hipLaunchKernelGGL((mha_fwd_kernel<D, MASK, SAVE, false>), dim3(grid), dim3(256), 0, s, a);
}

template <int D>
int launch_fwd(const MhaArgs &a, hipStream_t s) {
    const int grid = a.batch * a.heads * (a.q_pair ? (a.q_tiles + 1) / 2 : a.q_tiles);
    const bool mask = a.mask != nullptr, save = a.scores != nullptr;
    if ((g_attn_fwd8 == 2 || (g_attn_fwd8 == 1 && D < 128)) && !a.trace) {
        if (mask && save) hipLaunchKernelGGL((mha_fwd8_kernel<D, true, true>), dim3(grid), dim3(512), 0, s, a);
        else if (mask) hipLaunchKernelGGL((mha_fwd8_kernel<D, true, false>), dim3(grid), dim3(512), 0, s, a);
        else if (save) hipLaunchKernelGGL((mha_fwd8_kernel<D, false, true>), dim3(grid), dim3(512), 0, s, a);
        else hipLaunchKernelGGL((mha_fwd8_kernel<D, false, false>), dim3(grid), dim3(512), 0, s, a);
        note_kernel("mha_fwd8_kernel", D, mask, save);
        NPM_CHECK_LAUNCH();
        return NPM_OK;
    }
    if (mask && save) launch_fwd_instance<D, true, true>(a, grid, s);
    else if (mask) launch_fwd_instance<D, true, false>(a, grid, s);
    else if (save) launch_fwd_instance<D, false, true>(a, grid, s);
    else launch_fwd_instance<D, false, false>(a, grid, s);
    note_kernel("mha_fwd_kernel", D, mask, save);
    NPM_CHECK_LAUNCH();
    return NPM_OK;
}

template <int D, bool MASK, bool SAVED>
void launch_bwd_instance(const MhaArgs &a, int grid, hipStream_t s) {
    if (D == 128 && !MASK && a.trace) hipLaunchKernelGGL((mha_bwd_kernel<D, MASK, SAVED, D == 128 && !MASK>), dim3(grid), dim3(256), 0, s, a);
    else hipLaunchKernelGGL((mha_bwd_kernel<D, MASK, SAVED, false>), dim3(grid), dim3(256), 0, s, a);
}

template <int D>
int launch_bwd8(const MhaArgs &a, hipStream_t s) {
    const int grid = a.batch * a.heads;
    const bool mask = a.mask != nullptr, saved = a.scores != nullptr;
    if (saved) hipLaunchKernelGGL((mha_bwd8_kernel<D, false, true>), dim3(grid), dim3(512), 0, s, a);
    else if (mask) hipLaunchKernelGGL((mha_bwd8_kernel<D, true, false>), dim3(grid), dim3(512), 0, s, a);
    else hipLaunchKernelGGL((mha_bwd8_kernel<D, false, false>), dim3(grid), dim3(512), 0, s, a);
    note_kernel("mha_bwd8_kernel", D, mask && !saved, saved);
    NPM_CHECK_LAUNCH();
    return NPM_OK;
}

template <int D>
int launch_bwd(const MhaArgs &a, hipStream_t s) {
    const int grid = a.batch * a.heads;
    const bool mask = a.mask != nullptr, saved = a.scores != nullptr;
    // Saved scores already CARRY the mask (the forward stored -inf at every masked position, so P = exp2(-inf) = 0
    // there): the backward needs the mask bytes only when it recomputes q.k.  One instance less per head size -- the
    // one whose 16 extra byte loads per tile did not fit the register file at D = 128.
    // (mha_bwd16_kernel and mha_bwd8_kernel are launched by npm_mha_core_bwd itself: they take the padded row terms)
    if (saved) launch_bwd_instance<D, false, true>(a, grid, s);
    else if (mask) launch_bwd_instance<D, true, false>(a, grid, s);
    else launch_bwd_instance<D, false, false>(a, grid, s);
    note_kernel("mha_bwd_kernel", D, mask && !saved, saved);
    NPM_CHECK_LAUNCH();
    return NPM_OK;
}


inline bool al16(const void *ptr) { return ((uintptr_t)ptr & 15) == 0; }

int fill_args(const npm_mha_core *c, bool backward, MhaArgs &a) {
    NPM_ARG(c != nullptr);
    NPM_ARG(c->batch >= 0 && c->heads >= 1 && c->seq_q >= 0 && c->seq_kv >= 1);
    const int d = c->head_dim;
    if (!(d == 16 || d == 32 || d == 64 || d == 128))
        return npm::fail(NPM_E_UNSUPPORTED, "npm_mha_core: head_dim %d is not one of 16, 32, 64, 128 (use the GEMM composition)", d);
    NPM_ARG(c->scale > 0.f);
    NPM_ARG(c->q && c->k && c->v && c->ctx && c->lse);
    auto pitch_ok = [&](int64_t pitch) { return pitch >= (int64_t)c->heads * d && pitch % 4 == 0; };
    NPM_ARG(pitch_ok(c->q_pitch) && pitch_ok(c->k_pitch) && pitch_ok(c->v_pitch) && pitch_ok(c->ctx_pitch));
    NPM_ARG(al16(c->q) && al16(c->k) && al16(c->v) && al16(c->ctx));
    auto span_ok = [&](int64_t rows, int64_t pitch) { return (rows + 128) * pitch * 4 < (1LL << 31); };   // per (batch) extent below 2^31 bytes
    NPM_ARG(span_ok(c->seq_q, c->q_pitch) && span_ok(c->seq_kv, c->k_pitch) && span_ok(c->seq_kv, c->v_pitch) && span_ok(c->seq_q, c->ctx_pitch));
    a = MhaArgs{};
    a.q = c->q; a.k = c->k; a.v = c->v;
    a.q_pitch = c->q_pitch; a.k_pitch = c->k_pitch; a.v_pitch = c->v_pitch;
    a.ctx = c->ctx; a.ctx_pitch = c->ctx_pitch; a.lse = c->lse;
    a.mask = c->mask; a.mask_sb = c->mask_stride_b; a.mask_sh = c->mask_stride_h; a.mask_sq = c->mask_stride_q;
    a.scores = c->scores;
    // tile summary: the kernels keep one bit per tile in a 64-bit scalar, so sequences beyond 2048 just do not skip
    // (and the older backward kernels know nothing of it: with them selected the forward must write every score)
    // (nor do the diagnostic instances of npm_debug_attn_trace: with a trace buffer set the backward falls through to the 4-wave
    //  kernel, which reads every saved score -- the forward must then have written every one)
    if (c->mask && c->tile_summary && c->seq_q <= 2048 && c->seq_kv <= 2048 && g_attn_bwd16 >= 2 && !g_attn_trace) {
        NPM_ARG(c->summary_stride_b >= 0 && c->summary_stride_h >= 0);
        a.skip = c->tile_summary;
        a.skip_sb = c->summary_stride_b;
        a.skip_sh = c->summary_stride_h;
        a.skip_nkb = (c->seq_kv + 127) / 128;
        a.skip_all = c->summary_all_offset > 0 ? c->summary_all_offset : 0;
    }
    NPM_ARG(!c->scores || al16(c->scores));
    NPM_ARG(!c->scores || ((int64_t)c->seq_q + 32) * c->seq_kv * 4 < (1LL << 31));      // one (b, h) score matrix behind one descriptor
    NPM_ARG(!c->mask || (c->mask_stride_q >= 0 && ((int64_t)c->seq_q + 32) * c->mask_stride_q + c->seq_kv < (1LL << 31)));
    a.batch = c->batch; a.heads = c->heads; a.seq_q = c->seq_q; a.seq_kv = c->seq_kv;
    a.scale = c->scale;
    a.q_tiles = (c->seq_q + 127) / 128;
    a.q_pair = a.skip != nullptr && a.q_tiles >= 2 && g_attn_pair;
    a.trace = g_attn_trace;
    a.stagger = g_attn_stagger;
    if (backward) {
        NPM_ARG(c->dctx && c->dq && c->dk && c->dv);
        NPM_ARG(pitch_ok(c->dctx_pitch) && pitch_ok(c->dq_pitch) && pitch_ok(c->dk_pitch) && pitch_ok(c->dv_pitch));
        NPM_ARG(al16(c->dctx) && al16(c->dq) && al16(c->dk) && al16(c->dv));
        NPM_ARG(span_ok(c->seq_q, c->dctx_pitch) && span_ok(c->seq_q, c->dq_pitch) && span_ok(c->seq_kv, c->dk_pitch) && span_ok(c->seq_kv, c->dv_pitch));
        a.dctx = c->dctx; a.dctx_pitch = c->dctx_pitch;
        a.dq = c->dq; a.dk = c->dk; a.dv = c->dv;
        a.dq_pitch = c->dq_pitch; a.dk_pitch = c->dk_pitch; a.dv_pitch = c->dv_pitch;
    }
    return NPM_OK;
}

}  // namespace

extern "C" int npm_debug_attn_trace(long long *buf) { g_attn_trace = buf; return NPM_OK; }
extern "C" int npm_attn_set_stagger(int units) {
    g_attn_pair = !(units >= 0 && (units & 256));
    g_attn_stagger = units < 0 ? 0 : (units & 255);
    return NPM_OK;
}
extern "C" int npm_attn_set_fwd8(int mode) { g_attn_fwd8 = mode < 0 ? 0 : mode > 2 ? 2 : mode; return NPM_OK; }
extern "C" int npm_attn_set_bwd16(int on) { g_attn_bwd16 = on < 0 ? 0 : on > 3 ? 3 : on; return NPM_OK; }

extern "C" const char *npm_last_attn_kernel(void) { return g_attn_last; }

extern "C" int npm_mha_mask_summary(const uint8_t *mask, int64_t stride_b, int64_t stride_h, int64_t stride_q, int32_t planes_b,
                                    int32_t planes_h, int32_t seq_q, int32_t seq_kv, uint8_t *summary) {
    NPM_REQUIRE_INIT();
    NPM_ARG(mask && summary && planes_b >= 1 && planes_h >= 1 && seq_q >= 0 && seq_kv >= 1);
    NPM_ARG(stride_b >= 0 && stride_h >= 0 && stride_q >= 0);
    const int nqt = (seq_q + 31) / 32, nkb = (seq_kv + 127) / 128;
    const long blocks = (long)planes_b * planes_h * nqt * nkb;
    if (blocks == 0) return NPM_OK;
    NPM_ARG(blocks < (1L << 31));
    hipLaunchKernelGGL(mha_mask_summary_kernel, dim3((int)blocks), dim3(256), 0, npm::ctx().stream, mask, (long)stride_b, (long)stride_h,
                       (long)stride_q, planes_h, seq_q, seq_kv, nqt, nkb, blocks, summary);
    NPM_CHECK_LAUNCH();
    hipLaunchKernelGGL(mha_mask_nokey_rows_kernel, dim3(planes_b * planes_h), dim3(256), 0, npm::ctx().stream, mask, (long)stride_b,
                       (long)stride_h, (long)stride_q, planes_h, seq_q, seq_kv, nqt * nkb, summary);
    NPM_CHECK_LAUNCH();
    return NPM_OK;
}

extern "C" int npm_mha_core_supported(int head_dim) { return head_dim == 16 || head_dim == 32 || head_dim == 64 || head_dim == 128; }

extern "C" int npm_mha_core_fwd(const npm_mha_core *c) {
    NPM_REQUIRE_INIT();
    MhaArgs a;
    int rc = fill_args(c, false, a);
    if (rc) return rc;
    if ((long)a.batch * a.heads * a.q_tiles == 0) return NPM_OK;
    NPM_ARG((long)a.batch * a.heads * a.q_tiles < (1L << 31));
    hipStream_t s = npm::ctx().stream;
    npm::note_math(NPM_MATH_F32);
    switch (c->head_dim) {
        case 16: return launch_fwd<16>(a, s);
        case 32: return launch_fwd<32>(a, s);
        case 64: return launch_fwd<64>(a, s);
        default: return launch_fwd<128>(a, s);
    }
}

extern "C" int npm_mha_core_bwd(const npm_mha_core *c) {
    NPM_REQUIRE_INIT();
    MhaArgs a;
    int rc = fill_args(c, true, a);
    if (rc) return rc;
    if ((long)a.batch * a.heads == 0) return NPM_OK;
    hipStream_t s = npm::ctx().stream;
    if (a.seq_q == 0) {                                  // no queries: nothing flows back to the keys and values
        const size_t width = sizeof(float) * (size_t)a.heads * c->head_dim, rows = (size_t)a.batch * a.seq_kv;
        NPM_HIP(hipMemset2DAsync(a.dk, sizeof(float) * a.dk_pitch, 0, width, rows, s));
        NPM_HIP(hipMemset2DAsync(a.dv, sizeof(float) * a.dv_pitch, 0, width, rows, s));
        return NPM_OK;
    }
    npm::Scratch ws;                                  // stream-ordered pool: safe to release when this call returns
    const bool wide128 = c->head_dim == 128 && a.scores != nullptr && a.skip == nullptr;      // the case mha_bwd16_kernel was built for
    const bool use8 = (g_attn_bwd16 == 3 || (g_attn_bwd16 == 2 && !wide128)) && !a.trace;
    const bool use16 = !use8 && wide128 && (g_attn_bwd16 == 1 || g_attn_bwd16 == 2) && (!a.trace || g_attn_bwd16 == 1);   // traced: knob 1
    if (use8 || use16) {
        // mha_bwd8_kernel: row terms padded to whole query tiles ([B, H, sq_pad] each: delta, then log2(e) LSE)
        a.sq_pad = (a.seq_q + 31) / 32 * 32;
        const long padded = (long)a.batch * a.heads * a.sq_pad;
        rc = ws.alloc(sizeof(float) * 2 * (size_t)padded);
        if (rc) return rc;
        a.lse2 = (float *)ws.ptr + padded;
        if (c->neg_delta != nullptr && a.seq_q % 4 == 0) {
            // the caller computed MINUS scale * (dctx_i . ctx_i) where dctx was produced (NPM_EPI_ROWDOT): no pass over dctx and ctx.
            // The kernels fetch four row terms per 16-byte load: rows of seq_q % 4 != 0 floats are not aligned for that, and
            // such a call recomputes the terms into the padded scratch below like a call without neg_delta.
            NPM_ARG(c->neg_delta_stride_b >= 0 && c->neg_delta_stride_h >= 0 && al16(c->neg_delta) &&
                    c->neg_delta_stride_b % 4 == 0 && c->neg_delta_stride_h % 4 == 0);
            a.delta = const_cast<float *>(c->neg_delta);
            a.delta_sb = c->neg_delta_stride_b; a.delta_sh = c->neg_delta_stride_h; a.delta_len = a.seq_q;
            NPM_ARG((padded + 255) / 256 < (1L << 31));
            hipLaunchKernelGGL(mha_lse2_kernel, dim3((int)((padded + 255) / 256)), dim3(256), 0, s, (const float *)a.lse, a.lse2,
                               (long)a.batch * a.heads, (long)a.seq_q, (long)a.sq_pad);
        } else {
            a.delta = (float *)ws.ptr;
            a.delta_sb = (long)a.heads * a.sq_pad; a.delta_sh = a.sq_pad; a.delta_len = a.sq_pad;
            NPM_ARG((padded * 32 + 255) / 256 < (1L << 31));
            hipLaunchKernelGGL(mha_rowterms_kernel, dim3((int)((padded * 32 + 255) / 256)), dim3(256), 0, s, a.dctx, a.dctx_pitch,
                               (const float *)a.ctx, a.ctx_pitch, (const float *)a.lse, a.delta, a.lse2, (long)a.batch, (long)a.seq_q,
                               (long)a.sq_pad, a.heads, c->head_dim, a.scale);
        }
        NPM_CHECK_LAUNCH();
        npm::note_math(NPM_MATH_F32);
        if (use16) {
            if (a.trace) hipLaunchKernelGGL(mha_bwd16_kernel<true>, dim3(a.batch * a.heads), dim3(512), 0, s, a);
            else hipLaunchKernelGGL(mha_bwd16_kernel<false>, dim3(a.batch * a.heads), dim3(512), 0, s, a);
            note_kernel("mha_bwd16_kernel", 128, false, true);
            NPM_CHECK_LAUNCH();
            return NPM_OK;
        }
        switch (c->head_dim) {
            case 16: return launch_bwd8<16>(a, s);
            case 32: return launch_bwd8<32>(a, s);
            case 64: return launch_bwd8<64>(a, s);
            default: return launch_bwd8<128>(a, s);
        }
    }
    const long rows = (long)a.batch * a.seq_q * a.heads;
    rc = ws.alloc(sizeof(float) * (size_t)rows);
    if (rc) return rc;
    a.delta = (float *)ws.ptr;
    NPM_ARG((rows * 32 + 255) / 256 < (1L << 31));
    hipLaunchKernelGGL(mha_delta_kernel, dim3((int)((rows * 32 + 255) / 256)), dim3(256), 0, s, a.dctx, a.dctx_pitch,
                       (const float *)a.ctx, a.ctx_pitch, a.delta, (long)a.batch, (long)a.seq_q, a.heads, c->head_dim, a.scale);
    NPM_CHECK_LAUNCH();
    npm::note_math(NPM_MATH_F32);
    switch (c->head_dim) {
        case 16: return launch_bwd<16>(a, s);
        case 32: return launch_bwd<32>(a, s);
        case 64: return launch_bwd<64>(a, s);
        default: return launch_bwd<128>(a, s);
    }
}
