// Marker-controlled watershed (W1): skimage.segmentation.watershed(image,
// markers, mask=mask), connectivity 1, no compactness, no watershed line
// (refine_boundaries.py:73).
//
// The reference is a sequential priority flood keyed by (value, age).  Its pop
// order is ordered first by the minimax ("pass") level
//     L(seed) = value(seed),  L(q) = max(value(q), min over 4-neighbours L(n)),
// because every comparison that decides "a pops before b" for L(a) < L(b) is a
// strict value comparison.  Hence the pixel that labels q (its first-popped
// neighbour) has L = min over q's neighbours, and
//   * parallel path: (1) L by tile-iterated relaxation (rounds alternate
//     between two half-tile-shifted tilings), (2) labels by a seed-first
//     union-find over the links "pixel -- each neighbour with L == Lmin", (3) a
//     proof condition: for every non-seed reachable pixel ALL neighbours with
//     L == Lmin carry the pixel's label (equivalently: no union-find component
//     holds two marker ids).  If it holds for the whole frame the result equals
//     the sequential flood's for ANY tie-breaking (induction over the pop
//     order), so the frame is bit-exact without emulating the heap;
//   * frames that fail get a second-level order (L, K2) -- see below -- and the
//     same union-find on the refined keys;
//   * frames that fail the check (equal-valued bottlenecks or seeds between two
//     basins -- ubiquitous in quantised probability maps) are recomputed by an
//     exact emulation of the reference's binary heap, one wave per frame.
#include <algorithm>
#include <atomic>
#include <mutex>
#include <type_traits>

#include "common.h"

namespace pcseg {

constexpr int WS_T = 64;           // tile edge
constexpr int WS_S = WS_T + 2;     // with halo
constexpr int WS_P = 67;           // LDS row pitch in elements (odd: row-per-lane sweeps are bank-conflict free)
constexpr int WS_N = WS_S * WS_P;  // LDS elements per tile array
constexpr unsigned WS_INF = 0xFFFFFFFFu;
constexpr int WS_CNT0 = 32, WS_CNT_STRIDE = 32, WS_CHANGED_INTS = WS_CNT0 + 16 * WS_CNT_STRIDE;  // layout of the `changed` block

// order-preserving key of a float32 (the reference compares float64(image)); -0.0 == +0.0
__device__ __forceinline__ unsigned ws_key(float f)
{
    if (f == 0.0f) f = 0.0f;
    unsigned u = __float_as_uint(f);
    u = (u & 0x80000000u) ? ~u : (u | 0x80000000u);
    return u == WS_INF ? WS_INF - 1 : u;
}

__global__ void __launch_bounds__(256) ws_init_kernel(const float *__restrict__ img, int64_t frame_stride,
                                                       const int *__restrict__ markers, const uint8_t *__restrict__ mask,
                                                       unsigned *__restrict__ val, unsigned *__restrict__ L,
                                                       int *__restrict__ out, int64_t n, int64_t total)
{
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    bool m = mask[i] != 0;
    unsigned v = m ? ws_key(img[(i / n) * frame_stride + (i % n)]) : WS_INF;
    int mk = m ? markers[i] : 0;
    val[i] = v;
    L[i] = mk != 0 ? v : WS_INF;
    out[i] = mk;
}

template <typename T>
__device__ __forceinline__ void ws_load_tile(T *s, const T *__restrict__ g, int r0, int c0, int H, int W, T fill)
{
    for (int i = threadIdx.x; i < WS_S * WS_S; i += 256) {
        int lr = i / WS_S, lc = i % WS_S;
        int r = r0 + lr - 1, c = c0 + lc - 1;
        s[lr * WS_P + lc] = (r >= 0 && r < H && c >= 0 && c < W) ? g[rowoff(r, W) + c] : fill;
    }
}

// The tile fixed points below are run as DIRECTIONAL SWEEPS: wave 0 sweeps the 64 rows left->right (one row per
// lane), wave 1 right->left, wave 2 the 64 columns top->bottom, wave 3 bottom->top, all at once.  Every update is
// monotone (min / 0->label), so concurrent sweeps may read each other's half-finished values: a stale read only
// costs another outer iteration.  One sweep carries information across the whole tile, so the number of outer
// iterations is the number of direction changes of the dependency paths, not their length.
struct SweepLine {
    int start, step;  // LDS index of the halo element in front of the line, and the index step along the line
};
__device__ __forceinline__ SweepLine ws_line()
{
    const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
    SweepLine s;
    if (w == 0) { s.start = (lane + 1) * WS_P; s.step = 1; }
    else if (w == 1) { s.start = (lane + 1) * WS_P + WS_S - 1; s.step = -1; }
    else if (w == 2) { s.start = lane + 1; s.step = WS_P; }
    else { s.start = (WS_S - 1) * WS_P + lane + 1; s.step = -WS_P; }
    return s;
}

template <typename T>
__device__ __forceinline__ void ws_store_tile(const T *s, T *__restrict__ g, int r0, int c0, int H, int W)
{
    for (int i = threadIdx.x; i < WS_T * WS_T; i += blockDim.x) {
        int lr = i / WS_T, lc = i % WS_T;
        int r = r0 + lr, c = c0 + lc;
        if (r < H && c < W) g[rowoff(r, W) + c] = s[(lr + 1) * WS_P + lc + 1];
    }
}

// After a tile converged: wave e compares edge e of the tile (0 top, 1 bottom, 2 left, 3 right) with what is still in
// global memory and marks only the neighbour that shares a CHANGED edge.  Must run BEFORE the tile is stored.
template <typename T>
__device__ __forceinline__ void ws_mark_changed_edges(const T *s, const T *__restrict__ g, uint8_t *dirty_out,
                                                      int b, int tx, int ty, int tilesX, int tilesY, int r0, int c0, int H, int W)
{
    if (threadIdx.x >= 256) return;  // (wave e = edge e: the first four waves of a wider block)
    const int e = threadIdx.x >> 6, j = threadIdx.x & 63;
    const int lr = e == 0 ? 1 : (e == 1 ? WS_T : j + 1);
    const int lc = e == 2 ? 1 : (e == 3 ? WS_T : j + 1);
    const int r = r0 + lr - 1, c = c0 + lc - 1;
    bool ch = false;
    if (r < H && c < W) ch = s[lr * WS_P + lc] != g[rowoff(r, W) + c];
    if (__any(ch) && j == 0) {
        uint8_t *d = dirty_out + (int64_t)b * tilesX * tilesY;
        if (e == 0 && ty > 0) d[(ty - 1) * tilesX + tx] = 1;
        else if (e == 1 && ty + 1 < tilesY) d[(ty + 1) * tilesX + tx] = 1;
        else if (e == 2 && tx > 0) d[ty * tilesX + tx - 1] = 1;
        else if (e == 3 && tx + 1 < tilesX) d[ty * tilesX + tx + 1] = 1;
    }
}

// (1) minimax relaxation, tile-local fixed point in LDS.  L and the pixel value share one 64-bit LDS word (x = L,
// y = value): a sweep step is ONE ds_read_b64 instead of two ds_read_b32 -- the kernel is bound by LDS-array cycles, and
// the b64 form moves twice the bytes per cycle (odd pitch: conflict-free for row and column sweeps alike).
//
// Geometry of a relaxation tile of edge T (64 or 128; PCSEG_WS_RELAX_TILE picks, 64 is the default).  T = 128 fills the
// 160 KB LDS of a CDNA4 CU -- the tile with its halo is 130 x 131 x 8 B = 133 KB, ONE workgroup of 16 waves per CU (as
// many waves as four 64-tiles bring), the four 64 x 64 quadrants exchange their rims through LDS inside one iteration
// -- and MEASURED SLOWER on the benchmark batch (3.30 ms of relaxation per step against 2.16 ms, rounds 1133 / 1121 /
// 533 / 240 us against 742 / 606 / 327 / 175 us): with one workgroup per CU nothing runs under a tile's load, store and
// 16-wave barriers, the shifted tiling has 81 tiles per 1024^2 frame instead of 64 (+27 %) where 64-tiles have 289
// instead of 256 (+13 %), and the number of rounds hardly drops because levels travel along winding paths, not tile
// diameters.
#ifndef PCSEG_WS_SWEEP_MASKS
// change tracking of the quadrant sweep (A/B builds): 0 = per-lane xor / or, every cell of a changed batch goes through its
// LDS atomic (round 2); 1 = scalar lane masks from one v_cmp per step; 2 = 1 + only the lanes that lowered a cell issue
// the atomic.  Same box, relaxation us per launch / overlapped ms per step (profiles/r03/ab_run.sh): 0: 119.7 / 5.85-5.93,
// 1: 116.7 / 5.83-5.84, 2: 124.1 / 6.09-6.31 -- an exec-masked ds_min costs the LDS what a full one does, and the eight
// scalar branches per batch come on top: 1 stays
#define PCSEG_WS_SWEEP_MASKS 1
#endif
#ifndef PCSEG_WS_FSM
#define PCSEG_WS_FSM 1  // 1: quadrant (raster-order wavefront) sweeps, 0: line sweeps -- see ws_quadrant_sweep
#endif
template <int T>
struct RelaxGeom {
    static_assert(T == 64 || T == 128, "relaxation tiles are 64 or 128 pixels wide");
    static constexpr int S = T + 2;        // with halo
    // LDS row pitch in elements.  Line sweeps (one row or column per lane) want it odd; the wavefront sweeps of the
    // quadrant scheme walk lane l along row l at column (step - l), i.e. lanes are P - 1 or P + 1 elements apart: even P
    static constexpr int P = (PCSEG_WS_FSM && T == 64) ? T + 4 : T + 3;
    static constexpr int PAD = 8;          // elements in front of and behind the tile (wavefront lanes read past their row's ends)
    static constexpr int N = S * P + 2 * PAD;  // LDS elements
    static constexpr int G = T / 64;       // 64-line groups per direction, and 64-cell segments per line
    static constexpr int THREADS = 4 * G * G * 64;  // one wave per (direction, line group, segment)
    static constexpr int HE = T / 2;       // cells of a half edge
    static constexpr size_t LDS_BYTES = sizeof(uint2) * N;
};

// one directional sweep of a 64-cell segment of a line: L(i) = min(L(i), max(value(i), L(previous cell))).  Eight cells
// are fetched before any of them is updated: the reads of a batch carry no dependency on the batch's writes (a cell is
// written only when it is processed), and what another wave writes meanwhile is picked up an iteration later, which
// the monotone update tolerates.  The batch loop is NOT unrolled: four direction-specific bodies have to stay resident
// in the I-cache.
#ifndef WS_BATCH
#define WS_BATCH 8
#endif
#ifndef PCSEG_WS_ASM_READ
#define PCSEG_WS_ASM_READ 1
#endif
template <int STEP>
__device__ __forceinline__ bool ws_sweep(uint2 *sLV, int start)
{
    unsigned *sLw = reinterpret_cast<unsigned *>(sLV);  // the L half of element i is word 2 * i
    unsigned diff = 0;  // != 0 once a cell of this line was lowered
    unsigned prev = sLV[start].x;
    int base = start + STEP;
#pragma unroll 1
    for (int k0 = 0; k0 < 64; k0 += WS_BATCH, base += WS_BATCH * STEP) {
        uint2 lv[WS_BATCH];
#if PCSEG_WS_ASM_READ
        // The eight reads are written as ds_read_b64 by hand: the compiler pairs neighbouring reads into ds_read2_b64,
        // which the LDS serves at HALF the bytes per clock of ds_read_b64 (MI355X_MICROARCH.md, LDS table: 8 cycles
        // for 2 x 512 B against 2 cycles per 512 B), and this kernel is bound by LDS-array cycles.  Offsets are
        // unsigned: a batch is addressed from its lowest cell.
        {
            static_assert(WS_BATCH == 8 || WS_BATCH == 16, "the hand-written batch read is eight or sixteen wide");
            typedef unsigned u2v __attribute__((ext_vector_type(2)));
            constexpr int A = STEP < 0 ? -STEP : STEP;
            const int low = STEP < 0 ? base + (WS_BATCH - 1) * STEP : base;
            const unsigned addr = (unsigned)(uintptr_t)(sLV + low);
            u2v t[WS_BATCH];
#define PCSEG_DS_READ(j) asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(t[j]) : "v"(addr), "n"(8 * A * (STEP < 0 ? WS_BATCH - 1 - (j) : (j))))
            PCSEG_DS_READ(0); PCSEG_DS_READ(1); PCSEG_DS_READ(2); PCSEG_DS_READ(3);
            PCSEG_DS_READ(4); PCSEG_DS_READ(5); PCSEG_DS_READ(6); PCSEG_DS_READ(7);
#if WS_BATCH == 16
            PCSEG_DS_READ(8); PCSEG_DS_READ(9); PCSEG_DS_READ(10); PCSEG_DS_READ(11);
            PCSEG_DS_READ(12); PCSEG_DS_READ(13); PCSEG_DS_READ(14); PCSEG_DS_READ(15);
            asm volatile("s_waitcnt lgkmcnt(8)" : "+v"(t[0]), "+v"(t[1]), "+v"(t[2]), "+v"(t[3]), "+v"(t[4]), "+v"(t[5]), "+v"(t[6]), "+v"(t[7]));
            asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(t[8]), "+v"(t[9]), "+v"(t[10]), "+v"(t[11]), "+v"(t[12]), "+v"(t[13]), "+v"(t[14]), "+v"(t[15]));
#else
            asm volatile("s_waitcnt lgkmcnt(4)" : "+v"(t[0]), "+v"(t[1]), "+v"(t[2]), "+v"(t[3]));
            asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(t[4]), "+v"(t[5]), "+v"(t[6]), "+v"(t[7]));
#endif
#undef PCSEG_DS_READ
#pragma unroll
            for (int j = 0; j < WS_BATCH; ++j) lv[j] = make_uint2(t[j].x, t[j].y);
        }
#else
#pragma unroll
        for (int j = 0; j < WS_BATCH; ++j) lv[j] = sLV[base + j * STEP];
#endif
        // the running values come from registers alone; the LDS updates of a batch are skipped as a whole when no lane
        // lowered anything in it (late iterations, converged neighbourhoods): one ballot instead of eight atomics
        unsigned nw[WS_BATCH];
        unsigned batch_diff = 0;
#pragma unroll
        for (int j = 0; j < WS_BATCH; ++j) {
            // value <= L holds for every cell (seeds start at their value, everything else at +inf, and a level never
            // drops below its cell's value), so min(L, max(value, prev)) is the MEDIAN of the three: one v_med3_u32 on
            // the serial chain instead of v_max followed by v_min
            const unsigned cur = lv[j].x, v = lv[j].y;
            nw[j] = min(max(v, prev), max(min(v, prev), cur));
            batch_diff |= cur ^ nw[j];
            prev = nw[j];
        }
        if (__any(batch_diff != 0)) {
            // unconditional LDS atomic min per cell: nothing under an exec mask (compare + store to the cell or to a pad
            // word measured 13 % slower, a compare + masked store 10 %), and still monotone when another wave lowered
            // the cell since the batch was read
#pragma unroll
            for (int j = 0; j < WS_BATCH; ++j) atomicMin(&sLw[2 * (base + j * STEP)], nw[j]);
            diff |= batch_diff;
        }
    }
    return diff != 0;
}

// QUADRANT SWEEPS (fast-sweeping order; PCSEG_WS_FSM, the default).  A line sweep carries a level along one axis only,
// so a minimax path that turns costs an iteration per turn -- and the paths of a noisy probability map turn every few
// pixels: isolated 64 x 64 tiles of the benchmark frames need 14 iterations of the four line sweeps on average (35 at
// worst).  A raster-order Gauss-Seidel sweep
//     L(r, c) = min(L(r, c), max(value(r, c), min(L(r - dr, c), L(r, c - dc))))   rows in direction dr, columns in dc
// carries a level along ANY path that is monotone in both axes, so what counts is the number of QUADRANT changes of a
// path: 3.6 iterations on average, 7 at worst; over a whole frame and all rounds 2951 tile-iterations instead of 7751
// (numpy simulation of the round structure, same fixed point).  One wave runs one quadrant as a wavefront: lane l owns
// row l (from the top for dr > 0, from the bottom otherwise), the row above is exactly one step ahead on the same column
// and hands its result down through a DPP wave shift -- no LDS read for either neighbour; the left neighbour is the
// lane's own previous result.
// History: the plain wavefront (lane l at column step - l, 127 steps, half of the lane-steps idle, activity selects
// on every step) cost 3.9 x a line iteration and lost (2.76 ms of relaxation per step against 2.0).  The CYCLIC
// wavefront below keeps all lanes busy for exactly 64 steps: 2.4 x and 1.62 ms per step with per-lane compares for
// the wrap, 2.0 x and 1.34 ms with the selects driven by scalar lane masks and the atomics reusing the read addresses
// (rounds 377 / 365 / 240 / 144 us against 742 / 606 / 327 / 175).  VALU-bound: six operations per step.
template <int DR, int DC, int P>
__device__ __forceinline__ bool ws_quadrant_sweep(uint2 *sLV, int lane)
{
    // CYCLIC wavefront: lane l owns row l and is at cyclic position u = (step - l) mod 64 of it (column u for dc > 0,
    // 63 - u otherwise), so all 64 lanes work in all 64 steps.  The row above is still exactly one step ahead on the
    // same column; a row is swept from wherever its lane starts to its end and then from its beginning -- a valid
    // Gauss-Seidel order (2.5 x fewer tile-iterations than line sweeps in the whole-frame simulation, against 2.6 x for
    // the pure raster order).  At u == 0 the running value restarts from the halo cell in front of the row.
    unsigned *sLw = reinterpret_cast<unsigned *>(sLV);
    typedef unsigned u2v __attribute__((ext_vector_type(2)));
    // "did any lane lower a cell" is kept as a SCALAR lane mask: one v_cmp per step (its result lands in an SGPR pair, the
    // OR into the running mask is a scalar instruction) instead of a per-lane xor + or -- five VALU operations per cell-step
    // instead of six in a VALU-bound loop
    unsigned long long diff = 0;
    {   // the first row takes the halo row's levels before the sweep (lane = column): lane 0 then needs no `up`
        constexpr int fr = DR > 0 ? 1 : WS_T, hr = DR > 0 ? 0 : WS_T + 1;
        const uint2 c = sLV[fr * P + 1 + lane];
        const unsigned h = sLV[hr * P + 1 + lane].x;
        const unsigned nw = min(c.x, max(c.y, h));
        atomicMin(&sLw[2 * (fr * P + 1 + lane)], nw);
        diff |= __ballot(nw != c.x);
    }
    const int lr = DR > 0 ? lane + 1 : WS_T - lane;  // the lane's row (LDS coordinates 1 .. 64)
    const uint2 *row = sLV + lr * P;
    const unsigned halo = row[DC > 0 ? 0 : WS_T + 1].x;
    const int u0 = (0 - lane) & 63;                   // cyclic position at step 0
    // what the first step finds behind it: the row's cell at position u0 - 1 as it is now
    unsigned left = row[DC > 0 ? u0 : WS_T + 1 - u0].x;  // (u0 == 0: replaced by the halo cell at the first step anyway)
    // what the next lane finds above its first cell: this row's cell at THAT lane's first position
    const int un = (0 - (lane + 1)) & 63;
    unsigned prev = row[DC > 0 ? 1 + un : WS_T - un].x;
    // byte address of the row's cell at cyclic position 0, and the step per position
    const unsigned a0 = (unsigned)(uintptr_t)(row + (DC > 0 ? 1 : WS_T));
#pragma unroll 1
    for (int j0 = 0; j0 < WS_T; j0 += 8) {
        const int ub = (u0 + j0) & 63;
        // Step k reads position ub + k, 64 positions (512 bytes) back once the lane has wrapped: one of two bases, the
        // step itself as the instruction's offset (counted from the batch's lowest address for dc < 0).  WHICH lanes have
        // wrapped is wave-uniform knowledge: lane l is at position 0 at step l, so at step k of this batch the wrapped
        // lanes are j0 + 1 .. j0 + k and the lane at position 0 is j0 + k -- the selects take their condition from a
        // scalar mask (one VALU operation each, no per-lane compare).
        const unsigned base_a = DC > 0 ? a0 + 8u * (unsigned)ub : a0 - 8u * (unsigned)ub - 56u;
        const unsigned base_b = DC > 0 ? base_a - 512u : base_a + 512u;
        unsigned ba[8];
        u2v t[8];
#define PCSEG_DS_READ(k)                                                                                                   \
        {                                                                                                                      \
            const unsigned long long wrapped = (((1ull << (k)) - 1ull) << 1) << j0;  /* lanes j0 + 1 .. j0 + k */              \
            asm("v_cndmask_b32_e64 %0, %1, %2, %3" : "=v"(ba[k]) : "v"(base_a), "v"(base_b), "s"(wrapped));                     \
            asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(t[k]) : "v"(ba[k]), "n"(DC > 0 ? 8 * (k) : 8 * (7 - (k))));     \
        }
        PCSEG_DS_READ(0) PCSEG_DS_READ(1) PCSEG_DS_READ(2) PCSEG_DS_READ(3)
        PCSEG_DS_READ(4) PCSEG_DS_READ(5) PCSEG_DS_READ(6) PCSEG_DS_READ(7)
#undef PCSEG_DS_READ
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(t[0]), "+v"(t[1]), "+v"(t[2]), "+v"(t[3]), "+v"(t[4]), "+v"(t[5]), "+v"(t[6]), "+v"(t[7]));
        unsigned wr[8];
        bool lowered[8];  // per lane "this step lowered its cell": lives as a lane mask in an SGPR pair (the v_cmp's result)
        unsigned long long batch_diff = 0;
#if PCSEG_WS_SWEEP_MASKS == 0
        unsigned lane_diff = 0;
#endif
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            // lane l takes lane l - 1's result of the previous step; lane 0 has no row above (wave_shr:1, `old` = +inf)
            const unsigned up = (unsigned)__builtin_amdgcn_update_dpp((int)WS_INF, (int)prev, 0x138, 0xF, 0xF, false);
            const unsigned long long at0 = (1ull << k) << j0;  // the lane that is at position 0 now restarts from its halo cell
            unsigned lf;
            asm("v_cndmask_b32_e64 %0, %1, %2, %3" : "=v"(lf) : "v"(left), "v"(halo), "s"(at0));
            const unsigned cur = t[k].x, v = t[k].y;
            const unsigned m = min(up, lf);
            const unsigned cand = min(max(v, m), max(min(v, m), cur));  // median(value, m, cur) = min(cur, max(value, m))
            wr[k] = cand;
#if PCSEG_WS_SWEEP_MASKS == 0
            lane_diff |= cur ^ cand;
            lowered[k] = true;
#else
            lowered[k] = cur != cand;
            batch_diff |= __ballot(lowered[k]);
#endif
            left = cand;
            prev = cand;
        }
#if PCSEG_WS_SWEEP_MASKS == 0
        batch_diff = __ballot(lane_diff != 0);
#endif
        if (batch_diff != 0) {
            // unconditional LDS atomic min per cell of a changed batch (variant 2 predicates it per lane: measured slower);
            // still monotone when another wave lowered the cell since the batch was read.  (The level is the first word of
            // the cell: the read's address and offset serve the atomic as they are.)
#define PCSEG_DS_MIN(k)                                                                                                             \
            if (PCSEG_WS_SWEEP_MASKS < 2 || lowered[k])                                                                                 \
                asm volatile("ds_min_u32 %0, %1 offset:%2" : : "v"(ba[k]), "v"(wr[k]), "n"(DC > 0 ? 8 * (k) : 8 * (7 - (k))) : "memory");
            PCSEG_DS_MIN(0) PCSEG_DS_MIN(1) PCSEG_DS_MIN(2) PCSEG_DS_MIN(3)
            PCSEG_DS_MIN(4) PCSEG_DS_MIN(5) PCSEG_DS_MIN(6) PCSEG_DS_MIN(7)
#undef PCSEG_DS_MIN
            diff |= batch_diff;
        }
    }
    return diff != 0;
}

// The FIRST round also does the set-up (what ws_init_kernel does for the exact-only mode): value keys, seed levels and
// the seed labels are computed while the tile is loaded, so the three input arrays are read once and val / L are not
// written and read back in between.
struct WsInputs {
    const float *img;
    int64_t frame_stride;
    const int *markers;
    const uint8_t *mask;
    int *out;
    bool vec;  // W % 4 == 0 and every array 16-byte aligned per frame: the tile load / store use 16-byte accesses
};

//
// Rounds ALTERNATE between two tilings: tile (tx, ty) starts at (ty * T - off, tx * T - off) with off = 0 in even
// rounds and T / 2 in odd ones, so the tile borders of one round are tile centres of the next and a path that winds
// across a border is resolved inside one tile a round later instead of costing a round per crossing.  Which tiles the
// next round has to visit follows from the half edges that changed: after a tile reached its fixed point its cells are
// consistent with each other, a violated pixel can only sit on the tile's rim next to a neighbour outside, and rim
// cell plus outside neighbour both lie in the ONE tile of the other tiling that is centred on the nearest tile
// corner.  So each tile marks, per corner quadrant, that corner's tile if one of the quadrant's two outer half
// edges changed.  (grid = this round's tiling, dirty_in in its layout; dirty_out in the other tiling's layout.)
struct WsTiling {
    int off;     // 0 or T / 2
    int nx, ny;  // tiles per frame in x and y
};

// One tile of one round (block-uniform control flow: every return is taken by all threads of the block).
template <int T>
__device__ __forceinline__ void ws_relax_tile(uint2 *sLV, const WsInputs &in, const bool FIRST, unsigned *__restrict__ val,
                                              unsigned *__restrict__ L, uint8_t *__restrict__ dirty_in,
                                              uint8_t *__restrict__ dirty_out, int *__restrict__ any_changed, int H, int W,
                                              const WsTiling &cur, const WsTiling &nxt, int max_iter, int tx, int ty, int b,
                                              int *__restrict__ list_out = nullptr, int *__restrict__ count_out = nullptr)
{
    using G = RelaxGeom<T>;
    constexpr int S = G::S, P = G::P, NT = G::THREADS, QW = T / 4;  // QW: 16-byte quads per tile row
    const int tid = threadIdx.x;
    if (!FIRST) {
        // a visited tile takes its mark down itself: the buffer is all zero again when it becomes the output of the
        // round after next, and no memset has to sit between two rounds
        uint8_t *mark = dirty_in + ((int64_t)b * cur.ny + ty) * cur.nx + tx;
        if (!*mark) return;
        __syncthreads();
        if (tid == 0) *mark = 0;
    }
    // 64 x 64 units actually processed (measurement: bench.py roofline), spread over 16 cache lines: one counter would
    // make every block of the launch queue on the same line
    if (tid == 0) atomicAdd(any_changed + WS_CNT0 + WS_CNT_STRIDE * ((tx + 5 * ty + 3 * b) & 15), G::G * G::G);
    const int r0 = ty * T - cur.off, c0 = tx * T - cur.off;
    const int64_t fbase = (int64_t)b * H * W;
    // (L, value) of a pixel before any relaxation
    auto initial = [&](int r, int c) -> uint2 {
        const int64_t g = fbase + rowoff(r, W) + c;
        if (!in.mask[g]) return make_uint2(WS_INF, WS_INF);
        const unsigned v = ws_key(in.img[(int64_t)b * in.frame_stride + rowoff(r, W) + c]);
        return make_uint2(in.markers[g] != 0 ? v : WS_INF, v);
    };
    if (in.vec) {
        // The tile load is ONE batch of loads per thread, not a loop of dependent round trips: every access goes to a
        // clamped (always valid) address without a branch, so that the unrolled loop is a single basic block whose
        // loads the compiler issues back to back, and the out-of-frame cells are fixed up by a select afterwards.
        // Interior columns as 16-byte quads (S rows x T / 4 quads: 5 trips), the two halo columns as one scalar trip.
        // (A loop with an `if (inside)` around each load measured one full memory latency per trip: 17 trips, i.e.
        // most of a revisited tile's time.)
        constexpr int QUADS = S * QW, TRIPS = (QUADS + NT - 1) / NT;
        const float *img_f = in.img + (int64_t)b * in.frame_stride;
        const bool halo_thread = tid < 2 * S;
        const int h_lr = tid >> 1, h_lc = (tid & 1) ? S - 1 : 0;
        const int h_r = r0 + h_lr - 1, h_c = c0 + h_lc - 1;
        const bool h_in = halo_thread && h_r >= 0 && h_r < H && h_c >= 0 && h_c < W;
        const int64_t h_p = rowoff(min(max(h_r, 0), H - 1), W) + min(max(h_c, 0), W - 1);
        if (FIRST) {
            float4 f4[TRIPS];
            int4 m4[TRIPS];
            unsigned k4[TRIPS];
#pragma unroll
            for (int t = 0; t < TRIPS; ++t) {
                const int idx = min(tid + NT * t, QUADS - 1);
                const int lr = idx / QW, q = idx % QW;
                const int64_t p = rowoff(min(max(r0 + lr - 1, 0), H - 1), W) + min(max(c0 + 4 * q, 0), W - 4);
                f4[t] = *reinterpret_cast<const float4 *>(img_f + p);
                m4[t] = *reinterpret_cast<const int4 *>(in.markers + fbase + p);
                k4[t] = *reinterpret_cast<const unsigned *>(in.mask + fbase + p);
            }
            const float hf = img_f[h_p];
            const int hm = in.markers[fbase + h_p];
            const uint8_t hk = in.mask[fbase + h_p];
#pragma unroll
            for (int t = 0; t < TRIPS; ++t) {
                const int idx = tid + NT * t;
                if (idx < QUADS) {
                    const int lr = idx / QW, q = idx % QW;
                    const int r = r0 + lr - 1, c = c0 + 4 * q;
                    const bool inside = r >= 0 && r < H && c >= 0 && c < W;
                    const float fv[4] = {f4[t].x, f4[t].y, f4[t].z, f4[t].w};
                    const int mv[4] = {m4[t].x, m4[t].y, m4[t].z, m4[t].w};
                    unsigned key[4];
                    int lab[4];
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const bool msk = inside && ((k4[t] >> (8 * j)) & 255u) != 0;
                        key[j] = msk ? ws_key(fv[j]) : WS_INF;
                        lab[j] = msk ? mv[j] : 0;
                        sLV[lr * P + 1 + 4 * q + j] = make_uint2(lab[j] != 0 ? key[j] : WS_INF, key[j]);
                    }
                    if (inside && lr >= 1 && lr <= T) {  // own pixels: publish value keys and seed labels
                        const int64_t g = fbase + rowoff(r, W) + c;
                        *reinterpret_cast<uint4 *>(val + g) = make_uint4(key[0], key[1], key[2], key[3]);
                        *reinterpret_cast<int4 *>(in.out + g) = make_int4(lab[0], lab[1], lab[2], lab[3]);
                    }
                }
            }
            if (halo_thread) {
                const bool msk = h_in && hk != 0;
                const unsigned key = msk ? ws_key(hf) : WS_INF;
                sLV[h_lr * P + h_lc] = make_uint2(msk && hm != 0 ? key : WS_INF, key);
            }
        } else
#ifdef PCSEG_EXP_RELAX_LOADS  // sensitivity build: a revisited tile is fetched this many times (what do its loads cost?)
#pragma unroll 1
        for (int load_rep = 0; load_rep < PCSEG_EXP_RELAX_LOADS; ++load_rep)
#endif
        {
#ifdef PCSEG_EXP_RELAX_LOADS
            asm volatile("" ::: "memory");
#endif
            uint4 l4[TRIPS], v4[TRIPS];
#pragma unroll
            for (int t = 0; t < TRIPS; ++t) {
                const int idx = min(tid + NT * t, QUADS - 1);
                const int lr = idx / QW, q = idx % QW;
                const int64_t p = fbase + rowoff(min(max(r0 + lr - 1, 0), H - 1), W) + min(max(c0 + 4 * q, 0), W - 4);
                l4[t] = *reinterpret_cast<const uint4 *>(L + p);
                v4[t] = *reinterpret_cast<const uint4 *>(val + p);
            }
            const unsigned hl = L[fbase + h_p], hv = val[fbase + h_p];
#pragma unroll
            for (int t = 0; t < TRIPS; ++t) {
                const int idx = tid + NT * t;
                if (idx < QUADS) {
                    const int lr = idx / QW, q = idx % QW;
                    const int r = r0 + lr - 1, c = c0 + 4 * q;
                    const bool inside = r >= 0 && r < H && c >= 0 && c < W;
                    uint2 *dst = sLV + lr * P + 1 + 4 * q;
                    dst[0] = inside ? make_uint2(l4[t].x, v4[t].x) : make_uint2(WS_INF, WS_INF);
                    dst[1] = inside ? make_uint2(l4[t].y, v4[t].y) : make_uint2(WS_INF, WS_INF);
                    dst[2] = inside ? make_uint2(l4[t].z, v4[t].z) : make_uint2(WS_INF, WS_INF);
                    dst[3] = inside ? make_uint2(l4[t].w, v4[t].w) : make_uint2(WS_INF, WS_INF);
                }
            }
            if (halo_thread) sLV[h_lr * P + h_lc] = h_in ? make_uint2(hl, hv) : make_uint2(WS_INF, WS_INF);
        }
    } else
    for (int i = tid; i < S * S; i += NT) {
        int lr = i / S, lc = i % S;
        int r = r0 + lr - 1, c = c0 + lc - 1;
        uint2 lv = make_uint2(WS_INF, WS_INF);
        if (r >= 0 && r < H && c >= 0 && c < W) {
            if (FIRST) {
                lv = initial(r, c);
                if (lr >= 1 && lr <= T && lc >= 1 && lc <= T) {  // own pixels: publish value key and seed label
                    const int64_t g = fbase + rowoff(r, W) + c;
                    val[g] = lv.y;
                    in.out[g] = lv.y != WS_INF ? in.markers[g] : 0;
                }
            } else {
                lv.x = L[fbase + rowoff(r, W) + c];
                lv.y = val[fbase + rowoff(r, W) + c];
            }
        }
        sLV[lr * P + lc] = lv;
    }
    __syncthreads();
    // thread = (half edge e, cell j): e 0..3 horizontal (top-left, top-right, bottom-left, bottom-right), e 4..7 vertical
    // (left-top, left-bottom, right-top, right-bottom).  Each of the first 8 * T / 2 threads remembers the level its rim
    // cell had on arrival: what the tile changed on its rim is then known without going back to global memory
    const bool rim_thread = tid < 8 * G::HE;
    const int e = tid / G::HE, ej = tid % G::HE;
    const int qy = e < 4 ? (e >> 1) : (e & 1), qx = e < 4 ? (e & 1) : ((e >> 1) & 1);
    const int rim_lr = e < 4 ? (qy ? T : 1) : qy * G::HE + ej + 1;
    const int rim_lc = e < 4 ? qx * G::HE + ej + 1 : (qx ? T : 1);
    const unsigned rim_before = rim_thread ? sLV[rim_lr * P + rim_lc].x : 0u;
    // wave = (direction, group of 64 lines, 64-cell segment of the lines): the segments of a line are swept at the same
    // time, each starting from the cell in front of it -- its neighbour segment's last cell, or the halo
    const int wave = tid >> 6, lane = tid & 63;
    const int dir = wave & 3, seg = wave >> 2;
    const int line = 1 + (seg % G::G) * 64 + lane, along = (seg / G::G) * 64;
    const int start = dir == 0 ? line * P + along : dir == 1 ? line * P + (S - 1) - along
                    : dir == 2 ? along * P + line : ((S - 1) - along) * P + line;
    bool changed_any = false;
    bool capped = true;  // left before the tile's fixed point (round 0 stops after max_iter sweeps per direction)
    for (int iter = 0; iter < max_iter; ++iter) {
        // one code path per direction: the step is a compile-time constant there, so the LDS addresses of a batch are
        // base + constant
        bool changed;
        if constexpr (PCSEG_WS_FSM && T == 64) {
            if (dir == 0) changed = ws_quadrant_sweep<1, 1, P>(sLV, lane);
            else if (dir == 1) changed = ws_quadrant_sweep<1, -1, P>(sLV, lane);
            else if (dir == 2) changed = ws_quadrant_sweep<-1, 1, P>(sLV, lane);
            else changed = ws_quadrant_sweep<-1, -1, P>(sLV, lane);
        } else {
            if (dir == 0) changed = ws_sweep<1>(sLV, start);
            else if (dir == 1) changed = ws_sweep<-1>(sLV, start);
            else if (dir == 2) changed = ws_sweep<P>(sLV, start);
            else changed = ws_sweep<-P>(sLV, start);
        }
        if (!__syncthreads_or(changed)) { capped = false; break; }
        changed_any = true;
    }
    if (!FIRST && !changed_any) return;
    // half edges that changed -> tiles of the next round; then store
    if (changed_any) {
        const int r = r0 + rim_lr - 1, c = c0 + rim_lc - 1;
        // a tile that stopped early is not consistent inside: all four corner tiles have to look at it again
        bool ch = rim_thread && capped;
        if (rim_thread && !capped && r >= 0 && r < H && c >= 0 && c < W) ch = sLV[rim_lr * P + rim_lc].x != rim_before;
        // the lanes of this thread's half edge inside its wave
        const unsigned long long mine = G::HE >= 64 ? ~0ull : (((1ull << (G::HE & 63)) - 1ull) << ((tid & 63) / G::HE * G::HE));
        const unsigned long long edge_changed = __ballot(ch) & mine;
        if (rim_thread && edge_changed && ej == 0) {
            // the tile of the other tiling that holds this corner quadrant (rows r0 + qy * T / 2 .., cols c0 + qx * T / 2 ..)
            const int oy = (r0 + qy * G::HE + nxt.off) / T, ox = (c0 + qx * G::HE + nxt.off) / T;
            if (oy >= 0 && oy < nxt.ny && ox >= 0 && ox < nxt.nx) {
                const int64_t m = ((int64_t)b * nxt.ny + oy) * nxt.nx + ox;
                if (list_out) {
                    // the next round walks a LIST of marked tiles (ws_relax_list_kernel): the first marker of a tile appends it
                    // (test-and-set on the mark's byte inside its 32-bit word: the mark arrays are 256-byte aligned)
                    const unsigned bit = 1u << (8 * (int)(m & 3));
                    const unsigned old = atomicOr(reinterpret_cast<unsigned *>(dirty_out + (m & ~(int64_t)3)), bit);
                    if (!(old & bit)) list_out[atomicAdd(count_out, 1)] = (int)m;
                } else {
                    dirty_out[m] = 1;
                }
            }
        }
    }
    __syncthreads();
    if (in.vec) {
#ifdef PCSEG_EXP_RELAX_STORES  // sensitivity build: a tile's levels are written this many times
#pragma unroll 1
        for (int store_rep = 0; store_rep < PCSEG_EXP_RELAX_STORES; ++store_rep) {
            asm volatile("" ::: "memory");
#endif
#pragma unroll
        for (int t = 0; t < T * QW / NT; ++t) {
            const int idx = tid + NT * t, lr = idx / QW, q = idx % QW;
            const int r = r0 + lr, c = c0 + 4 * q;
            const uint2 *src = sLV + (lr + 1) * P + 1 + 4 * q;
            if (r >= 0 && r < H && c >= 0 && c < W)
                *reinterpret_cast<uint4 *>(L + fbase + rowoff(r, W) + c) = make_uint4(src[0].x, src[1].x, src[2].x, src[3].x);
        }
#ifdef PCSEG_EXP_RELAX_STORES
        }
#endif
        return;
    }
    for (int i = tid; i < T * T; i += NT) {
        int lr = i / T, lc = i % T;
        int r = r0 + lr, c = c0 + lc;
        if (r >= 0 && r < H && c >= 0 && c < W) L[fbase + rowoff(r, W) + c] = sLV[(lr + 1) * P + lc + 1].x;
    }
}

template <int T>
__global__ void __launch_bounds__(RelaxGeom<T>::THREADS) ws_relax_kernel(WsInputs in, const bool FIRST, unsigned *__restrict__ val,
                                                                         unsigned *__restrict__ L, uint8_t *__restrict__ dirty_in,
                                                                         uint8_t *__restrict__ dirty_out, int *__restrict__ any_changed,
                                                                         int H, int W, WsTiling cur, WsTiling nxt, int max_iter,
                                                                         int *__restrict__ list_out, int *__restrict__ count_out)
{
    extern __shared__ __attribute__((aligned(16))) uint2 relax_lds[];  // RelaxGeom<T>::N cells
    const TileIndex t = xcd_tile_index();  // a tile's halo is its neighbours' rim: keep them on one XCD's L2
    ws_relax_tile<T>(relax_lds + RelaxGeom<T>::PAD, in, FIRST, val, L, dirty_in, dirty_out, any_changed, H, W, cur, nxt, max_iter, t.x, t.y,
                     t.z, list_out, count_out);
}

// Late rounds visit a few per cent of the tiles.  Launched over every tile they retire thousands of workgroups that read one
// byte and leave -- each of which first has to be given 36 KB of LDS and four wave slots on a CU that seven other batches'
// kernels are using (see WsTileList: the empty grids of the second level cost the STEP 3 %).  From round WS_LIST_FROM on a
// round is a fixed small grid walking the list of tiles the round before marked (appended by the first marker of a tile).
#ifndef PCSEG_WS_LIST_FROM
#define PCSEG_WS_LIST_FROM 4  // first list-walking round (12 = never)
#endif
constexpr int WS_LIST_FROM = PCSEG_WS_LIST_FROM, WS_LIST_GRID = 1024;

template <int T>
__global__ void __launch_bounds__(RelaxGeom<T>::THREADS) ws_relax_list_kernel(WsInputs in, unsigned *__restrict__ val, unsigned *__restrict__ L,
                                                                              uint8_t *__restrict__ dirty_in, uint8_t *__restrict__ dirty_out,
                                                                              int *__restrict__ any_changed, int H, int W, WsTiling cur,
                                                                              WsTiling nxt, int max_iter, const int *__restrict__ list_in,
                                                                              const int *__restrict__ count_in, int *__restrict__ list_out,
                                                                              int *__restrict__ count_out)
{
    extern __shared__ __attribute__((aligned(16))) uint2 relax_lds[];
    const int n = *count_in, per_frame = cur.nx * cur.ny;
    for (int i = blockIdx.x; i < n; i += gridDim.x) {
        const int m = list_in[i], b = m / per_frame, t = m % per_frame;
        ws_relax_tile<T>(relax_lds + RelaxGeom<T>::PAD, in, false, val, L, dirty_in, dirty_out, any_changed, H, W, cur, nxt, max_iter,
                         t % cur.nx, t / cur.nx, b, list_out, count_out);
        __syncthreads();  // the next listed tile reuses the LDS tile
    }
}

// The fixed point is driven WITHOUT the host: a fixed number of grid rounds is enqueued (a round whose tiles carry no
// mark costs a few microseconds: every block reads one byte and leaves), and whatever is still marked after them --
// a few tiles of a few frames, if anything -- is finished by this kernel: one block per frame walks the frame's marked
// tiles round by round until a round marks nothing.  Rounds of one frame only depend on that frame's tiles, so the
// block's own barrier is the only synchronisation (stores and loads of one workgroup go through the same L1).
constexpr int WS_TAIL_LIST = 1024;  // marked tiles a tail kernel lists per round (more: it walks every tile)

template <int T>
__global__ void __launch_bounds__(RelaxGeom<T>::THREADS) ws_relax_tail_kernel(WsInputs in, unsigned *__restrict__ val,
                                                                              unsigned *__restrict__ L, uint8_t *__restrict__ dirtyA,
                                                                              uint8_t *__restrict__ dirtyB, int *__restrict__ any_changed,
                                                                              int *__restrict__ not_converged, int *__restrict__ exact_flags,
                                                                              int H, int W, WsTiling t0, WsTiling t1, int first_round,
                                                                              int max_rounds)
{
    extern __shared__ __attribute__((aligned(16))) uint2 relax_lds[];
    __shared__ int tail_list[WS_TAIL_LIST];
    __shared__ int tail_count;
    const int b = blockIdx.x;
    uint8_t *din = dirtyA, *dout = dirtyB;
    for (int round = first_round;; ++round) {
        const WsTiling cur = (round & 1) ? t1 : t0, nxt = (round & 1) ? t0 : t1;
        const int ntiles = cur.nx * cur.ny;
        const uint8_t *marks = din + (int64_t)b * ntiles;
        // the round's work list: the marked tiles, gathered in parallel (walking ALL tiles and letting each look at its own
        // mark costs a dependent global load per tile -- 256 round trips per round for a handful of marked tiles)
        if (threadIdx.x == 0) tail_count = 0;
        __syncthreads();
        for (int t = threadIdx.x; t < ntiles; t += RelaxGeom<T>::THREADS)
            if (marks[t] != 0) {
                const int k = atomicAdd(&tail_count, 1);
                if (k < WS_TAIL_LIST) tail_list[k] = t;
            }
        __syncthreads();
        const int marked = tail_count;
        if (marked == 0) return;
        if (round - first_round >= max_rounds) {  // cannot happen for a monotone fixed point; never spin for ever
            // L of this frame is not a fixed point: whatever the union-find makes of it must not be reported as exact --
            // the frame's tie flag is raised, so the exact flood (mode 0) recomputes it and mode 2 reports it
            if (threadIdx.x == 0) { *not_converged = 1; exact_flags[b] = 1; }
            return;
        }
        const int walk = marked <= WS_TAIL_LIST ? marked : ntiles;  // (a list that overflowed: every tile, each checks its mark)
        for (int k = 0; k < walk; ++k) {
            const int t = marked <= WS_TAIL_LIST ? tail_list[k] : k;
            ws_relax_tile<T>(relax_lds + RelaxGeom<T>::PAD, in, false, val, L, din, dout, any_changed, H, W, cur, nxt, 100000, t % cur.nx,
                             t / cur.nx, b);
            __syncthreads();  // the tile's stores (L, marks) before the next tile loads its halo / the next round scans
        }
        uint8_t *tmp = din; din = dout; dout = tmp;
    }
}


// kernels of the second level are launched over the flagged frames only: grid index -> frame id through a list
// The list and its length (frame_list[-1]) stay on the device.  A launch over a list only spans WS_LIST_SPAN entries in
// the frame dimension of its grid and every block walks the list with that stride: a launch that finds the list
// empty (the usual case on tie-free data) retires few blocks, a batch in which every frame is listed loops.
constexpr int WS_LIST_SPAN = 8;
template <typename Body>
__device__ __forceinline__ void ws_for_frames(const int *frame_list, int grid_index, int grid_size, Body &&body)
{
    if (!frame_list) {
        body(grid_index);
        return;
    }
    const int n = frame_list[-1];
    for (int gi = grid_index; gi < n; gi += grid_size) body(frame_list[gi]);
}
static inline int ws_frame_span(const int *frame_list, int B) { return frame_list ? (B < WS_LIST_SPAN ? B : WS_LIST_SPAN) : B; }

// stage-2 work is restricted to the 64x64 tiles that hold a pixel of an unresolved component (active == nullptr: all)
__device__ __forceinline__ bool ws_active(const uint8_t *active, int b, int r, int c, int tilesX, int tilesY)
{
    return active == nullptr || active[((int64_t)b * tilesY + r / WS_T) * tilesX + c / WS_T] != 0;
}

// The second level's work is a few dozen 64 x 64 tiles of a few frames (benchmark batch: 31 tiles in 7 frames).  Its passes
// used to be launched over the worst-case grid of the listed frames -- every tile or pixel of up to eight frames, almost all
// of whose blocks leave at once.  Alone that costs little; with eight batches in flight every such block still has to find a
// CU with its LDS and wave slots free among the other batches' kernels (2 048 blocks of 1 024 threads and 58 KB for one
// K2 round): same box, fewer grid rounds made the STEP faster although the serial time went up (4.53-4.57 ms against
// 4.62-4.75, profiles/r04/ab_logs/r4f_*).  So the active tiles are listed once on the device (entry = frame * tiles per frame
// + tile, the count in front of the list) and the second level's kernels are small fixed grids that walk the list.
struct WsTileList {
    const int *list;   // list[-1] = number of entries
    int ntpf, tilesX;  // tiles per frame, tiles per tile row
};
constexpr int WS_TILE_GRID = 512;  // blocks of a list-walking launch (two per CU: a quantised batch lists every tile)

template <typename Body>
__device__ __forceinline__ void ws_for_tiles(const WsTileList &tl, Body &&body)
{
    const int n = tl.list[-1];
    for (int i = blockIdx.x; i < n; i += gridDim.x) {
        const int e = tl.list[i], b = e / tl.ntpf, t = e % tl.ntpf;
        body(b, t % tl.tilesX, t / tl.tilesX);
    }
}

__global__ void __launch_bounds__(256) ws_list_tiles_kernel(const int *__restrict__ frame_list, const uint8_t *__restrict__ active, int ntpf,
                                                             int *__restrict__ tile_list)
{
    ws_for_frames(frame_list, blockIdx.x, gridDim.x, [&](const int b) {
        for (int t = threadIdx.x; t < ntpf; t += 256)
            if (active[(int64_t)b * ntpf + t]) tile_list[atomicAdd(&tile_list[-1], 1)] = b * ntpf + t;
    });
}

// (2) label assignment by union-find: every non-seed reachable pixel is united with ALL its
// neighbours that hold the minimum neighbour key.  If no pixel has minimum-key neighbours in two basins (the proof
// check's premise) the components are exactly the basins, each holding the seeds of one marker; a component that
// holds two different marker ids flags the frame instead.  One LDS tile pass + one border pass, like A2 -- but the
// union-find orders its nodes by a VIRTUAL index in which every seed precedes every other pixel (seed: index,
// non-seed: index + UF_NS).  Roots are minima, so the root of a component that holds a seed IS a seed, and a pixel's
// label is simply out[root]: no separate pass that publishes marker ids at the roots.
constexpr int UF_TW = 64, UF_TH = 32, UF_SW = UF_TW + 2, UF_SH = UF_TH + 2;
constexpr int UF_LNS = UF_TW * UF_TH;  // non-seed offset of tile-local virtual indices
constexpr int UF_NS = 1 << 30;         // non-seed offset of frame-wide virtual indices (H * W < 2^30)

__device__ __forceinline__ int vfind_lds(int *par, int v)
{
    // path halving: re-pointing v at its grandparent keeps it inside its set (concurrent unions only ever lower parents)
    int p;
    while ((p = ld_lds(par + (v & (UF_LNS - 1)))) != v) {
        const int g = ld_lds(par + (p & (UF_LNS - 1)));
        if (g != p) st_lds(par + (v & (UF_LNS - 1)), g);
        v = g;
    }
    return v;
}
__device__ __forceinline__ void vunite_lds(int *par, int a, int b)
{
    for (;;) {
        a = vfind_lds(par, a);
        b = vfind_lds(par, b);
        if (a == b) return;
        if (a < b) { int t = a; a = b; b = t; }
        int old = atomicMin(&par[a & (UF_LNS - 1)], b);
        if (old == a) return;
        a = old;
    }
}
// The walks over the frame-wide parent image are FENCED like the plain ones (common.h, walk_ok): an entry stored for the node
// with virtual index x is a virtual index <= x (roots are minima of the virtual order, unions only lower entries) whose
// pixel lies inside the frame.  Anything else -- the image being rewritten by a second instance of the same captured chain
// (profiles/r03/exp_graph_r3a.log), a stale workspace -- stops the walk and raises the frame's tie flag (the exact flood
// then recomputes the frame in mode 0, mode 2 reports it) instead of loading from a wild address or walking for ever.
__device__ __forceinline__ bool vwalk_ok(int x, int p, int n) { return (unsigned)p <= (unsigned)x && (p & (UF_NS - 1)) < n; }

// p, q: pixel indices; the walk starts from their parents, which are virtual indices of nodes in the same sets
__device__ __forceinline__ void vunite_glb(int *par, int p, int q, int n, int *corrupt)
{
    int a = ld_agent(par + p), b = ld_agent(par + q);
    // (the pixels' own entries: any virtual index of the frame -- a seed elsewhere may precede both)
    bool ok = a >= 0 && b >= 0 && (a & (UF_NS - 1)) < n && (b & (UF_NS - 1)) < n;
    while (ok) {
        // both walks in lockstep (see find2_glb)
        for (;;) {
            if (a == b) return;
            const int pa = ld_agent(par + (a & (UF_NS - 1))), pb = ld_agent(par + (b & (UF_NS - 1)));
            if (pa == a && pb == b) break;
            if (!vwalk_ok(a, pa, n) || !vwalk_ok(b, pb, n)) { ok = false; break; }
            const int ga = ld_agent(par + (pa & (UF_NS - 1))), gb = ld_agent(par + (pb & (UF_NS - 1)));
            if (!vwalk_ok(pa, ga, n) || !vwalk_ok(pb, gb, n)) { ok = false; break; }
            if (pa != a) {
                if (ga != pa) __hip_atomic_store(par + (a & (UF_NS - 1)), ga, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                a = ga;
            }
            if (pb != b) {
                if (gb != pb) __hip_atomic_store(par + (b & (UF_NS - 1)), gb, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                b = gb;
            }
        }
        if (!ok) break;
        if (a == b) return;
        if (a < b) { int t = a; a = b; b = t; }
        int old = atomicMin(par + (a & (UF_NS - 1)), b);
        if (old == a) return;
        if (!vwalk_ok(a, old, n)) break;
        a = old;
    }
    walk_corrupt(corrupt);
}
// read-only walk to the root of virtual index x (x's pixel must lie inside the frame)
__device__ __forceinline__ int vwalk_root(const int *par, int x, int n, bool &bad)
{
    int q;
    while ((q = par[x & (UF_NS - 1)]) != x) {
        if (!vwalk_ok(x, q, n)) {
            bad = true;
            break;
        }
        x = q;
    }
    return x;
}

template <typename KeyT>
__device__ __forceinline__ void ws_uf_tile_frame(KeyT *sK, int *par, uint8_t *sM, const int b, const int tile_x, const int tile_y,
                                                 const KeyT *__restrict__ K,
                                                 const int *__restrict__ F, const uint8_t *__restrict__ active,
                                                 int *__restrict__ parent, uint8_t *__restrict__ minmask, uint8_t *__restrict__ bad,
                                                 int H, int W, int tilesX, int tilesY)
{
    const KeyT KINF = ~(KeyT)0;
    const int r0 = tile_y * UF_TH, c0 = tile_x * UF_TW;
    if (!ws_active(active, b, r0, c0, tilesX, tilesY)) return;  // UF tiles (64x32) nest inside the 64x64 tiles
    const int64_t fbase = (int64_t)b * H * W;
    if (sizeof(KeyT) == 4 && (W & 3) == 0 && c0 + UF_TW <= W && (((uintptr_t)K) & 15) == 0) {
        // full-width tile, 4-byte keys, 16-byte aligned rows: the 64 interior columns of the 34 rows as 16-byte quads (three
        // trips instead of nine), the two halo columns as one scalar trip; clamped addresses, one batch of loads
        constexpr int QUADS = UF_SH * (UF_TW / 4), TRIPS = (QUADS + 255) / 256;
        uint4 kq[TRIPS];
#pragma unroll
        for (int t = 0; t < TRIPS; ++t) {
            const int i = min((int)threadIdx.x + 256 * t, QUADS - 1);
            const int r = r0 + i / (UF_TW / 4) - 1;
            kq[t] = *reinterpret_cast<const uint4 *>(K + fbase + rowoff(min(max(r, 0), H - 1), W) + c0 + 4 * (i % (UF_TW / 4)));
        }
        const int hl = min((int)threadIdx.x, UF_SH * 2 - 1);
        const int hr = r0 + (hl >> 1) - 1, hx = (hl & 1) ? UF_TW : -1, hc = c0 + hx;
        const KeyT hv = K[fbase + rowoff(min(max(hr, 0), H - 1), W) + min(max(hc, 0), W - 1)];
#pragma unroll
        for (int t = 0; t < TRIPS; ++t) {
            const int i = (int)threadIdx.x + 256 * t;
            if (i < QUADS) {
                const int lr = i / (UF_TW / 4), r = r0 + lr - 1;
                const bool in = r >= 0 && r < H;
                KeyT *dst = sK + lr * UF_SW + 1 + 4 * (i % (UF_TW / 4));
                dst[0] = in ? (KeyT)kq[t].x : KINF; dst[1] = in ? (KeyT)kq[t].y : KINF;
                dst[2] = in ? (KeyT)kq[t].z : KINF; dst[3] = in ? (KeyT)kq[t].w : KINF;
            }
        }
        if (threadIdx.x < UF_SH * 2) sK[(hl >> 1) * UF_SW + hx + 1] = (hr >= 0 && hr < H && hc >= 0 && hc < W) ? hv : KINF;
    } else {
        // one batch of loads, not a loop of round trips: clamped (always valid) addresses, no branch around the loads,
        // out-of-frame cells fixed up by a select (see ws_relax_tile)
        constexpr int TRIPS = (UF_SH * UF_SW + 255) / 256;
        KeyT kv[TRIPS];
#pragma unroll
        for (int t = 0; t < TRIPS; ++t) {
            const int i = min((int)threadIdx.x + 256 * t, UF_SH * UF_SW - 1);
            const int r = r0 + i / UF_SW - 1, c = c0 + i % UF_SW - 1;
            kv[t] = K[fbase + rowoff(min(max(r, 0), H - 1), W) + min(max(c, 0), W - 1)];
        }
#pragma unroll
        for (int t = 0; t < TRIPS; ++t) {
            const int i = (int)threadIdx.x + 256 * t;
            const int r = r0 + i / UF_SW - 1, c = c0 + i % UF_SW - 1;
            if (i < UF_SH * UF_SW) sK[i] = (r >= 0 && r < H && c >= 0 && c < W) ? kv[t] : KINF;
        }
    }
    __syncthreads();
    // (1) per pixel: own virtual index and which neighbours hold the minimum neighbour key (bit0 up, 1 left, 2 right,
    // 3 down; 0 for seeds / unreachable).  The masks also go to global memory for the border pass, which then needs
    // two bytes per cross-tile pair instead of ten keys.
    int self[UF_LNS / 256];  // -1: unreachable / outside
    int fv[UF_LNS / 256];    // labels of the thread's pixels, fetched as one batch (clamped address: no branch)
#pragma unroll
    for (int k = 0; k < UF_LNS / 256; ++k) {
        const int t = threadIdx.x + k * 256;
        fv[k] = F[fbase + rowoff(min(r0 + t / UF_TW, H - 1), W) + min(c0 + t % UF_TW, W - 1)];
    }
#pragma unroll
    for (int k = 0; k < UF_LNS / 256; ++k) {
        const int t = threadIdx.x + k * 256;
        const int lr = t / UF_TW, lc = t % UF_TW;
        const int r = r0 + lr, c = c0 + lc;
        const int i = (lr + 1) * UF_SW + lc + 1;
        int v = -1;
        if ((unsigned)(sK[i] >> (8 * sizeof(KeyT) - 32)) != WS_INF)  // reachable => inside the frame
            v = fv[k] != 0 ? t : t + UF_LNS;  // labelled pixels (seeds) order first
        uint8_t m8 = 0;
        if (v >= UF_LNS) {  // seeds take no label from neighbours
            const KeyT ku = sK[i - UF_SW], kl = sK[i - 1], kr = sK[i + 1], kd = sK[i + UF_SW];
            const KeyT m = min(min(ku, kd), min(kl, kr));
            if (m != KINF) m8 = (ku == m ? 1 : 0) | (kl == m ? 2 : 0) | (kr == m ? 4 : 0) | (kd == m ? 8 : 0);
        }
        self[k] = v;
        sM[t] = m8;
        if (r < H && c < W) {
            minmask[fbase + rowoff(r, W) + c] = m8;
            // the "component cannot be resolved" marks of this level start clear: every root the label passes will look at
            // lies in a tile this pass visits (instead of a memset of the whole array per level) -- and is a labelled
            // pixel (roots are minima of the virtual order, in which labelled pixels come first; a component without one
            // is never marked): only those bytes are written, a few thousand a frame instead of every pixel's
            if (fv[k] != 0) bad[fbase + rowoff(r, W) + c] = 0;
        }
    }
    __syncthreads();
    // (2) row runs without atomics: a wave covers one 64-pixel tile row per trip; pixels joined by horizontal links form
    // a run, every pixel of it points straight at the run's smallest virtual index (its first seed if it has one,
    // else its first pixel)
#pragma unroll
    for (int k = 0; k < UF_LNS / 256; ++k) {
        const int t = threadIdx.x + k * 256;
        const int lc = t % UF_TW;
        const bool joins_left = lc > 0 && ((sM[t] & 2) || (sM[t - 1] & 4));
        const unsigned long long heads = __ballot(!joins_left);
        const unsigned long long seeds = __ballot(self[k] >= 0 && self[k] < UF_LNS);
        const unsigned long long upto = heads & (lc == 63 ? ~0ull : ((2ull << lc) - 1ull));
        const int start = 63 - __clzll((long long)upto);
        const unsigned long long later = lc == 63 ? 0ull : (heads >> (lc + 1));
        const int end = later ? lc + (__ffsll((long long)later) - 1) : 63;  // last pixel of the run
        const unsigned long long run = (end == 63 ? ~0ull : ((2ull << end) - 1ull)) & ~((1ull << start) - 1ull);
        const unsigned long long run_seeds = seeds & run;
        int root = -1;
        if (self[k] >= 0) {
            if (run_seeds) root = (t - lc) + (__ffsll((long long)run_seeds) - 1);  // a seed: virtual index = position
            else root = (t - lc) + start + UF_LNS;
        }
        par[t] = root;
    }
    __syncthreads();
    // (3) vertical links, skipped where three links that are made anyway already imply them (the left neighbour is in
    // the same run, so is the upper-left one with the upper one, and the left neighbour has its own vertical link)
#pragma unroll
    for (int k = 0; k < UF_LNS / 256; ++k) {
        const int t = threadIdx.x + k * 256;
        const int lr = t / UF_TW, lc = t % UF_TW;
        if (lr == 0 || self[k] < 0) continue;
        const uint8_t m = sM[t], mu = sM[t - UF_TW];
        if (!((m & 1) || (mu & 8))) continue;
        if (lc > 0) {
            const uint8_t ml = sM[t - 1], mul = sM[t - UF_TW - 1];
            const bool same_run = (m & 2) || (ml & 4), same_run_up = (mu & 2) || (mul & 4), left_vertical = (ml & 1) || (mul & 8);
            if (same_run && same_run_up && left_vertical) continue;
        }
#if !defined(PCSEG_EXP_UFTILE) || !(PCSEG_EXP_UFTILE & 1)  // (ablation builds: wrong labels, the pass's time without a phase)
        vunite_lds(par, par[t], par[t - UF_TW]);
#endif
    }
    __syncthreads();
    for (int t = threadIdx.x; t < UF_TH * UF_TW; t += 256) {
        const int r = r0 + t / UF_TW, c = c0 + t % UF_TW;
        if (r >= H || c >= W) continue;
        int v = -1;
        if (par[t] >= 0) {
#if defined(PCSEG_EXP_UFTILE) && (PCSEG_EXP_UFTILE & 2)
            const int root = par[t];
#else
            const int root = vfind_lds(par, par[t]);
#endif
            const int lt = root & (UF_LNS - 1);
            v = ((r0 + lt / UF_TW) * W + c0 + lt % UF_TW) | (root >= UF_LNS ? UF_NS : 0);
        }
        parent[fbase + rowoff(r, W) + c] = v;
    }
}

// LIST = false: every frame (first level), one frame per grid slice, straight-line code (32 VGPRs, 8 waves / SIMD; the
// looping variant needs 80); LIST = true: the listed frames, see ws_for_frames
template <typename KeyT, bool LIST>
__global__ void __launch_bounds__(256) ws_uf_tile_kernel(WsTileList tiles, const KeyT *__restrict__ K, const int *__restrict__ F,
                                                          const uint8_t *__restrict__ active, int *__restrict__ parent,
                                                          uint8_t *__restrict__ minmask, uint8_t *__restrict__ bad, int H, int W,
                                                          int tilesX, int tilesY)
{
    __shared__ KeyT sK[UF_SH * UF_SW];
    __shared__ int par[UF_TH * UF_TW];
    __shared__ uint8_t sM[UF_LNS];
    if constexpr (!LIST) {
        const TileIndex t = xcd_tile_index();
        ws_uf_tile_frame<KeyT>(sK, par, sM, t.z, t.x, t.y, K, F, active, parent, minmask, bad, H, W, tilesX, tilesY);
    } else {
        // the listed 64 x 64 tiles, two union-find tiles (64 x 32) each
        ws_for_tiles(tiles, [&](const int b, const int tx, const int ty) {
            for (int half = 0; half < WS_T / UF_TH; ++half) {
                if ((ty * (WS_T / UF_TH) + half) * UF_TH < H)
                    ws_uf_tile_frame<KeyT>(sK, par, sM, b, tx, ty * (WS_T / UF_TH) + half, K, F, active, parent, minmask, bad, H, W, tilesX,
                                           tilesY);
                __syncthreads();  // the next tile reuses the tile arrays
            }
        });
    }
}

// cross-tile links from the neighbour masks the tile pass left behind
// (launch bounds: 104 scalar registers without them = six workgroups per CU for a pass of dependent walks; 78 with -- see label4_chains)
__global__ void __launch_bounds__(256, 8) ws_uf_border_kernel(const int *__restrict__ frame_list, const uint8_t *__restrict__ minmask, const uint8_t *__restrict__ active,
                                                            int *__restrict__ parent, int H, int W, int tilesX, int tilesY,
                                                            int *__restrict__ exact_flags)
{
    // dense enumeration of the border pixels: rows that start a tile row (lanes along the row), then the first column
    // of every tile column for the remaining rows (lanes along the column)
    const int n_top_rows = (H - 1) / UF_TH, n_edges = (W - 1) / UF_TW;
    int i = blockIdx.x * 256 + threadIdx.x;
    int r, c;
    if (i < n_top_rows * W) {
        r = (i / W + 1) * UF_TH;
        c = i % W;
    } else {
        i -= n_top_rows * W;
        if (i >= n_edges * H) return;
        r = i % H;
        if ((r % UF_TH) == 0 && r > 0) return;  // handled with its row above
        c = (i / H + 1) * UF_TW;
    }
    ws_for_frames(frame_list, blockIdx.y, gridDim.y, [&](const int b) {
    if (!ws_active(active, b, r, c, tilesX, tilesY)) return;
    const bool top = (r % UF_TH) == 0 && r > 0;
    const bool left = (c % UF_TW) == 0 && c > 0;
    if (!top && !left) return;
    const int64_t fbase = (int64_t)b * H * W;
    int *par = parent + fbase;
    const int p = r * W + c;
    const uint8_t *mm = minmask + fbase;
    // the four masks as one batch of loads (offsets clamped at the frame's edge, where the value is not used): tested one
    // after the other they were up to four dependent memory round trips per pixel
    const int up = r > 0 ? W : 0, lf = c > 0 ? 1 : 0;
    const uint8_t mp = mm[p], mu = mm[p - up], ml = mm[p - lf], mul = mm[p - up - lf];
    // A cross-tile link is skipped when three links that are made anyway already join the two pixels: for a vertical
    // one the horizontal links p ~ p-1 and p-W ~ p-W-1 (both inside one tile) and the vertical link of p-1; likewise
    // for a horizontal one.  (At a tile corner both links cross tiles and would justify each other: keep both there.)
    if (top && ws_active(active, b, r - 1, c, tilesX, tilesY)) {
        if ((mp & 1) || (mu & 8)) {
            bool implied = false;
            if (!left && c > 0) implied = ((mp & 2) || (ml & 4)) && ((mu & 2) || (mul & 4)) && ((ml & 1) || (mul & 8));
            if (!implied) vunite_glb(par, p, p - W, H * W, exact_flags + b);
        }
    }
    if (left && ws_active(active, b, r, c - 1, tilesX, tilesY)) {
        if ((mp & 2) || (ml & 4)) {
            bool implied = false;
            if (!top && r > 0) implied = ((mp & 1) || (mu & 8)) && ((ml & 1) || (mul & 8)) && ((mu & 2) || (mul & 4));
            if (!implied) vunite_glb(par, p, p - 1, H * W, exact_flags + b);
        }
    }
    });
}

// Labels from the roots.  A component that holds two differently labelled pixels cannot be resolved at this level:
// its root is marked in `bad`, its frame is flagged.
//   UF_OPTIMISTIC  first level: write out[root] to every unlabelled pixel and look for conflicts in the same pass; the
//                  rare frames with a conflict are repaired by UF_REPAIR afterwards (labelled == original seed here)
//   UF_REPAIR      pixels of bad components go back to their seed state, their 64x64 tiles become active
//   UF_DETECT      second level, pass 1: conflicts only (labelled pixels include first-level results: no repair by
//   UF_ASSIGN      reset possible), pass 2: label the pixels of the components that are not bad
enum { UF_OPTIMISTIC = 0, UF_REPAIR = 1, UF_DETECT = 2, UF_ASSIGN = 3 };

//   UF_REPAIR also leaves, per pixel of a listed frame, "belongs to an unresolved component" in `in_bad` (if given): the
//   second level's lake propagation only has to run inside those components (see ws_k2_relax_tile)
template <int MODE>
__device__ __forceinline__ void ws_uf_label_pixel(const int b, const int64_t i, const int *__restrict__ parent, int *__restrict__ F,
                                                  const uint8_t *__restrict__ active, uint8_t *__restrict__ bad,
                                                  const int *__restrict__ markers, const uint8_t *__restrict__ mask,
                                                  int *__restrict__ tie_flags, uint8_t *__restrict__ mark_active, int64_t n, int W,
                                                  int tilesX, int tilesY, int *__restrict__ exact_flags, uint8_t *__restrict__ in_bad,
                                                  uint8_t *__restrict__ mark_dirty)
{
    const int r = (int)(i / W), c = (int)(i % W);
    if (i >= n || !ws_active(active, b, r, c, tilesX, tilesY)) return;
    const int64_t fbase = (int64_t)b * n, g = fbase + i;
    if (MODE == UF_REPAIR && in_bad) in_bad[g] = 0;
    int x = parent[g];
    if (x < 0) return;
    const int *par = parent + fbase;
    bool broken = (x & (UF_NS - 1)) >= (int)n;
    if (!broken) x = vwalk_root(par, x, (int)n, broken);
    if (broken) {  // a walk left its fence (vwalk_ok): the frame is recomputed by the exact flood / reported
        exact_flags[b] = 1;
        return;
    }
    if (x >= UF_NS) return;  // no labelled pixel in the component
    const int64_t groot = fbase + x;
    if (MODE == UF_REPAIR) {
        if (bad[groot]) {
            F[g] = mask[g] ? markers[g] : 0;
            // the tile joins the second level's active set and carries a mark for its first round (two arrays: the set stays,
            // the marks are taken down as tiles are visited)
            if (mark_active) mark_active[((int64_t)b * tilesY + r / WS_T) * tilesX + c / WS_T] = 1;
            if (mark_dirty) mark_dirty[((int64_t)b * tilesY + r / WS_T) * tilesX + c / WS_T] = 1;
            if (in_bad) in_bad[g] = 1;
        }
        return;
    }
    const int f = F[g];
    if (MODE == UF_ASSIGN) {
        if (f == 0 && !bad[groot]) F[g] = F[groot];
        return;
    }
    const int lab = F[groot];  // roots are labelled pixels, never written by this pass
    if (f == 0) {
        if (MODE == UF_OPTIMISTIC) F[g] = lab;
    } else if (f != lab) {
        bad[groot] = 1;
        if (tie_flags[b] == 0) tie_flags[b] = 1;
    }
}

template <int MODE>
__global__ void __launch_bounds__(256) ws_uf_label_kernel(const int *__restrict__ frame_list, const int *__restrict__ parent, int *__restrict__ F,
                                                           const uint8_t *__restrict__ active, uint8_t *__restrict__ bad,
                                                           const int *__restrict__ markers, const uint8_t *__restrict__ mask,
                                                           int *__restrict__ tie_flags, uint8_t *__restrict__ mark_active, int64_t n,
                                                           int W, int tilesX, int tilesY, int *__restrict__ exact_flags,
                                                           uint8_t *__restrict__ in_bad = nullptr, uint8_t *__restrict__ mark_dirty = nullptr)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    ws_for_frames(frame_list, blockIdx.y, gridDim.y, [&](const int b) {
        ws_uf_label_pixel<MODE>(b, i, parent, F, active, bad, markers, mask, tie_flags, mark_active, n, W, tilesX, tilesY, exact_flags,
                                in_bad, mark_dirty);
    });
}

// the same pass over the listed 64 x 64 tiles only (second level: UF_DETECT, UF_ASSIGN)
template <int MODE>
__global__ void __launch_bounds__(256) ws_uf_label_tiles_kernel(WsTileList tiles, const int *__restrict__ parent, int *__restrict__ F,
                                                                 const uint8_t *__restrict__ active, uint8_t *__restrict__ bad,
                                                                 const int *__restrict__ markers, const uint8_t *__restrict__ mask,
                                                                 int *__restrict__ tie_flags, int H, int W, int tilesX, int tilesY,
                                                                 int *__restrict__ exact_flags)
{
    ws_for_tiles(tiles, [&](const int b, const int tx, const int ty) {
        const int c = tx * WS_T + (threadIdx.x & 63);
        for (int rr = threadIdx.x >> 6; rr < WS_T; rr += 4) {
            const int r = ty * WS_T + rr;
            if (r < H && c < W)
                ws_uf_label_pixel<MODE>(b, (int64_t)r * W + c, parent, F, active, bad, markers, mask, tie_flags, nullptr, (int64_t)H * W, W,
                                        tilesX, tilesY, exact_flags, nullptr, nullptr);
        }
    });
}

// UF_OPTIMISTIC over whole frames with W % 4 == 0: four pixels per lane (16-byte parent / label accesses; neighbours
// mostly share their root, so the walk is repeated only when the parent entry changes).  The label image is WRITTEN
// but not read: a reachable pixel whose minimum-neighbour mask is 0 is a seed (ws_uf_tile_frame; every other reachable
// pixel got its level from a neighbour), only seeds carry a label before this pass, and only their four bytes are
// fetched -- 1 byte of mask per pixel instead of 4 of labels.
// (F carries no __restrict__: seed and root words are READ while the int4 stores of other lanes rewrite them with the
// value they already hold -- seeds are the only labelled pixels before this pass and the pass never changes one)
// Like ccl_relabel_quads_kernel a lane owns LABEL4_Q quads (LABEL4_Q * 1024 pixels per block, quad q of a lane 1024
// pixels further on) and walks their chains -- parent entry, root, the root's label -- in lockstep: a lane with one
// chain waits a memory latency per step (89 % of this kernel's wave cycles were waits).  A quad's chain belongs to its
// first reachable pixel; a pixel with another parent entry walks on its own afterwards.
#ifndef PCSEG_LABEL4_Q
#define PCSEG_LABEL4_Q 3  // quads a lane carries (3 x 2 lockstep chains): 78 scalar / 53 vector registers, 194 us; 4: 94 / 69, 216 us
#endif
constexpr int LABEL4_Q = PCSEG_LABEL4_Q;

// NCH chains in lockstep: root[q] (-1 = none) walks to its root, lab[q] becomes the root's label (0: no seed)
// The fence of these chains costs TWO vector instructions a step and no compare: every load goes to a CLAMPED index (never
// outside the frame, whatever the entry says) and an entry above its node's virtual index is cut down to the node itself by an
// unsigned min -- the chain then simply ends there (strictly decreasing: never a cycle); `root - next < 0` accumulates as a
// sign bit and one compare at the end raises the frame's flag.  (vwalk_ok written out per chain -- as `if`s or as selects fed
// by compares -- parks sixteen lane masks in scalar register pairs: the kernel went from 96 to 106 scalar registers, i.e.
// from seven to six workgroups per CU, and this pass is nothing but memory latency: 204 -> 265 us.  What the compiler's
// occupancy figure does not show: residency of 256-thread workgroups is min(8, 800 / (ceil(sgpr / 16) * 16 + 16)) --
// MI355X_MICROARCH.md -- so `__launch_bounds__(256, 8)` below is what makes it budget scalar registers too: 78, eight
// workgroups per CU.)
template <int NCH>
__device__ __forceinline__ void label4_chains(const int *par, const int *F, int64_t fbase, int n, int (&root)[NCH], int (&lab)[NCH], bool &bad)
{
    const int n1 = n - 1;
    int viol = 0;
    bool more = true;
    while (more) {
        int nx[NCH];
#pragma unroll
        for (int q = 0; q < NCH; ++q) nx[q] = root[q] >= 0 ? par[min(root[q] & (UF_NS - 1), n1)] : -1;
        int differ = 0;
#pragma unroll
        for (int q = 0; q < NCH; ++q) {
            viol |= root[q] - nx[q];                               // sign bit: an entry above its node (both < 2^31; no chain: 0)
            nx[q] = (int)min((unsigned)nx[q], (unsigned)root[q]);  // ... which ends the chain where it stands
            differ |= nx[q] ^ root[q];
            root[q] = nx[q];
        }
        more = differ != 0;
    }
    bad = bad || viol < 0;
    const int seeds_end = min(UF_NS, n);  // (a root is a seed pixel of the frame -- anything else is not to be trusted)
#pragma unroll
    for (int q = 0; q < NCH; ++q)
        lab[q] = (root[q] >= 0 && root[q] < seeds_end) ? F[fbase + root[q]] : 0;  // roots are labelled pixels, never changed by this pass
}

#ifndef PCSEG_LABEL4_OCC
// workgroups per CU the register allocator is asked for.  With four quads a lane: 7 = 94 scalar / 69 vector registers, 217 us a
// launch; 8 = 78 / 64 with 28 bytes of scratch, 254 us; without the argument 106 / 67 = six workgroups, 265 us.  With THREE quads
// a lane eight workgroups fit without scratch: 194 us (profiles/r04/ab_logs/r4i_*, r4j_*)
#define PCSEG_LABEL4_OCC 8
#endif
__global__ void __launch_bounds__(256, PCSEG_LABEL4_OCC) ws_uf_label4_kernel(const int *__restrict__ parent, const uint8_t *__restrict__ minmask,
                                                            int *F, uint8_t *__restrict__ bad, int *__restrict__ tie_flags,
                                                            int64_t n, int *__restrict__ exact_flags)
{
    const int64_t i0 = (int64_t)blockIdx.x * (1024 * LABEL4_Q) + threadIdx.x * 4;
    const int b = blockIdx.y;
    const int64_t fbase = (int64_t)b * n;
    const int *par = parent + fbase;
    int4 p4[LABEL4_Q];
    unsigned m4[LABEL4_Q];
#pragma unroll
    for (int q = 0; q < LABEL4_Q; ++q) {
        const int64_t i = i0 + q * 1024;
        p4[q] = i < n ? *reinterpret_cast<const int4 *>(par + i) : make_int4(-1, -1, -1, -1);
        m4[q] = i < n ? *reinterpret_cast<const unsigned *>(minmask + fbase + i) : 0u;
    }
    // the labels of the quads that hold a seed (reachable pixel with an empty minimum-neighbour mask), as one batch
    unsigned seeds[LABEL4_Q];
    int4 f4[LABEL4_Q];
#pragma unroll
    for (int q = 0; q < LABEL4_Q; ++q) {
        const int pv[4] = {p4[q].x, p4[q].y, p4[q].z, p4[q].w};
        seeds[q] = 0;
#pragma unroll
        for (int j = 0; j < 4; ++j)
            if (pv[j] >= 0 && ((m4[q] >> (8 * j)) & 255u) == 0) seeds[q] |= 1u << j;
        f4[q] = make_int4(0, 0, 0, 0);
    }
#pragma unroll
    for (int q = 0; q < LABEL4_Q; ++q)
        if (seeds[q]) f4[q] = *reinterpret_cast<const int4 *>(F + fbase + i0 + q * 1024);
    // chains 0..3: the quads' first entries (only the first lane of each run of lanes with the same entry walks; the others
    // read its answer across lanes); chains 4..7: the second entry of quads that straddle two components (walked pixel by
    // pixel these were most of the pass: see ccl_relabel_quads_kernel).  All eight advance in lockstep.
    int lead[LABEL4_Q], lead2[LABEL4_Q], rt[2 * LABEL4_Q], lb[2 * LABEL4_Q], head_lane[LABEL4_Q];
    const int lane = lane_id();
#pragma unroll
    for (int q = 0; q < LABEL4_Q; ++q) {
        const int pv[4] = {p4[q].x, p4[q].y, p4[q].z, p4[q].w};
        lead[q] = pv[0] >= 0 ? pv[0] : (pv[1] >= 0 ? pv[1] : (pv[2] >= 0 ? pv[2] : pv[3]));
        const int left = __shfl_up(lead[q], 1);
        const bool head = lane == 0 || lead[q] != left;
        const unsigned long long heads = __ballot(head);
        head_lane[q] = 63 - __clzll((long long)(heads & (~0ull >> (63 - lane))));
        rt[q] = head ? lead[q] : -1;
        lead2[q] = -1;
#pragma unroll
        for (int j = 1; j < 4; ++j)
            if (pv[j] >= 0 && pv[j] != lead[q] && lead2[q] < 0) lead2[q] = pv[j];
        rt[LABEL4_Q + q] = lead2[q];
    }
    bool broken = false;
    label4_chains(par, F, fbase, (int)n, rt, lb, broken);
    int root[LABEL4_Q], lab[LABEL4_Q], root2[LABEL4_Q], lab2[LABEL4_Q];
#pragma unroll
    for (int q = 0; q < LABEL4_Q; ++q) {
        root[q] = __shfl(rt[q], head_lane[q]);
        lab[q] = __shfl(lb[q], head_lane[q]);
        root2[q] = rt[LABEL4_Q + q];
        lab2[q] = lb[LABEL4_Q + q];
    }
#pragma unroll
    for (int q = 0; q < LABEL4_Q; ++q) {
        const int64_t i = i0 + q * 1024;
        if (i >= n) continue;
        const int pv[4] = {p4[q].x, p4[q].y, p4[q].z, p4[q].w};
        const int sv[4] = {f4[q].x, f4[q].y, f4[q].z, f4[q].w};
        int fv[4] = {0, 0, 0, 0};  // unreachable pixels and pixels of components without a seed stay unlabelled
        bool wrote = false;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            int x = pv[j];
            if (x < 0) continue;
            const bool seed = (seeds[q] >> j) & 1u;
            if (seed) fv[j] = sv[j];
            int x_root, x_lab;
            if (x == lead[q]) { x_root = root[q]; x_lab = lab[q]; }
            else if (x == lead2[q]) { x_root = root2[q]; x_lab = lab2[q]; }
            else {  // (third: a quad with three different entries)
                // fenced like the chains: clamped index, an entry above its node ends the walk
                int t;
                while ((t = (int)min((unsigned)par[min(x & (UF_NS - 1), (int)n - 1)], (unsigned)x)) != x) x = t;
                x_root = x;
                x_lab = (x >= 0 && x < min(UF_NS, (int)n)) ? F[fbase + x] : 0;
            }
            if (x_root < 0 || x_root >= UF_NS) continue;  // no labelled pixel in the component (or a chain that broke its fence)
            if (!seed) {
                fv[j] = x_lab;
                wrote = true;
            } else if (fv[j] != x_lab) {
                bad[fbase + x_root] = 1;
                if (tie_flags[b] == 0) tie_flags[b] = 1;
            }
        }
        if (wrote) *reinterpret_cast<int4 *>(F + fbase + i) = make_int4(fv[0], fv[1], fv[2], fv[3]);
    }
    if (broken) exact_flags[b] = 1;  // a walk left its fence (vwalk_ok): recomputed by the exact flood / reported
}

// (3) proof check: every neighbour whose key equals the minimum neighbour key carries the pixel's label
template <typename KeyT>
__global__ void __launch_bounds__(256) ws_check_kernel(const KeyT *__restrict__ K, const int *__restrict__ F,
                                                        const int *__restrict__ markers, const uint8_t *__restrict__ mask,
                                                        const int *__restrict__ frame_flags, int *__restrict__ tie_flags,
                                                        int H, int W)
{
    const int c = blockIdx.x * 64 + (threadIdx.x & 63);
    const int r = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (r >= H || c >= W) return;
    const int b = blockIdx.z;
    if (frame_flags && frame_flags[b] == 0) return;
    const KeyT KINF = ~(KeyT)0;
    const int64_t fbase = (int64_t)b * H * W;
    const int64_t i = fbase + rowoff(r, W) + c;
    const KeyT l = K[i];
    if ((unsigned)(l >> (8 * sizeof(KeyT) - 32)) == WS_INF) return;  // outside the mask or unreachable: stays 0
    if (mask[i] != 0 && markers[i] != 0) return;                     // seed: keeps its marker
    const KeyT lu = r > 0 ? K[i - W] : KINF, ld = r + 1 < H ? K[i + W] : KINF;
    const KeyT ll = c > 0 ? K[i - 1] : KINF, lr = c + 1 < W ? K[i + 1] : KINF;
    const KeyT m = min(min(lu, ld), min(ll, lr));
    const int f = F[i];
    bool ok = f != 0;
    if (lu == m) ok = ok && F[i - W] == f;
    if (ll == m) ok = ok && F[i - 1] == f;
    if (lr == m) ok = ok && F[i + 1] == f;
    if (ld == m) ok = ok && F[i + W] == f;
    if (!ok && tie_flags[b] == 0) tie_flags[b] = 1;
}

// ---- second-level order for frames whose first check failed ---------------------------------------------------
// Inside one level v = L the reference pops, in this order: every "primary entry" (a pixel of value v that was
// pushed from a lower level, or a seed of value v) by age, each immediately followed by the whole connected lake of
// lower-valued pixels it opens; then the value-v pixels pushed during the level ("secondary").  The age of a primary
// entry is the pop time of its parent, whose level is K2 = min{L(n) : n neighbour, L(n) < v}; seeds have age 0.
// Hence T is refined by (L, K2) with K2(seed) = 0, K2(entry) = that minimum, K2(secondary) = WS_SECONDARY and
// K2(lake pixel) = min K2 over the neighbours of the same level (the earliest entry floods the whole lake).
constexpr unsigned WS_SECONDARY = 0xFFFFFFFEu;

__global__ void __launch_bounds__(256) ws_k2_init_kernel(WsTileList tiles, const unsigned *__restrict__ val, const unsigned *__restrict__ L,
                                                          const int *__restrict__ markers, const uint8_t *__restrict__ mask,
                                                          unsigned *__restrict__ K2, int H, int W)
{
    ws_for_tiles(tiles, [&](const int b, const int tx, const int ty) {
    const int c = tx * WS_T + (threadIdx.x & 63);
    for (int rr = threadIdx.x >> 6; rr < WS_T; rr += 4) {
        const int r = ty * WS_T + rr;
        if (r >= H || c >= W) continue;
        const int64_t i = (int64_t)b * H * W + (int64_t)r * W + c;
        const unsigned l = L[i];
        unsigned k = WS_INF;
        if (l != WS_INF) {
            if (mask[i] != 0 && markers[i] != 0) k = 0;
            else if (val[i] == l) {
                const unsigned lu = r > 0 ? L[i - W] : WS_INF, ld = r + 1 < H ? L[i + W] : WS_INF;
                const unsigned ll = c > 0 ? L[i - 1] : WS_INF, lr = c + 1 < W ? L[i + 1] : WS_INF;
                const unsigned m = min(min(lu, ld), min(ll, lr));
                k = m < l ? max(m, 1u) : WS_SECONDARY;
            }
        }
        K2[i] = k;
    }
    });
}

// threads per tile of the second level: its tile pass is a handful of dependent LDS phases -- with 1024 threads a thread owns
// 4 cells of the tile (256 threads, 16 cells each: 58 us a grid round against 32; profiles/r04/ab_logs/r4d_*, r4e_*)
constexpr int K2T = 1024;
struct WsK2Lds {
    unsigned sL[WS_N];
    unsigned sK[WS_N];
    uint8_t sLake[WS_N];
    int par[WS_N];
};

// one tile of one second-level round (block-uniform control flow, like ws_relax_tile)
// `in_bad` (may be null: every lake): the per-pixel mark UF_REPAIR leaves.  K2 is only ever COMPARED between the
// minimum-level neighbours of a pixel that is still unlabelled, i.e. inside one unresolved first-level component; a lake
// (connected pixels of one level below it) and the entries that open it are linked by minimum-level links, so they lie
// in ONE component, and the K2 of a component's pixels depends on nothing outside it.  An active 64 x 64 tile of a noise
// field holds hundreds of lakes, the slowest of which used to set the iteration count of every visit (60 us per grid
// round for a few dozen tiles); the unresolved components hold a handful.
__device__ __forceinline__ void ws_k2_relax_tile(WsK2Lds &lds, const unsigned *__restrict__ val, const unsigned *__restrict__ L,
                                                 unsigned *__restrict__ K2, const uint8_t *__restrict__ active,
                                                 uint8_t *__restrict__ dirty_in, uint8_t *__restrict__ dirty_out, int H, int W,
                                                 int tilesX, int tilesY, int tx, int ty, int b, const uint8_t *__restrict__ in_bad)
{
    unsigned *sL = lds.sL, *sK = lds.sK;
    uint8_t *sLake = lds.sLake;
    if (!active[((int64_t)b * tilesY + ty) * tilesX + tx]) return;  // K2 is only defined inside the active tiles
    {
        uint8_t *mark = dirty_in + ((int64_t)b * tilesY + ty) * tilesX + tx;
        if (!*mark) return;
        __syncthreads();
        if (threadIdx.x == 0) *mark = 0;  // see ws_relax_tile
    }
    const int r0 = ty * WS_T, c0 = tx * WS_T;
    const int64_t fbase = (int64_t)b * H * W;
    {
        // the three arrays of the tile + halo as ONE batch of loads per thread (clamped addresses, no branch around a
        // load; see ws_relax_tile): a loop of dependent round trips here cost more than the sweeps
        constexpr int TRIPS = (WS_S * WS_S + K2T - 1) / K2T;
        unsigned lv[TRIPS], kv[TRIPS], vv[TRIPS];
        uint8_t bv[TRIPS];
#pragma unroll
        for (int t = 0; t < TRIPS; ++t) {
            const int i = min((int)threadIdx.x + K2T * t, WS_S * WS_S - 1);
            const int r = r0 + i / WS_S - 1, c = c0 + i % WS_S - 1;
            const int64_t p = fbase + rowoff(min(max(r, 0), H - 1), W) + min(max(c, 0), W - 1);
            lv[t] = L[p];
            kv[t] = K2[p];
            vv[t] = val[p];
            bv[t] = in_bad ? in_bad[p] : (uint8_t)1;
        }
#pragma unroll
        for (int t = 0; t < TRIPS; ++t) {
            const int i = (int)threadIdx.x + K2T * t;
            if (i < WS_S * WS_S) {
                const int lr = i / WS_S, lc = i % WS_S;
                const int r = r0 + lr - 1, c = c0 + lc - 1;
                const bool in = r >= 0 && r < H && c >= 0 && c < W;
                sL[lr * WS_P + lc] = in ? lv[t] : WS_INF;
                sK[lr * WS_P + lc] = in ? kv[t] : WS_INF;
                sLake[lr * WS_P + lc] = in && lv[t] != WS_INF && vv[t] < lv[t] && bv[t] != 0;
            }
        }
    }
    __syncthreads();
    // The fixed point in ONE pass (round 4): an interior lake cell
    // takes the smallest key among its same-level 4-neighbours, and lake cells of one level that touch take it from each
    // other -- so every connected set of interior lake cells ends with the minimum over its own keys and the keys of the
    // same-level cells around it (entries, which are not lake cells and keep their key, and halo cells, which belong to the
    // neighbour tiles).  That is a union-find over the tile's lake cells plus one atomic min per cell at its root: a handful
    // of LDS passes whatever the shape of the lake.  (Rounds 2-3 ran directional line sweeps here, one wave per direction: an
    // iteration per turn of a winding lake -- tens of iterations of pure LDS latency, 127 us for the first grid round of the
    // benchmark batch's 31 active tiles; four grid rounds 196 us against 128 now, profiles/r04/relax_sweep_variants_and_k2.diff.)
    int *par = lds.par;
    constexpr int CELLS = WS_T * WS_T / K2T;  // interior cells per thread
    unsigned k_old[CELLS];
#pragma unroll
    for (int q = 0; q < CELLS; ++q) {
        const int t = threadIdx.x + K2T * q, i = (t / WS_T + 1) * WS_P + t % WS_T + 1;
        par[i] = sLake[i] ? i : -1;
        k_old[q] = sK[i];
    }
    __syncthreads();
    // links between interior lake cells of one level: right and down (the other two are some cell's right / down)
#pragma unroll
    for (int q = 0; q < CELLS; ++q) {
        const int t = threadIdx.x + K2T * q, lr = t / WS_T + 1, lc = t % WS_T + 1, i = lr * WS_P + lc;
        if (!sLake[i]) continue;
        const unsigned l = sL[i];
        if (lc < WS_T && sLake[i + 1] && sL[i + 1] == l) unite_lds_halving(par, i, i + 1);
        if (lr < WS_T && sLake[i + WS_P] && sL[i + WS_P] == l) unite_lds_halving(par, i, i + WS_P);
    }
    __syncthreads();
    // every lake cell brings its own key and the keys of the same-level cells around it that are not members (entries,
    // halo cells) to its root
#pragma unroll
    for (int q = 0; q < CELLS; ++q) {
        const int t = threadIdx.x + K2T * q, lr = t / WS_T + 1, lc = t % WS_T + 1, i = lr * WS_P + lc;
        if (!sLake[i]) continue;
        const unsigned l = sL[i];
        unsigned m = k_old[q];
        const int nb[4] = {i - WS_P, i - 1, i + 1, i + WS_P};
        const bool interior[4] = {lr > 1, lc > 1, lc < WS_T, lr < WS_T};
#pragma unroll
        for (int d = 0; d < 4; ++d)
            if (sL[nb[d]] == l && !(interior[d] && sLake[nb[d]])) m = min(m, sK[nb[d]]);  // (members meet at the root anyway)
        const int root = find_lds_halving(par, i);
        if (root != i || m != k_old[q]) atomicMin(&sK[root], m);
    }
    __syncthreads();
    bool changed = false;
    unsigned k_new[CELLS];
#pragma unroll
    for (int q = 0; q < CELLS; ++q) {
        const int t = threadIdx.x + K2T * q, i = (t / WS_T + 1) * WS_P + t % WS_T + 1;
        k_new[q] = sLake[i] ? sK[find_lds_halving(par, i)] : k_old[q];
        changed = changed || k_new[q] != k_old[q];
    }
    __syncthreads();  // every root has been read before a member overwrites its own cell (a root's own write is the value it holds)
#pragma unroll
    for (int q = 0; q < CELLS; ++q) {
        const int t = threadIdx.x + K2T * q, i = (t / WS_T + 1) * WS_P + t % WS_T + 1;
        if (k_new[q] != k_old[q]) sK[i] = k_new[q];
    }
    const bool changed_any = __syncthreads_or(changed);
    if (!changed_any) return;
    ws_mark_changed_edges(sK, (const unsigned *)K2 + fbase, dirty_out, b, tx, ty, tilesX, tilesY, r0, c0, H, W);
    __syncthreads();
    ws_store_tile(sK, K2 + fbase, r0, c0, H, W);
}

__global__ void __launch_bounds__(K2T) ws_k2_relax_kernel(WsTileList tiles, const unsigned *__restrict__ val, const unsigned *__restrict__ L,
                                                           unsigned *__restrict__ K2, const uint8_t *__restrict__ active,
                                                           uint8_t *__restrict__ dirty_in, uint8_t *__restrict__ dirty_out,
                                                           int H, int W, int tilesX, int tilesY, const uint8_t *__restrict__ in_bad)
{
    __shared__ WsK2Lds lds;
    ws_for_tiles(tiles, [&](const int b, const int tx, const int ty) {
        ws_k2_relax_tile(lds, val, L, K2, active, dirty_in, dirty_out, H, W, tilesX, tilesY, tx, ty, b, in_bad);
        __syncthreads();  // the next listed tile reuses the tile arrays
    });
}

// the rest of the second-level fixed point after its grid rounds, one block per flagged frame (see ws_relax_tail_kernel)
__global__ void __launch_bounds__(K2T) ws_k2_relax_tail_kernel(const int *__restrict__ frame_list, const unsigned *__restrict__ val,
                                                                const unsigned *__restrict__ L, unsigned *__restrict__ K2,
                                                                const uint8_t *__restrict__ active, uint8_t *__restrict__ dirtyA,
                                                                uint8_t *__restrict__ dirtyB, int *__restrict__ not_converged,
                                                                int *__restrict__ exact_flags, int H, int W, int tilesX, int tilesY,
                                                                int max_rounds, const uint8_t *__restrict__ in_bad)
{
    __shared__ WsK2Lds lds;
    __shared__ int tail_list[WS_TAIL_LIST];
    __shared__ int tail_count;
    ws_for_frames(frame_list, blockIdx.x, gridDim.x, [&](const int b) {
    uint8_t *din = dirtyA, *dout = dirtyB;
    const int ntiles = tilesX * tilesY;
    for (int round = 0;; ++round) {
        // (a mark on a tile outside the active set is never taken down -- such tiles are not visited -- and is no work)
        const uint8_t *marks = din + (int64_t)b * ntiles, *act = active + (int64_t)b * ntiles;
        // the round's work list: marked active tiles, gathered in parallel (see ws_relax_tail_kernel)
        __syncthreads();
        if (threadIdx.x == 0) tail_count = 0;
        __syncthreads();
        for (int t = threadIdx.x; t < ntiles; t += K2T)
            if (marks[t] != 0 && act[t] != 0) {
                const int k = atomicAdd(&tail_count, 1);
                if (k < WS_TAIL_LIST) tail_list[k] = t;
            }
        __syncthreads();
        const int marked = tail_count;
        if (marked == 0) return;
        if (round >= max_rounds) {
            if (threadIdx.x == 0) { *not_converged = 1; exact_flags[b] = 1; }  // see ws_relax_tail_kernel
            return;
        }
        const int walk = marked <= WS_TAIL_LIST ? marked : ntiles;
        for (int k = 0; k < walk; ++k) {
            const int t = marked <= WS_TAIL_LIST ? tail_list[k] : k;
            ws_k2_relax_tile(lds, val, L, K2, active, din, dout, H, W, tilesX, tilesY, t % tilesX, t / tilesX, b, in_bad);
            __syncthreads();
        }
        uint8_t *tmp = din; din = dout; dout = tmp;
    }
    });
}

// K64 = (L << 32) | K2 inside the active tiles.  A pixel just outside an active tile can sit in the halo of its union-find
// tiles, where it must never look like a minimum-key neighbour (it is not in the component, so its L is larger than the
// minimum anyway): the one-pixel ring around a listed tile gets (L, worst K2) wherever it does not belong to another active
// tile (which packs its own pixels).  Nothing further out is ever read at this level.
__global__ void __launch_bounds__(256) ws_pack_kernel(WsTileList tiles, const unsigned *__restrict__ L, const unsigned *__restrict__ K2,
                                                       const uint8_t *__restrict__ active, unsigned long long *__restrict__ K64, int H, int W,
                                                       int tilesX, int tilesY)
{
    ws_for_tiles(tiles, [&](const int b, const int tx, const int ty) {
        const int64_t fbase = (int64_t)b * H * W;
        const int c = tx * WS_T + (threadIdx.x & 63);
        for (int rr = threadIdx.x >> 6; rr < WS_T; rr += 4) {
            const int r = ty * WS_T + rr;
            if (r < H && c < W) {
                const int64_t g = fbase + (int64_t)r * W + c;
                K64[g] = ((unsigned long long)L[g] << 32) | K2[g];
            }
        }
        // ring: thread t < 66 -> row above (t - 1 = column offset), 66..131 -> row below, 132..195 -> left column, 196..259 -> right
        for (int t = threadIdx.x; t < 2 * (WS_T + 2) + 2 * WS_T; t += 256) {
            int r, cc;
            if (t < WS_T + 2) { r = ty * WS_T - 1; cc = tx * WS_T - 1 + t; }
            else if (t < 2 * (WS_T + 2)) { r = ty * WS_T + WS_T; cc = tx * WS_T - 1 + (t - (WS_T + 2)); }
            else if (t < 2 * (WS_T + 2) + WS_T) { r = ty * WS_T + (t - 2 * (WS_T + 2)); cc = tx * WS_T - 1; }
            else { r = ty * WS_T + (t - 2 * (WS_T + 2) - WS_T); cc = tx * WS_T + WS_T; }
            if (r < 0 || r >= H || cc < 0 || cc >= W || ws_active(active, b, r, cc, tilesX, tilesY)) continue;
            const int64_t g = fbase + (int64_t)r * W + cc;
            K64[g] = ((unsigned long long)L[g] << 32) | 0xFFFFFFFFull;
        }
    });
}

// verification builds of the second level run on whole flagged frames: every tile of such a frame becomes active and
// its labels are reset to the seeds
__global__ void __launch_bounds__(256) ws_activate_frames_kernel(const int *__restrict__ frame_list, const int *__restrict__ frame_flags, uint8_t *__restrict__ active,
                                                                  uint8_t *__restrict__ dirty, const int *__restrict__ markers,
                                                                  const uint8_t *__restrict__ mask, int *__restrict__ out, int64_t n, int W,
                                                                  int tilesX, int tilesY)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    ws_for_frames(frame_list, blockIdx.y, gridDim.y, [&](const int b) {
    if (i >= n || frame_flags[b] == 0) return;
    active[((int64_t)b * tilesY + (int)(i / W) / WS_T) * tilesX + (int)(i % W) / WS_T] = 1;
    dirty[((int64_t)b * tilesY + (int)(i / W) / WS_T) * tilesX + (int)(i % W) / WS_T] = 1;
    out[(int64_t)b * n + i] = mask[(int64_t)b * n + i] ? markers[(int64_t)b * n + i] : 0;
    });
}

// flagged frames, in order, as a device-side list; its length sits in front of it (frame_list[-1], see ws_for_frames)
__global__ void ws_list_flagged_kernel(const int *__restrict__ flags, int B, int *__restrict__ frame_list)
{
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    int n = 0;
    for (int b = 0; b < B; ++b)
        if (flags[b]) frame_list[n++] = b;
    frame_list[-1] = n;
}

// the call's tile counters (WS_CNT0 ..) go into a per-device running total, read by pcseg_watershed_counters
__global__ void ws_accumulate_kernel(const int *__restrict__ changed, unsigned long long *__restrict__ total)
{
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    unsigned long long s = 0;
    for (int i = 0; i < 16; ++i) s += (unsigned long long)changed[WS_CNT0 + WS_CNT_STRIDE * i];
    atomicAdd(total, s);
}

__global__ void ws_set_flags_kernel(int *flags, int B, int v)
{
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < B) flags[i] = v;
}

// ---- exact emulation of the reference's heap-driven flood for flagged frames
//
// The reference's queue is a binary heap ordered by (value, age) in which all seeds carry age 0: seeds of equal value
// leave it in an order only the heap's own sift rules define, so the emulation has to be the literal binary heap, not
// just any priority queue.  A lone lane would pay one memory round trip and a few dozen instructions per heap level
// (one wave issues an instruction every four cycles at best); instead one WAVE runs one frame and works on five
// levels at a time:
//   * heap slots 1..EX_LDS-1 (the top levels) live in LDS, deeper slots in the workspace;
//   * sift-down: lane r of 1..31 owns relative node r of the 5-level subtree under the current hole, loads both of
//     its children and decides -- against the wave-uniform key being placed -- which child would move up.  The path
//     is then five v_readlane hops, and all lanes on it write their slot at once;
//   * sift-up: lane s loads ancestor c >> s, one ballot gives the number of levels the new key climbs;
//   * the four neighbours of the popped pixel are examined by four lanes while the sift-down is in flight.
// Comparisons and moves are those of the sequential heappush / heappop, so the pop order is identical.
// Heap slots kept in LDS: 2^13 (levels 0..12, 96 KB: one frame per CU, two of the three sift-down windows of a 1024^2
// frame stay in LDS) when the call has at most one frame per CU, or 2^7 (levels 0..6, 1.5 KB: only the first window) for
// larger calls, where throughput comes from the number of frames in flight -- the flood of a frame is sequential by
// definition, so a batch of N frames uses N waves whatever else is done, and 16+ of them fit a CU once the LDS share
// is small.  Both sizes keep every six-level window entirely in LDS or entirely in the workspace.
#ifndef PCSEG_EX_SMALL_LDS
#define PCSEG_EX_SMALL_LDS 128
#endif
#ifndef PCSEG_EX_SMALL_D
#define PCSEG_EX_SMALL_D 6
#endif
constexpr int EX_LDS_BIG = 8192, EX_LDS_SMALL = PCSEG_EX_SMALL_LDS, EX_D_BIG = 6, EX_D_SMALL = PCSEG_EX_SMALL_D;
static_assert((EX_LDS_SMALL & (EX_LDS_SMALL - 1)) == 0 && EX_LDS_SMALL >= (2 << EX_D_SMALL), "the first window must fit the LDS share");

// (the LDS halves are typed by address space: with four generic pointers the compiler folds `d < lds ? lk[d] : gk[d]` into
// ONE flat access through a selected pointer -- 18 flat loads / stores per pop and push, each of which waits on both the
// LDS and the memory counter)
typedef __attribute__((address_space(3))) unsigned long long ex_lds_key_t;
typedef __attribute__((address_space(3))) unsigned ex_lds_idx_t;

struct ExactHeap {
    ex_lds_key_t *lk;  // LDS keys   (slot d at lk[d], slot 0 unused)
    ex_lds_idx_t *lx;  // LDS pixel indices
    unsigned long long *gk;  // workspace keys (slot d at gk[d])
    unsigned *gx;
    int lds;  // slots below this index live in LDS
    int items;
    unsigned long long tail_key;  // content of slot `items` (the entry the next heappop re-inserts), kept in registers:
    unsigned tail_idx;            // known after a push, fetched ahead of time at the end of a pop

    __device__ __forceinline__ unsigned long long key_at(int d) const
    {
        return d < lds ? lk[d] : gk[d];
    }
    __device__ __forceinline__ unsigned idx_at(int d) const
    {
        return d < lds ? lx[d] : gx[d];
    }
    __device__ __forceinline__ void put(int d, unsigned long long k, unsigned x) const
    {
        if (d < lds) {
            lk[d] = k;
            lx[d] = x;
        } else {
            gk[d] = k;
            gx[d] = x;
        }
    }
};

__device__ __forceinline__ unsigned lane_u32(unsigned v, int lane)
{
    return (unsigned)__builtin_amdgcn_readlane((int)v, __builtin_amdgcn_readfirstlane(lane));
}
__device__ __forceinline__ unsigned long long lane_u64(unsigned long long v, int lane)
{
    const int l = __builtin_amdgcn_readfirstlane(lane);
    unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)v, l);
    unsigned hi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(v >> 32), l);
    return ((unsigned long long)hi << 32) | lo;
}

// slot of relative position `rel` (1 = the node itself, children of r are 2r and 2r+1) in the subtree rooted at `j`
__device__ __forceinline__ long long ex_slot(int j, int rel)
{
    const int lv = 31 - __clz(rel);
    return ((long long)j << lv) + (rel - (1 << lv));
}

// heappush: the new entry starts in slot c = ++items and climbs while it is smaller than its parent
__device__ __forceinline__ void ex_push(ExactHeap &h, unsigned long long key, unsigned idx)
{
    asm volatile("" ::: "memory");  // lanes hand heap slots to each other: no value may be carried across in registers
    const int t = threadIdx.x;
    const int c = ++h.items;
    const int a = t < 31 ? (c >> t) : 0;  // lane s >= 1 owns ancestor c >> s
    unsigned long long ak = 0;            // 0: never above the key -> the climb stops there (also for absent ancestors)
    unsigned ax = 0;
    if (t >= 1 && a >= 1) {
        ak = h.key_at(a);
        ax = h.idx_at(a);
    }
    const unsigned long long climbs = __ballot(t >= 1 && a >= 1 && key < ak) >> 1;
    const int m = __ffsll((long long)~climbs) - 1;  // consecutive ancestors above the key
    if (t >= 1 && t <= m) h.put(c >> (t - 1), ak, ax);
    if (t == 0) h.put(c >> m, key, idx);
    // slot c now holds the new entry, or its old parent if the entry climbed
    h.tail_key = m ? lane_u64(ak, 1) : key;
    h.tail_idx = m ? lane_u32(ax, 1) : idx;
}

// heappop after the caller has read the root: the last entry is re-inserted from the root downwards.
// Lane r = 1..63 owns relative node r of the six levels under the hole and loads the pair of its children (adjacent
// slots 2d, 2d+1: one 16-byte key load, one 8-byte index load).  Windows advance by exactly six levels, so with
// EX_LDS = 2^13 a window reads either LDS only (hole on level 0 or 6) or the workspace only (hole on level 12, 18, ..).
template <bool FROM_LDS>
__device__ __forceinline__ void ex_children(const ExactHeap &h, long long d, unsigned long long &lkey, unsigned long long &rkey,
                                            unsigned &lidx, unsigned &ridx)
{
    lkey = rkey = ~0ull;
    lidx = ridx = 0;
    if (2 * d <= h.items) {
        typedef unsigned long long __attribute__((ext_vector_type(2))) key2_t;
        typedef unsigned __attribute__((ext_vector_type(2))) idx2_t;
        typedef key2_t __attribute__((aligned(8))) key2_glb_t;  // workspace slots start at an odd element
        typedef idx2_t __attribute__((aligned(4))) idx2_glb_t;
        key2_t k;
        idx2_t x;
        if (FROM_LDS) {
            k = *(const __attribute__((address_space(3))) key2_t *)(h.lk + 2 * d);
            x = *(const __attribute__((address_space(3))) idx2_t *)(h.lx + 2 * d);
        } else {
            k = *(const key2_glb_t *)(h.gk + 2 * d);
            x = *(const idx2_glb_t *)(h.gx + 2 * d);
        }
        lkey = k.x;
        lidx = x.x;
        if (2 * d + 1 <= h.items) {
            rkey = k.y;
            ridx = x.y;
        }
    }
}

// D = levels a sift-down window spans; node r of the window's 2^D - 1 belongs to lane r % 64 (D = 6: one node per lane).
// A deeper window saves memory round trips and loads more entries that are not on the path; a shallower one the
// reverse.  Quantised benchmark frames (heaps of 2^16..2^17 entries), 1024 frames per call, Mpixels/s: D = 3 (LDS 2^10)
// 271, D = 4 (2^9) 295, D = 5 (2^11) 327, D = 6 (2^7) 324, D = 8 (2^9: the whole heap in two windows) 213 -- the
// memory system is loaded by the entries as much as the waves wait for them, and 6 stays (PCSEG_EX_SMALL_D / _LDS).
template <int D>
__device__ __forceinline__ void ex_pop(ExactHeap &h)
{
    constexpr int SLOTS = ((1 << D) - 1 + 63) / 64;
    asm volatile("" ::: "memory");
    const int t = threadIdx.x;
    if (--h.items == 0) return;
    const unsigned long long key = h.tail_key;
    const unsigned idx = h.tail_idx;
    int j = 1;  // slot of the hole
    for (;;) {
        long long d[SLOTS];
        unsigned long long lkey[SLOTS], rkey[SLOTS];
        unsigned lidx[SLOTS], ridx[SLOTS];
        const bool from_lds = j < (h.lds >> D);  // block-uniform: a window is read from one place
#pragma unroll
        for (int k = 0; k < SLOTS; ++k) {
            const int rel = t + 64 * k;
            d[k] = (rel >= 1 && rel < (1 << D)) ? ex_slot(j, rel) : (long long)h.items + 1;
            if (from_lds) ex_children<true>(h, d[k], lkey[k], rkey[k], lidx[k], ridx[k]);
            else ex_children<false>(h, d[k], lkey[k], rkey[k], lidx[k], ridx[k]);
        }
        // which child would move into a node if the key being placed arrived there
        int next[SLOTS];
        unsigned long long sk[SLOTS];
        unsigned sx[SLOTS];
#pragma unroll
        for (int k = 0; k < SLOTS; ++k) {
            const int rel = t + 64 * k;
            next[k] = 0;
            sk[k] = key;
            sx[k] = 0;
            if (lkey[k] < sk[k]) { next[k] = 2 * rel; sk[k] = lkey[k]; sx[k] = lidx[k]; }
            if (rkey[k] < sk[k]) { next[k] = 2 * rel + 1; sk[k] = rkey[k]; sx[k] = ridx[k]; }
        }
        // follow the path from the hole (D hops at most) and collect the nodes on it
        unsigned long long on_path[SLOTS];
#pragma unroll
        for (int k = 0; k < SLOTS; ++k) on_path[k] = 0;
        int rel = 1;
#pragma unroll
        for (int s = 0; s < D; ++s) {
            const int kk = rel >> 6, ll = rel & 63;  // wave-uniform
            int nx = (int)lane_u32((unsigned)next[0], ll);
#pragma unroll
            for (int k = 1; k < SLOTS; ++k)
                if (kk == k) nx = (int)lane_u32((unsigned)next[k], ll);
            if (nx == 0) break;
#pragma unroll
            for (int k = 0; k < SLOTS; ++k)
                if (kk == k) on_path[k] |= 1ull << ll;
            rel = nx;
        }
#pragma unroll
        for (int k = 0; k < SLOTS; ++k)
            if ((on_path[k] >> t) & 1) h.put((int)d[k], sk[k], sx[k]);
        asm volatile("" ::: "memory");
        j = (int)ex_slot(j, rel);
        if (rel < (1 << D) || 2 * (long long)j > h.items) break;
    }
    if (t == 0) h.put(j, key, idx);
    asm volatile("" ::: "memory");
    // the entry the next heappop would re-insert; in flight while the caller looks at the popped pixel
    h.tail_key = h.key_at(h.items);
    h.tail_idx = h.idx_at(h.items);
}

// labels of the flagged frames back to their seeds (wide, one pass) before the one-wave floods start
__global__ void __launch_bounds__(256) ws_exact_init_kernel(const int *__restrict__ markers, const uint8_t *__restrict__ mask,
                                                             int *__restrict__ out, const int *__restrict__ flags, int64_t n)
{
    const int b = blockIdx.y;
    if (flags[b] == 0) return;
    const int64_t base = (int64_t)b * n;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256)
        out[base + i] = mask[base + i] ? markers[base + i] : 0;
}

template <int EX_LDS, int D>
__global__ void __launch_bounds__(64) ws_exact_kernel(const unsigned *__restrict__ val, const int *__restrict__ markers,
                                                       const uint8_t *__restrict__ mask, int *__restrict__ out,
                                                       const int *__restrict__ flags, unsigned long long *__restrict__ heap_key,
                                                       unsigned *__restrict__ heap_idx, int H, int W)
{
    __shared__ __attribute__((aligned(16))) unsigned long long lds_key[EX_LDS];
    __shared__ __attribute__((aligned(16))) unsigned lds_idx[EX_LDS];
    const int b = blockIdx.x;
    if (flags[b] == 0) return;
    const int64_t n = (int64_t)H * W;
    const unsigned *v = val + (int64_t)b * n;
    const uint8_t *ms = mask + (int64_t)b * n;
    int *o = out + (int64_t)b * n;
    const int t = threadIdx.x;
    // slot d (1-based) of the heap is element d - 1 of the sequential heap.  Slots >= EX_LDS live in the workspace, which
    // holds n elements per frame: slot d -> element d - 1, so that slot n (every pixel is pushed at most once) still fits
    ExactHeap h{(ex_lds_key_t *)lds_key, (ex_lds_idx_t *)lds_idx, heap_key + (int64_t)b * n - 1, heap_idx + (int64_t)b * n - 1, EX_LDS, 0, 0ull, 0u};
    for (int64_t base = 0; base < n; base += WAVE) {
        const int64_t i = base + t;
        const int seed = i < n ? o[i] : 0;
        const unsigned sv = i < n ? v[i] : 0u;
        unsigned long long m = __ballot(seed != 0);
        while (m) {
            const int l = __ffsll((long long)m) - 1;
            m &= m - 1;
            const unsigned si = (unsigned)(base + l);
            ex_push(h, (unsigned long long)lane_u32(sv, l) << 32, ((si / (unsigned)W) << 16) | (si % (unsigned)W));
        }
    }
    unsigned age = 0;
    while (h.items > 0) {
        const unsigned e_idx = lane_u32(h.idx_at(1), 0);
        const int r0 = (int)(e_idx >> 16), c0 = (int)(e_idx & 0xFFFFu);  // heap entries carry (row << 16) | col
        // lanes 0..3 look at the neighbours in the reference's order (up, left, right, down) while the heap is repaired
        const int rr = r0 + (t == 0 ? -1 : t == 3 ? 1 : 0), cc = c0 + (t == 1 ? -1 : t == 2 ? 1 : 0);
        bool open = false;
        unsigned qv = 0;
        const int64_t q = rowoff(rr, W) + cc;
        if (t < 4 && rr >= 0 && rr < H && cc >= 0 && cc < W) {
            const uint8_t in_mask = ms[q];  // three independent loads, in flight together
            const int taken = o[q];
            qv = v[q];
            open = in_mask != 0 && taken == 0;
        }
        const int lab = o[rowoff(r0, W) + c0];
        ex_pop<D>(h);
        unsigned long long m = __ballot(open);
        while (m) {
            const int l = __ffsll((long long)m) - 1;
            m &= m - 1;
            const unsigned qi = lane_u32((unsigned)q, l);
            ++age;
            if (t == 0) o[qi] = lab;
            ex_push(h, ((unsigned long long)lane_u32(qv, l) << 32) | age, lane_u32(((unsigned)rr << 16) | (unsigned)cc, l));
        }
    }
}

}  // namespace pcseg

using namespace pcseg;

// Round 0 stops every tile after this many iterations (one sweep per direction or quadrant each): the round after it
// visits every tile anyway (with the other tiling), so squeezing the last changes out of isolated tiles is wasted work;
// a tile cut short marks all four corner tiles.  Later grid rounds have a limit of their own (a tile cut short is
// finished by its corner tiles or a later round; the per-frame tail kernel has no limit).  Measured on the benchmark
// batch, relaxation per launch -- line sweeps: round 0 at 16 is 4.6 % better than no limit (8 -> 1 %, 6 -> none), later
// rounds no limit 179.9 us, 32 -> 177.5, 16 -> 170.5, 8 -> 175.1 plus 122 us of tail kernel; quadrant sweeps (which
// need about a third of the iterations): 4 / 6 -> 136 us, 3 / 4 -> 134, 5 / 5 -> 137, 4 / 16 -> 139, 16 / 16 -> 154.
#ifndef PCSEG_WS_ROUND0_SWEEPS
#define PCSEG_WS_ROUND0_SWEEPS (PCSEG_WS_FSM ? 4 : 16)
#endif
#ifndef PCSEG_WS_ROUND_SWEEPS
#define PCSEG_WS_ROUND_SWEEPS (PCSEG_WS_FSM ? 6 : 16)
#endif
#ifndef PCSEG_WS_RELAX_TILE
#define PCSEG_WS_RELAX_TILE 64
#endif
constexpr int WS_ROUND0_SWEEPS = PCSEG_WS_ROUND0_SWEEPS;
#ifndef PCSEG_WS_RELAX_LDS_PAD
// A/B aid: extra dynamic LDS per relaxation block.  0 = four tiles per CU (4 x 36 KB); 16384 -> three, 40960 -> two: fewer of a
// CU's waves belong to the relaxation and more of its LDS is left to the kernels of the other batches in flight
#define PCSEG_WS_RELAX_LDS_PAD 0
#endif

// the watershed may be called from several host threads at once (FramePipeline's lanes)
static std::atomic<long long> g_ws_counters[4];  // [0] unused (lives on the device), relax launches, calls, -
static unsigned long long *g_ws_dev_tiles[64] = {nullptr};  // per device: relaxation tiles processed since the last reset
static std::mutex g_ws_dev_mutex;

static unsigned long long *ws_dev_tiles()
{
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return nullptr;
    std::lock_guard<std::mutex> lk(g_ws_dev_mutex);
    if (!g_ws_dev_tiles[dev]) {
        unsigned long long *p = nullptr;
        if (hipMalloc((void **)&p, sizeof(unsigned long long)) != hipSuccess) return nullptr;
        (void)hipMemset(p, 0, sizeof(unsigned long long));
        g_ws_dev_tiles[dev] = p;
    }
    return g_ws_dev_tiles[dev];
}

extern "C" {

void pcseg_watershed_counters(int64_t *out, int reset)
{
    unsigned long long *dev = ws_dev_tiles();
    if (out) {
        unsigned long long tiles = 0;
        if (dev) (void)hipMemcpy(&tiles, dev, sizeof(tiles), hipMemcpyDeviceToHost);  // blocking: waits for the work queued so far
        out[0] = (int64_t)tiles;
        for (int i = 1; i < 4; ++i) out[i] = g_ws_counters[i].load();
    }
    if (reset) {
        if (dev) {
            (void)hipDeviceSynchronize();
            (void)hipMemset(dev, 0, sizeof(unsigned long long));
        }
        for (int i = 0; i < 4; ++i) g_ws_counters[i].store(0);
    }
}

size_t pcseg_watershed_workspace_bytes(int B, int H, int W)
{
    if (!check_shape(B, H, W)) return 0;
    size_t n = (size_t)B * H * W;
    int tilesX = (W + WS_T - 1) / WS_T, tilesY = (H + WS_T - 1) / WS_T;
    return 3 * align_up(n * 4) + 3 * align_up(n) + 2 * align_up((size_t)B * (tilesX + 1) * (tilesY + 1)) + align_up((size_t)B * tilesX * tilesY) + align_up(sizeof(int) * WS_CHANGED_INTS) + 2 * align_up(sizeof(int) * B) + align_up(sizeof(int) * ((size_t)B + 1)) +
           align_up(n * 8) + align_up(n * 4) + align_up(sizeof(int) * ((size_t)B * tilesX * tilesY + 1)) +
           align_up(sizeof(int) * 16) + 2 * align_up(sizeof(int) * (size_t)B * (tilesX + 1) * (tilesY + 1));
}

// grid rounds enqueued before the per-frame tail kernels take over (see ws_relax_tail_kernel): the benchmark batch needs
// 10 relaxation rounds; a round without marks costs a few microseconds.  The second level sees a few dozen tiles of a
// few frames (benchmark batch: 31 tiles in 7 frames; rounds of 127 / 60 / 25 / 18 us -- the first is one winding lake's
// fixed point); with two grid rounds the per-frame tail kernel walks the rest one tile at a time (177 us against 43)
#ifndef PCSEG_WS_K2_ROUNDS
#define PCSEG_WS_K2_ROUNDS 4
#endif
constexpr int WS_GRID_ROUNDS = 12, WS_K2_GRID_ROUNDS = PCSEG_WS_K2_ROUNDS;

int pcseg_watershed4_f32(const float *img, int64_t frame_stride, const int32_t *markers, const uint8_t *mask, int32_t *out,
                         int32_t *tie_flags, int B, int H, int W, int mode, void *workspace, size_t workspace_bytes,
                         pcseg_stream_t stream)
{
    PCSEG_REQUIRE(img && markers && mask && out && workspace && check_shape(B, H, W) && mode >= 0 && (mode & 3) <= 2 && mode < 8, "bad arguments");
    const bool verify = (mode & 4) != 0;  // also run the explicit per-pixel proof check (implied by the component test)
    mode &= 3;
    PCSEG_REQUIRE(frame_stride == 0 || frame_stride >= (int64_t)H * W, "frame_stride smaller than a frame");
    if (frame_stride == 0) frame_stride = (int64_t)H * W;
    hipStream_t s = (hipStream_t)stream;
    const size_t n = (size_t)B * H * W;
    const int tilesX = (W + WS_T - 1) / WS_T, tilesY = (H + WS_T - 1) / WS_T;
    const size_t ntiles = (size_t)B * tilesX * tilesY;
    const size_t ntiles_max = (size_t)B * (tilesX + 1) * (tilesY + 1);  // the half-tile-shifted tiling has one more per axis
    Carver cv(workspace, workspace_bytes);
    unsigned *val = cv.take<unsigned>(n);
    unsigned *L = cv.take<unsigned>(n);
    uint8_t *dirtyA = cv.take<uint8_t>(ntiles_max);
    uint8_t *dirtyB = cv.take<uint8_t>(ntiles_max);
    uint8_t *active_tiles = cv.take<uint8_t>(ntiles);
    int *changed = cv.take<int>(WS_CHANGED_INTS);  // [6] "a tail kernel gave up", [WS_CNT0 + 32 i] tile counters
    int *flags = cv.take<int>(B);
    int *flags2 = cv.take<int>(B);
    int *frame_list = cv.take<int>(B + 1) + 1;  // frame_list[-1] = number of flagged frames (stays on the device)
    int *round_count = cv.take<int>(16);            // [r] = tiles listed for relaxation round r (see ws_relax_list_kernel)
    int *tile_list = cv.take<int>(ntiles + 1) + 1;  // the second level's active tiles, tile_list[-1] = their number
    int *round_list[2] = {cv.take<int>(ntiles_max), cv.take<int>(ntiles_max)};
    static_assert(WS_GRID_ROUNDS < 16, "one counter per grid round");
    unsigned long long *heap_key = cv.take<unsigned long long>(n);  // doubles as K64 of the second-level pass
    unsigned *heap_idx = cv.take<unsigned>(n);                      // doubles as K2
    int *uf_parent = cv.take<int>(n);
    uint8_t *uf_bad1 = cv.take<uint8_t>(n);  // roots of components the first / second level cannot resolve
    uint8_t *uf_bad2 = cv.take<uint8_t>(n);
    uint8_t *uf_mask = cv.take<uint8_t>(n);
    if (!cv.ok()) {
        set_error("watershed: workspace too small (%zu < %zu)", workspace_bytes, cv.off);
        return PCSEG_ERR_WORKSPACE;
    }
    const dim3 pgrid((W + 63) / 64, (H + 3) / 4, B);
    // No host round trip anywhere below: fixed points run a fixed number of grid rounds and finish in a per-frame tail
    // kernel, the frames that need the second level are listed (and counted) on the device and the second-level
    // kernels cover the worst-case grid, from which the blocks of unlisted frames leave at once.
    const int max_rounds = (tilesX * tilesY + 64) * 64;
    // ONE fill for everything that starts at zero: the two mark buffers (round 0 visits every tile regardless), the active-tile
    // set, the counters and both frame-flag arrays are carved next to each other (five separate fills were five launches)
    // (... and, behind the frame list, the number of listed tiles)
    PCSEG_CHECK_HIP(hipMemsetAsync(dirtyA, 0, (size_t)((char *)tile_list - (char *)dirtyA), s));
    long long relax_launches = 0;
    if (mode == 1) {
        PCSEG_LAUNCH(ws_init_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, img, frame_stride, markers, mask, val, L,
                     out, (int64_t)H * W, (int64_t)n);
        PCSEG_CHECK_LAUNCH();
        PCSEG_LAUNCH(ws_set_flags_kernel, dim3((B + 63) / 64), dim3(64), 0, s, flags2, B, 1);
        PCSEG_CHECK_LAUNCH();
    } else {
        const bool vec = (W & 3) == 0 && W >= 4 && (frame_stride & 3) == 0 && (((uintptr_t)img | (uintptr_t)markers | (uintptr_t)out |
                                                                                 (uintptr_t)val | (uintptr_t)L) & 15) == 0 &&
                         ((uintptr_t)mask & 3) == 0;
        const WsInputs inputs{img, frame_stride, markers, mask, out, vec};
        // minimax relaxation over alternating tilings
        constexpr int RT = PCSEG_WS_RELAX_TILE;  // edge of a relaxation tile (the later stages keep their 64 x 64 tiles)
        using RG = RelaxGeom<RT>;
        const WsTiling tilings[2] = {{0, (W + RT - 1) / RT, (H + RT - 1) / RT},
                                     {RT / 2, (W + RT / 2 + RT - 1) / RT, (H + RT / 2 + RT - 1) / RT}};
        static std::atomic<bool> lds_attr_set[64];
        {
            int dev = 0;
            if (RG::LDS_BYTES + PCSEG_WS_RELAX_LDS_PAD > 64 * 1024 && hipGetDevice(&dev) == hipSuccess && dev >= 0 && dev < 64 && !lds_attr_set[dev].load()) {
                PCSEG_CHECK_HIP(hipFuncSetAttribute((const void *)ws_relax_kernel<RT>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                                    (int)(RG::LDS_BYTES + PCSEG_WS_RELAX_LDS_PAD)));
                PCSEG_CHECK_HIP(hipFuncSetAttribute((const void *)ws_relax_list_kernel<RT>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                                    (int)(RG::LDS_BYTES + PCSEG_WS_RELAX_LDS_PAD)));
                PCSEG_CHECK_HIP(hipFuncSetAttribute((const void *)ws_relax_tail_kernel<RT>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                                    (int)RG::LDS_BYTES));
                lds_attr_set[dev].store(true);
            }
        }
        // (frame flags and mark buffers: cleared by the one fill above -- the tail kernels may raise flags2 for a fixed point
        // they had to abandon)
        {
            uint8_t *din = dirtyA, *dout = dirtyB;
            for (int round = 0; round < WS_GRID_ROUNDS; ++round) {
                const WsTiling &cur = tilings[round & 1], &nxt = tilings[(round + 1) & 1];
                // (a round's marks go into the next round's list once that round walks a list)
                const bool lists_next = round + 1 >= WS_LIST_FROM && round + 1 < WS_GRID_ROUNDS;
                int *lout = lists_next ? round_list[(round + 1) & 1] : nullptr, *cout = lists_next ? round_count + round + 1 : nullptr;
                if (round >= WS_LIST_FROM)
                    PCSEG_LAUNCH(ws_relax_list_kernel<RT>, dim3(WS_LIST_GRID), dim3(RG::THREADS), RG::LDS_BYTES + PCSEG_WS_RELAX_LDS_PAD, s, inputs,
                                 val, L, din, dout, changed, H, W, cur, nxt, PCSEG_WS_ROUND_SWEEPS, (const int *)round_list[round & 1],
                                 (const int *)(round_count + round), lout, cout);
                else
                PCSEG_LAUNCH(ws_relax_kernel<RT>, dim3(cur.nx, cur.ny, B), dim3(RG::THREADS), RG::LDS_BYTES + PCSEG_WS_RELAX_LDS_PAD, s, inputs, round == 0, val,
                             L, din, dout, changed, H, W, cur, nxt, round == 0 ? WS_ROUND0_SWEEPS : PCSEG_WS_ROUND_SWEEPS, lout, cout);
                PCSEG_CHECK_LAUNCH();
                ++relax_launches;
                uint8_t *t = din; din = dout; dout = t;
            }
            PCSEG_LAUNCH(ws_relax_tail_kernel<RT>, dim3(B), dim3(RG::THREADS), RG::LDS_BYTES, s, inputs, val, L, din, dout, changed,
                         changed + 6, flags2, H, W, tilings[0], tilings[1], WS_GRID_ROUNDS, max_rounds);
            PCSEG_CHECK_LAUNCH();
        }
        const dim3 ugrid_full((W + UF_TW - 1) / UF_TW, (H + UF_TH - 1) / UF_TH, B);
        const dim3 lgrid_full((unsigned)(((size_t)H * W + 255) / 256), B);
        const int64_t npx = (int64_t)H * W;
        const int64_t border_px = (int64_t)((H - 1) / UF_TH) * W + (int64_t)((W - 1) / UF_TW) * H;
        const dim3 bgrid_full((unsigned)((border_px + 255) / 256), B);
        // label assignment = union-find over "minimum-key neighbour" links.  A component holding two marker ids flags
        // its frame, stays unlabelled and (first level) marks its tiles active for the next level.
        const WsTileList tiles{tile_list, tilesX * tilesY, tilesX};
        auto assign_labels = [&](auto *keys, const int *flist, const uint8_t *act, int *out_flags, bool first_level) -> int {
            using KeyT = std::remove_const_t<std::remove_pointer_t<decltype(keys)>>;
            const int span = ws_frame_span(flist, B);  // frame dimension of the grids (see ws_for_frames)
            const dim3 ugrid(ugrid_full.x, ugrid_full.y, span), bgrid(bgrid_full.x, span), lgrid(lgrid_full.x, span);
            uint8_t *level_bad = first_level ? uf_bad1 : uf_bad2;  // cleared by the tile pass, tile by tile
            if (flist)  // second level: the listed tiles only
                PCSEG_LAUNCH((ws_uf_tile_kernel<KeyT, true>), dim3(WS_TILE_GRID), dim3(256), 0, s, tiles, (const KeyT *)keys, (const int *)out, act,
                             uf_parent, uf_mask, level_bad, H, W, tilesX, tilesY);
            else
                PCSEG_LAUNCH((ws_uf_tile_kernel<KeyT, false>), ugrid, dim3(256), 0, s, tiles, (const KeyT *)keys, (const int *)out, act,
                             uf_parent, uf_mask, level_bad, H, W, tilesX, tilesY);
            PCSEG_CHECK_LAUNCH();
            if (border_px > 0) {
                PCSEG_LAUNCH(ws_uf_border_kernel, bgrid, dim3(256), 0, s, flist, (const uint8_t *)uf_mask, act, uf_parent, H, W,
                             tilesX, tilesY, flags2);
                PCSEG_CHECK_LAUNCH();
            }
            if (first_level && act == nullptr && flist == nullptr && (W & 3) == 0 && (((uintptr_t)out | (uintptr_t)uf_parent) & 15) == 0 &&
                ((uintptr_t)uf_mask & 3) == 0) {
                PCSEG_LAUNCH(ws_uf_label4_kernel, dim3((unsigned)((npx + 1024 * LABEL4_Q - 1) / (1024 * LABEL4_Q)), B), dim3(256), 0, s,
                             (const int *)uf_parent, (const uint8_t *)uf_mask, out, uf_bad1, out_flags, npx, flags2);
                PCSEG_CHECK_LAUNCH();
            } else if (first_level) {
                PCSEG_LAUNCH(ws_uf_label_kernel<UF_OPTIMISTIC>, lgrid, dim3(256), 0, s, flist, (const int *)uf_parent, out, act,
                             uf_bad1, markers, mask, out_flags, (uint8_t *)nullptr, npx, W, tilesX, tilesY, flags2);
                PCSEG_CHECK_LAUNCH();
            } else {
                PCSEG_LAUNCH(ws_uf_label_tiles_kernel<UF_DETECT>, dim3(WS_TILE_GRID), dim3(256), 0, s, tiles, (const int *)uf_parent, out, act,
                             uf_bad2, markers, mask, out_flags, H, W, tilesX, tilesY, flags2);
                PCSEG_CHECK_LAUNCH();
                PCSEG_LAUNCH(ws_uf_label_tiles_kernel<UF_ASSIGN>, dim3(WS_TILE_GRID), dim3(256), 0, s, tiles, (const int *)uf_parent, out, act,
                             uf_bad2, markers, mask, out_flags, H, W, tilesX, tilesY, flags2);
                PCSEG_CHECK_LAUNCH();
            }
            return PCSEG_OK;
        };
        // both mark buffers start the second level empty (a fixed point leaves them empty; one that was abandoned may not):
        // they are carved next to each other -- one fill
        PCSEG_CHECK_HIP(hipMemsetAsync(dirtyA, 0, (size_t)((char *)dirtyB - (char *)dirtyA) + ntiles_max, s));
        int rc = assign_labels((const unsigned *)L, (const int *)nullptr, (const uint8_t *)nullptr, flags, true);
        if (rc) return rc;
        if (verify) {
            PCSEG_LAUNCH(ws_check_kernel<unsigned>, pgrid, dim3(256), 0, s, (const unsigned *)L, (const int *)out, markers, mask,
                         (const int *)nullptr, flags, H, W);
            PCSEG_CHECK_LAUNCH();
        }
        // which frames need the second level: list and length stay on the device
        PCSEG_LAUNCH(ws_list_flagged_kernel, dim3(1), dim3(64), 0, s, (const int *)flags, B, frame_list);
        PCSEG_CHECK_LAUNCH();
        {
            unsigned *K2 = heap_idx;
            unsigned long long *K64 = heap_key;
            const int span = ws_frame_span(frame_list, B);
            const dim3 tgrid(tilesX, tilesY, span), lgrid(lgrid_full.x, span), pgrid2(pgrid.x, pgrid.y, span);
            // the components that hold two marker ids go back to their seeds; their tiles are the second level's work
            // (uf_bad2 doubles as the per-pixel "in an unresolved component" mark until the second level's tile pass clears
            // it; the verification build checks keys over whole frames and keeps every lake)
            uint8_t *in_bad = verify ? nullptr : uf_bad2;
            PCSEG_LAUNCH(ws_uf_label_kernel<UF_REPAIR>, lgrid, dim3(256), 0, s, (const int *)frame_list, (const int *)uf_parent, out,
                         (const uint8_t *)nullptr, uf_bad1, markers, mask, flags, active_tiles, npx, W, tilesX, tilesY, flags2, in_bad,
                         dirtyA);
            PCSEG_CHECK_LAUNCH();
            // (the repair pass writes the active set -- still all zero since the call's first fill -- and the first round's
            // marks itself: two copies and a fill between the kernels gone)
            if (verify) {
                // whole flagged frames, so that the explicit per-pixel check of the second level sees valid keys everywhere
                PCSEG_LAUNCH(ws_activate_frames_kernel, lgrid, dim3(256), 0, s, (const int *)frame_list, (const int *)flags, active_tiles,
                             dirtyA, markers, mask, out, npx, W, tilesX, tilesY);
                PCSEG_CHECK_LAUNCH();
            }
            // the active tiles as a list: everything below walks it with small fixed grids (see WsTileList)
            PCSEG_LAUNCH(ws_list_tiles_kernel, dim3(span), dim3(256), 0, s, (const int *)frame_list, (const uint8_t *)active_tiles,
                         tilesX * tilesY, tile_list);
            PCSEG_CHECK_LAUNCH();
            PCSEG_LAUNCH(ws_k2_init_kernel, dim3(WS_TILE_GRID), dim3(256), 0, s, tiles, (const unsigned *)val, (const unsigned *)L, markers, mask,
                         K2, H, W);
            PCSEG_CHECK_LAUNCH();
            // marks of the first round = the active tiles (written by the repair pass); the other buffer is empty
            uint8_t *din = dirtyA, *dout = dirtyB;
            for (int round = 0; round < WS_K2_GRID_ROUNDS; ++round) {
                PCSEG_LAUNCH(ws_k2_relax_kernel, dim3(WS_TILE_GRID), dim3(K2T), 0, s, tiles, (const unsigned *)val,
                             (const unsigned *)L, K2, (const uint8_t *)active_tiles, din, dout, H, W, tilesX, tilesY,
                             (const uint8_t *)in_bad);
                PCSEG_CHECK_LAUNCH();
                uint8_t *t = din; din = dout; dout = t;
            }
            PCSEG_LAUNCH(ws_k2_relax_tail_kernel, dim3(B), dim3(K2T), 0, s, (const int *)frame_list, (const unsigned *)val,
                         (const unsigned *)L, K2, (const uint8_t *)active_tiles, din, dout, changed + 6, flags2, H, W, tilesX, tilesY,
                         max_rounds, (const uint8_t *)in_bad);
            PCSEG_CHECK_LAUNCH();
            PCSEG_LAUNCH(ws_pack_kernel, dim3(WS_TILE_GRID), dim3(256), 0, s, tiles, (const unsigned *)L, (const unsigned *)K2,
                         (const uint8_t *)active_tiles, K64, H, W, tilesX, tilesY);
            PCSEG_CHECK_LAUNCH();
            rc = assign_labels((const unsigned long long *)K64, (const int *)frame_list, (const uint8_t *)active_tiles, flags2, false);
            if (rc) return rc;
            if (verify) {
                PCSEG_LAUNCH(ws_check_kernel<unsigned long long>, pgrid, dim3(256), 0, s, (const unsigned long long *)K64,
                             (const int *)out, markers, mask, (const int *)flags, flags2, H, W);
                PCSEG_CHECK_LAUNCH();
            }
        }
    }
    if (tie_flags) PCSEG_CHECK_HIP(hipMemcpyAsync(tie_flags, flags2, sizeof(int) * B, hipMemcpyDeviceToDevice, s));
    if (mode != 2) {
        const int64_t npx = (int64_t)H * W;
        PCSEG_LAUNCH(ws_exact_init_kernel, dim3((unsigned)std::min<int64_t>((npx + 1023) / 1024, 64), B), dim3(256), 0, s, markers, mask,
                     out, (const int *)flags2, npx);
        PCSEG_CHECK_LAUNCH();
        int ncu = 256;
        int dev = 0;
        if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, dev);
        if (B <= ncu)
            PCSEG_LAUNCH((ws_exact_kernel<EX_LDS_BIG, EX_D_BIG>), dim3(B), dim3(64), 0, s, val, markers, mask, out, flags2, heap_key, heap_idx, H, W);
        else
            PCSEG_LAUNCH((ws_exact_kernel<EX_LDS_SMALL, EX_D_SMALL>), dim3(B), dim3(64), 0, s, val, markers, mask, out, flags2, heap_key, heap_idx, H, W);
        PCSEG_CHECK_LAUNCH();
    }
    if (unsigned long long *dev_tiles = ws_dev_tiles()) {
        PCSEG_LAUNCH(ws_accumulate_kernel, dim3(1), dim3(64), 0, s, (const int *)changed, dev_tiles);
        PCSEG_CHECK_LAUNCH();
    }
    g_ws_counters[1] += relax_launches;
    g_ws_counters[2] += 1;
    return PCSEG_OK;
}

}  // extern "C"
