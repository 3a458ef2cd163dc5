// LDS tile building blocks shared by the stand-alone kernels (median5, ccl_tile) and the fused class-map front end.
#pragma once
#include "common.h"

namespace pcseg {

constexpr int CCL_TW = 64, CCL_TH = 32, CCL_TILE = CCL_TW * CCL_TH;
static_assert(CCL_TH == 32 && CCL_TW == 64, "the tile pass assumes one 64-lane row per trip and one 32-row bit word per column");
constexpr int MED_TW = 64, MED_TH = 32, MED_LW = 72 /* 64 + 4 rounded to 8 */, MED_LH = MED_TH + 4;

// scipy mode='reflect' (numpy 'symmetric'): d c b a | a b c d | d c b a
__device__ __forceinline__ int reflect_idx(int i, int n)
{
    int p = 2 * n;
    i %= p;
    if (i < 0) i += p;
    return i < n ? i : p - 1 - i;
}

// 5x5 medians of a 4-pixel strip from one-hot histogram words (values <= 5): hot[] holds 1 << (5 * value) for the
// (MED_LH x MED_LW) tile with its 2-pixel halo; the 25 words of a window ADD up to the window's histogram (a count is
// at most 25 < 32, so the 5-bit fields never carry) and the median is the first value whose cumulative count reaches
// 13.  Column sums of 5 rows are shared by the 4 outputs: 42 adds instead of 300 compares.  (lr, lc): strip origin in
// tile coordinates (lc a multiple of 4).
__device__ __forceinline__ void median5_hot_strip(const uint32_t *hot, int lr, int lc, uint32_t med[4])
{
    uint32_t col[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
    for (int dr = 0; dr < 5; ++dr) {
        const uint4 *row = reinterpret_cast<const uint4 *>(hot + (lr + dr) * MED_LW + lc);
        const uint4 a = row[0], bq = row[1];
        col[0] += a.x; col[1] += a.y; col[2] += a.z; col[3] += a.w;
        col[4] += bq.x; col[5] += bq.y; col[6] += bq.z; col[7] += bq.w;
    }
    uint32_t w = col[0] + col[1] + col[2] + col[3] + col[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        if (j > 0) w += col[j + 4] - col[j - 1];
        // cumulative counts of the values <= k in field k (no carries: every prefix is <= 25)
        const uint32_t cum = w * 0x02108421u;
        uint32_t m = 0;
#pragma unroll
        for (int k = 0; k < 5; ++k) m += ((cum >> (5 * k)) & 31u) < 13u ? 1u : 0u;
        med[j] = m;
    }
}

// The same for CLASSES 1 .. 5 in 6-bit fields (hot[] holds 1 << (6 * (class - 1)): five fields, 30 bits).  The spare bit of a
// field turns the five "cumulative count < 13" tests of a pixel into one subtraction: 44 - count is at least 32 exactly when
// count <= 12 (no borrow crosses a field: counts are at most 25), so the median is 1 + the number of fields 0 .. 3 whose bit 5
// is set -- one multiply, one subtract, one and, one bit count per pixel instead of five extract / compare / add triples.
__device__ __forceinline__ void median5_hot6_strip(const uint32_t *hot, int lr, int lc, uint32_t med[4])
{
    uint32_t col[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
    for (int dr = 0; dr < 5; ++dr) {
        const uint4 *row = reinterpret_cast<const uint4 *>(hot + (lr + dr) * MED_LW + lc);
        const uint4 a = row[0], bq = row[1];
        col[0] += a.x; col[1] += a.y; col[2] += a.z; col[3] += a.w;
        col[4] += bq.x; col[5] += bq.y; col[6] += bq.z; col[7] += bq.w;
    }
    uint32_t w = col[0] + col[1] + col[2] + col[3] + col[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        if (j > 0) w += col[j + 4] - col[j - 1];
        const uint32_t cum = w * 0x01041041u;            // field k: pixels of the window with a class <= k + 1
        const uint32_t below = 0x2CB2CB2Cu - cum;        // 44 in every field (fields 0 .. 4), minus the counts
        med[j] = 1u + __popc(below & 0x00820820u);       // bit 5 of the fields 0 .. 3
    }
}

// Two strips, one below the other (rows lr and lr + 1): six rows of words are read ONCE, all twelve 16-byte reads in flight
// together, and the lower strip's column sums are the upper one's minus row lr plus row lr + 5.
__device__ __forceinline__ void median5_hot6_strip2(const uint32_t *hot, int lr, int lc, uint32_t med0[4], uint32_t med1[4])
{
    uint4 a[6], bq[6];
#pragma unroll
    for (int dr = 0; dr < 6; ++dr) {
        const uint4 *row = reinterpret_cast<const uint4 *>(hot + (lr + dr) * MED_LW + lc);
        a[dr] = row[0];
        bq[dr] = row[1];
    }
    uint32_t col[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
    for (int dr = 0; dr < 5; ++dr) {
        col[0] += a[dr].x; col[1] += a[dr].y; col[2] += a[dr].z; col[3] += a[dr].w;
        col[4] += bq[dr].x; col[5] += bq[dr].y; col[6] += bq[dr].z; col[7] += bq[dr].w;
    }
#pragma unroll
    for (int half = 0; half < 2; ++half) {
        if (half) {
            col[0] += a[5].x - a[0].x; col[1] += a[5].y - a[0].y; col[2] += a[5].z - a[0].z; col[3] += a[5].w - a[0].w;
            col[4] += bq[5].x - bq[0].x; col[5] += bq[5].y - bq[0].y; col[6] += bq[5].z - bq[0].z; col[7] += bq[5].w - bq[0].w;
        }
        uint32_t w = col[0] + col[1] + col[2] + col[3] + col[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            if (j > 0) w += col[j + 4] - col[j - 1];
            const uint32_t below = 0x2CB2CB2Cu - w * 0x01041041u;
            (half ? med1 : med0)[j] = 1u + __popc(below & 0x00820820u);
        }
    }
}

// Union-find of one 64x32 tile in LDS: key[] (0 = background, equal non-zero keys connect) -> par[] such that
// find_lds(par, i) is the tile-local root (smallest index) of pixel i; par[i] = -1 for background.  Row runs are
// pre-linked without atomics (a wave covers one 64-pixel tile row per trip, run heads come from a ballot, every pixel
// points straight at its run's first pixel), vertical / diagonal links only where a run does not already imply them.
// Contains the barriers it needs; key[] must be complete (and synchronised) on entry.
template <bool CONN8>
__device__ __forceinline__ void ccl_tile_unions(const int *key, int *par)
{
    for (int i = threadIdx.x; i < CCL_TILE; i += 256) {
        const int k = key[i], lc = i % CCL_TW;
        if (__ballot(k != 0) == 0) {  // a tile row without a foreground pixel (sparse keys: the local-maxima candidates)
            par[i] = -1;
            continue;
        }
        const bool head = lc == 0 || key[i - 1] != k;
        const unsigned long long heads = __ballot(head);
        const unsigned long long upto = heads & (lc == 63 ? ~0ull : ((2ull << lc) - 1ull));
        const int start = 63 - __clzll((long long)upto);
        par[i] = k == 0 ? -1 : (i - lc + start);
    }
    __syncthreads();
    for (int i = threadIdx.x; i < CCL_TILE; i += 256) {
        int k = key[i], lc = i % CCL_TW;
        if (k == 0 || i < CCL_TW) continue;
        bool w = lc > 0 && key[i - 1] == k;
        bool n = key[i - CCL_TW] == k;
        bool nw = lc > 0 && key[i - CCL_TW - 1] == k;
        if (n && !(w && nw)) unite_lds_pair(par, i, i - CCL_TW);
        if (CONN8) {
            bool ne = lc < CCL_TW - 1 && key[i - CCL_TW + 1] == k;
            if (ne && !n) unite_lds_pair(par, i, i - CCL_TW + 1);
            if (nw && !n && !w) unite_lds_pair(par, i, i - CCL_TW - 1);
        }
    }
    __syncthreads();
}

// tile-local roots -> frame-wide parent entries (linear pixel index of the root, -1 = background)
__device__ __forceinline__ void ccl_tile_store(const int *key, int *par, int *__restrict__ parent, int64_t fbase, int r0, int c0,
                                               int H, int W)
{
    for (int i = threadIdx.x; i < CCL_TILE; i += 256) {
        int r = r0 + i / CCL_TW, c = c0 + i % CCL_TW;
        if (r >= H || c >= W) continue;
        int v = -1;
        if (key[i] != 0) {
            int root = find_lds(par, i);
            v = (r0 + root / CCL_TW) * W + c0 + root % CCL_TW;
        }
        parent[fbase + (int64_t)r * W + c] = v;
    }
}

}  // namespace pcseg
