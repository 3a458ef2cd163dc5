// Connected-component labelling (A2) and everything built on it: raster-order
// label compaction (K2c), binary_fill_holes (A7), plateau-aware local maxima +
// marker numbering (R3/R4) and the DAPI/RFP overlap removal (C6).
//
// Union-find with "root = smallest linear index": a tile-local pass in LDS
// (64x32 pixels per 256-thread block, row runs pre-linked without atomics,
// vertical / diagonal links only where they are not implied by a run), one
// global pass over tile borders with agent-scope atomicMin, one flatten pass.
// Because the root of a component is its raster-first pixel, skimage's label
// numbering is the rank of the root among all roots: wave ballot/popcount +
// block scan + one exclusive scan of the block totals.
#include <type_traits>

#include "common.h"
#include "tile_ops.h"

namespace pcseg {

constexpr int SCAN_PIX = 1024;  // pixels per block in the count / assign / relabel passes

// ---- key functors: key(b, r, c) of pixel (r, c) of frame b; 0 = background, equal non-zero keys connect
struct KeyEqU8 {
    const uint8_t *p;
    int W;
    int64_t n;
    __device__ __forceinline__ int operator()(int b, int r, int c) const { return p[b * n + rowoff(r, W) + c]; }
};
struct KeyNzU8 {
    const uint8_t *p;
    int W;
    int64_t n;
    __device__ __forceinline__ int operator()(int b, int r, int c) const { return p[b * n + rowoff(r, W) + c] != 0; }
};
struct KeyZeroU8 {  // background components (fill holes)
    const uint8_t *p;
    int W;
    int64_t n;
    __device__ __forceinline__ int operator()(int b, int r, int c) const { return p[b * n + rowoff(r, W) + c] == 0; }
};
struct KeyIsOneU8 {  // dapi == 1
    const uint8_t *p;
    int W;
    int64_t n;
    __device__ __forceinline__ int operator()(int b, int r, int c) const { return p[b * n + rowoff(r, W) + c] == 1; }
};
struct KeyI32 {
    const int32_t *p;
    int W;
    int64_t n;
    __device__ __forceinline__ int operator()(int b, int r, int c) const { return p[b * n + rowoff(r, W) + c]; }
};
// candidate pixels of the local-maxima pass: one bit per pixel in a FLAT bit array over the batch (bit b * n + r * W + c),
// key = the pixel's value where the bit is set (0 -> INT_MIN: keys must be non-zero), 0 elsewhere
__device__ __forceinline__ bool cand_bit(const unsigned long long *bits, int64_t g) { return (bits[g >> 6] >> (g & 63)) & 1ull; }
struct KeyCandBits {
    const int32_t *img;
    const unsigned long long *bits;
    int W;
    int64_t n;
    __device__ __forceinline__ int operator()(int b, int r, int c) const
    {
        const int64_t g = b * n + rowoff(r, W) + c;
        if (!cand_bit(bits, g)) return 0;
        const int v = img[g];
        return v == 0 ? (int)0x80000000 : v;
    }
};
struct KeyBits {  // 1 bit per pixel in 32-row column words: words[(b * nch + r / 32) * W + c] bit (r % 32)
    const unsigned *words;
    int W, nch;
    __device__ __forceinline__ int operator()(int b, int r, int c) const
    {
        return (words[((int64_t)b * nch + (r >> 5)) * W + c] >> (r & 31)) & 1u;
    }
};

template <typename KeyFn, bool CONN8>
__global__ void __launch_bounds__(256) ccl_tile_kernel(KeyFn keyfn, int *__restrict__ parent, int H, int W)
{
    __shared__ int key[CCL_TILE];
    __shared__ int par[CCL_TILE];
    const int b = blockIdx.z, r0 = blockIdx.y * CCL_TH, c0 = blockIdx.x * CCL_TW;
    const int64_t fbase = (int64_t)b * H * W;
    if constexpr (std::is_same<KeyFn, KeyBits>::value) {
        // the tile's 32 rows are exactly one bit word per column, and a thread's pixels (i = tid + 256 k) all sit in
        // column tid % 64: one load instead of eight
        const int lc = threadIdx.x % CCL_TW, c = c0 + lc;
        const unsigned word = c < W ? keyfn.words[((int64_t)b * keyfn.nch + (r0 >> 5)) * W + c] : 0u;
        for (int i = threadIdx.x; i < CCL_TILE; i += 256) {
            const int lr = i / CCL_TW;
            key[i] = (r0 + lr < H) ? (int)((word >> lr) & 1u) : 0;
        }
    } else {
        for (int i = threadIdx.x; i < CCL_TILE; i += 256) {
            int r = r0 + i / CCL_TW, c = c0 + i % CCL_TW;
            key[i] = (r < H && c < W) ? keyfn(b, r, c) : 0;
        }
    }
    __syncthreads();
    ccl_tile_unions<CONN8>(key, par);
    ccl_tile_store(key, par, parent, fbase, r0, c0, H, W);
}

// Threads enumerate the tile-border pixels densely (a per-pixel grid would leave one or two working lanes per wave,
// each with a chain of dependent finds): first the rows that start a tile row, lanes along the row; then, for every
// other row, the columns either side of a vertical tile edge, lanes along the column.
template <typename KeyFn, bool CONN8>
__global__ void __launch_bounds__(256) ccl_border_kernel(KeyFn keyfn, int *__restrict__ parent, int H, int W, int *__restrict__ corrupt)
{
    const int n_top_rows = (H - 1) / CCL_TH, n_edges = (W - 1) / CCL_TW;
    const int n_cols = CONN8 ? 2 * n_edges : n_edges;
    int i = blockIdx.x * 256 + threadIdx.x;
    int r, c;
    if (i < n_top_rows * W) {
        r = (i / W + 1) * CCL_TH;
        c = i % W;
    } else {
        i -= n_top_rows * W;
        if (i >= n_cols * H) return;
        const int j = i / H;
        r = i % H;
        if ((r % CCL_TH) == 0 && r > 0) return;  // handled with its row above
        c = CONN8 ? ((j >> 1) + 1) * CCL_TW - (j & 1) : (j + 1) * CCL_TW;
    }
    const bool top = (r % CCL_TH) == 0 && r > 0;
    const bool left = (c % CCL_TW) == 0 && c > 0;
    const bool right = (c % CCL_TW) == CCL_TW - 1 && c + 1 < W;
    if (!top && !left && !(CONN8 && right && r > 0)) return;
    const int b = blockIdx.y;
    const int64_t fbase = (int64_t)b * H * W;
    int *par = parent + fbase;
    const int p = r * W + c;
    // the five keys as ONE batch of loads (clamped coordinates: every address is valid, what lies outside the frame is
    // masked afterwards) -- tested one after the other they were five dependent memory round trips per pixel
    // (candidate bits are sparse and their key costs two loads: there the pixel's own key is tested first)
    const int rn = max(r - 1, 0), cw = max(c - 1, 0), ce = min(c + 1, W - 1);
    const int k = keyfn(b, r, c);
    if (std::is_same<KeyFn, KeyCandBits>::value && k == 0) return;
    const int k_w = keyfn(b, r, cw), k_n = keyfn(b, rn, c), k_nw = keyfn(b, rn, cw), k_ne = CONN8 ? keyfn(b, rn, ce) : 0;
    if (k == 0) return;
    // the same "implied link" rule as inside a tile: a link is skipped when the two pixels are already joined through
    // a third one whose links are made elsewhere (run links inside a tile row, vertical links of the left neighbour)
    const bool w_same = c > 0 && k_w == k;
    const bool n_same = r > 0 && k_n == k;
    const bool nw_same = r > 0 && c > 0 && k_nw == k;
    // (at a tile corner both the W and the N link cross tiles and would justify each other: keep both there)
    const bool corner = top && left;
    int *bad = corrupt ? corrupt + b : nullptr;  // raised by a fenced walk (common.h, walk_ok)
    if (left && w_same && (corner || !(n_same && nw_same))) unite_glb(par, p, p - 1, bad);
    if (r > 0) {
        if (top && n_same && (corner || !(w_same && nw_same))) unite_glb(par, p, p - W, bad);
        if (CONN8) {
            if ((top || left) && nw_same && !n_same && !w_same) unite_glb(par, p, p - W - 1, bad);
            if (c + 1 < W && (top || right) && !n_same && k_ne == k) unite_glb(par, p, p - W + 1, bad);
        }
    }
}

// flatten + count roots accepted by `pred` per SCAN_PIX block
struct PredAll {
    __device__ __forceinline__ bool operator()(int64_t) const { return true; }
};
struct PredNotFlagged {  // root accepted unless flag[root] != 0 or its whole frame is switched off (frame_on[b] == 0)
    const uint8_t *flag;
    const int *frame_on;
    int64_t n;
    __device__ __forceinline__ bool operator()(int64_t gi) const { return flag[gi] == 0 && frame_on[gi / n] != 0; }
};

__device__ __forceinline__ int block_exclusive_scan(int v, int *total)
{
    // 256 threads = 4 waves; returns exclusive prefix of v, *total = block sum
    __shared__ int wsum[4];
    int lane = lane_id(), wid = threadIdx.x >> 6;
    int inc = v;
    for (int off = 1; off < 64; off <<= 1) {
        int t = __shfl_up(inc, off);
        if (lane >= off) inc += t;
    }
    if (lane == 63) wsum[wid] = inc;
    __syncthreads();
    int base = 0, tot = 0;
    for (int w = 0; w < 4; ++w) {
        if (w < wid) base += wsum[w];
        tot += wsum[w];
    }
    __syncthreads();
    *total = tot;
    return base + inc - v;
}

// Raster-order numbering = rank of the accepted roots.  Two pixel passes: (1) flatten, count the accepted roots of each
// SCAN_PIX block and leave -(rank inside the block) at every accepted root; (2) after the block totals were scanned,
// every pixel decodes its root's entry: negative = block offset - value, positive = a final label that the root's
// own thread has written meanwhile (both decode to the same number, so the in-place update needs no ordering).
template <typename Pred>
__global__ void __launch_bounds__(256) ccl_flatten_count_kernel(int *__restrict__ parent, int *__restrict__ labels,
                                                                 int *__restrict__ blockcount, Pred pred, int64_t n, int nblk,
                                                                 bool flatten, int *__restrict__ corrupt)
{
    const int b = blockIdx.y;
    int *par = parent + (int64_t)b * n;
    int cnt = 0;
    const int64_t i0 = (int64_t)blockIdx.x * SCAN_PIX + threadIdx.x * 4;
    int pv[4] = {-1, -1, -1, -1};
    if ((n & 3) == 0 && i0 + 3 < n) {
        const int4 v = *reinterpret_cast<const int4 *>(par + i0);
        pv[0] = v.x; pv[1] = v.y; pv[2] = v.z; pv[3] = v.w;
    } else {
        for (int j = 0; j < 4; ++j)
            if (i0 + j < n) pv[j] = par[i0 + j];
    }
    bool isroot[4];
    for (int j = 0; j < 4; ++j) {
        const int64_t i = i0 + j;
        int p = pv[j];
        isroot[j] = false;
        if (i >= n || p < 0) continue;
        if (!walk_ok((int)i, p)) {  // an entry above its own index (or past the frame): not a union-find image
            walk_corrupt(corrupt ? corrupt + b : nullptr);
            continue;
        }
        if (flatten) {
            const int x = walk_root(par, p, corrupt ? corrupt + b : nullptr);
            if (x != p) par[i] = x;
            p = x;
        }
        isroot[j] = p == (int)i && pred((int64_t)b * n + i);
        cnt += isroot[j];
    }
    int total;
    int rank = block_exclusive_scan(cnt, &total);
    for (int j = 0; j < 4; ++j)
        if (isroot[j]) labels[(int64_t)b * n + i0 + j] = -(++rank);
    if (threadIdx.x == 0) blockcount[b * nblk + blockIdx.x] = total;
}

// one block per frame: exclusive scan of the block totals, frame total -> counts
__global__ void __launch_bounds__(256) ccl_scan_blocks_kernel(int *__restrict__ blockcount, int *__restrict__ counts, int nblk,
                                                               const int *__restrict__ corrupt)
{
    int *bc = blockcount + (int64_t)blockIdx.x * nblk;
    int carry = 0;
    for (int base = 0; base < nblk; base += 256) {
        int i = base + threadIdx.x;
        int v = i < nblk ? bc[i] : 0;
        int total;
        int ex = block_exclusive_scan(v, &total);
        if (i < nblk) bc[i] = carry + ex;
        carry += total;
    }
    // a frame whose union-find image broke a walk's fence (common.h, walk_ok) reports -1 components
    if (threadIdx.x == 0 && counts) counts[blockIdx.x] = (corrupt && corrupt[blockIdx.x]) ? -1 : carry;
}

// chase: the parents are not flattened (the counting pass only looked for roots, parent[i] == i): every pixel walks to
// its root here -- one or two more dependent loads in a streaming, fully occupied kernel instead of a flatten pass
// that reads and rewrites the whole union-find image
template <typename Pred>
__global__ void __launch_bounds__(256) ccl_relabel_kernel(const int *__restrict__ parent, int *labels, const int *__restrict__ blockoff,
                                                           Pred pred, int64_t n, int nblk, bool chase, int *__restrict__ counts)
{
    const int b = blockIdx.y;
    const int *par = parent + (int64_t)b * n;
    int *lab = labels + (int64_t)b * n;
    const int64_t i0 = (int64_t)blockIdx.x * SCAN_PIX + threadIdx.x * 4;
    const bool vec = (n & 3) == 0 && i0 + 3 < n;
    int pv[4] = {-1, -1, -1, -1};
    if (vec) {
        const int4 v = *reinterpret_cast<const int4 *>(par + i0);
        pv[0] = v.x; pv[1] = v.y; pv[2] = v.z; pv[3] = v.w;
    } else {
        for (int j = 0; j < 4; ++j)
            if (i0 + j < n) pv[j] = par[i0 + j];
    }
    int out[4];
    int root_of_prev = -1, prev_p = -2;
    int bad = 0;  // a walk left its fence (common.h, walk_ok): the frame's count becomes -1, nothing is followed
    for (int j = 0; j < 4; ++j) {
        int p = pv[j];
        int v = 0;
        if (p >= 0 && !walk_ok((int)(i0 + j), p)) {
            bad = 1;
            p = -1;
        }
        if (chase && p >= 0) {
            if (p == prev_p) {
                p = root_of_prev;  // same entry as the pixel to the left: its walk is the answer
            } else {
                prev_p = p;
                p = walk_root(par, p, &bad);
                root_of_prev = p;
            }
        }
        if (p >= 0 && pred((int64_t)b * n + p)) {
            v = __hip_atomic_load(lab + p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);  // rank code or final label
            if (v < 0) v = blockoff[b * nblk + p / SCAN_PIX] - v;
        }
        out[j] = v;
    }
    if (vec) {
        *reinterpret_cast<int4 *>(lab + i0) = make_int4(out[0], out[1], out[2], out[3]);
    } else {
        for (int j = 0; j < 4; ++j)
            if (i0 + j < n) lab[i0 + j] = out[j];
    }
    if (bad && counts) counts[b] = -1;
}

// The same pass for n % 4 == 0, RELABEL_Q quads per lane (a block covers RELABEL_Q * SCAN_PIX pixels, quad q of a lane
// sits q * SCAN_PIX further on, so every access is still a coalesced 16 bytes per lane).  The pass is a chain of dependent
// gathers -- parent quad, one or two steps to the root, the root's rank code, its block offset -- and a lane that walks
// one chain at a time waits a full memory latency at each step (91 % of the kernel's wave cycles were waits).  Here the
// chains of a lane's RELABEL_Q quads advance in lockstep, so each step has RELABEL_Q independent loads in flight.  A quad's
// chain belongs to its first foreground pixel; the other three almost always carry the same parent entry and take its
// answer, one that does not walks on its own afterwards.
constexpr int RELABEL_Q = 4;

template <typename Pred>
__device__ __forceinline__ int relabel_decode(const int *par, int *lab, const int *blockoff, const Pred &pred, int b, int64_t n, int nblk,
                                              int p, bool chase, int *bad)
{
    if (chase) p = walk_root(par, p, bad);
    if (!pred((int64_t)b * n + p)) return 0;
    int v = __hip_atomic_load(lab + p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);  // rank code or final label
    if (v < 0) v = blockoff[b * nblk + p / SCAN_PIX] - v;
    return v;
}

// NCH chains in lockstep: root[q] (-1 = none) walks to its root, val[q] becomes the root's decoded label
template <typename Pred, int NCH>
__device__ __forceinline__ void relabel_chains(const int *par, int *lab, const int *blockoff, const Pred &pred, int b, int64_t n, int nblk,
                                               bool chase, int (&root)[NCH], int (&val)[NCH], int *bad)
{
    if (chase) {
        bool more = true;
        while (more) {
            int nx[NCH];
#pragma unroll
            for (int q = 0; q < NCH; ++q) nx[q] = root[q] >= 0 ? par[root[q]] : -1;
            // fenced (common.h, walk_ok), without branches: an entry above its index ends the chain where it stands and flags
            // the frame
            unsigned differ = 0, broke = 0;
#pragma unroll
            for (int q = 0; q < NCH; ++q) {
                const unsigned out = root[q] >= 0 && !walk_ok(root[q], nx[q]);
                broke |= out;
                nx[q] = out ? root[q] : nx[q];
                differ |= (unsigned)(nx[q] ^ root[q]);
                root[q] = nx[q];
            }
            more = differ != 0;
            if (broke) *bad = 1;
        }
    }
#pragma unroll
    for (int q = 0; q < NCH; ++q) {
        val[q] = 0;
        if (root[q] >= 0 && pred((int64_t)b * n + root[q]))
            val[q] = __hip_atomic_load(lab + root[q], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);  // rank code or final label
    }
    int off[NCH];
#pragma unroll
    for (int q = 0; q < NCH; ++q) off[q] = val[q] < 0 ? blockoff[b * nblk + root[q] / SCAN_PIX] : 0;
#pragma unroll
    for (int q = 0; q < NCH; ++q)
        if (val[q] < 0) val[q] = off[q] - val[q];
}

// (launch bounds: the pass is dependent gathers hidden by occupancy, and residency of 256-thread workgroups is bounded by
// SCALAR registers too -- min(8, 800 / (ceil(sgpr / 16) * 16 + 16)), MI355X_MICROARCH.md -- which the compiler only budgets when
// it is told the target: 106 scalar registers = six workgroups per CU without the second argument, 78 = eight with it)
template <typename Pred>
__global__ void __launch_bounds__(256, 8) ccl_relabel_quads_kernel(const int *__restrict__ parent, int *labels, const int *__restrict__ blockoff,
                                                                 Pred pred, int64_t n, int nblk, bool chase, int *__restrict__ counts)
{
    const int b = blockIdx.y;
    const int *par = parent + (int64_t)b * n;
    int *lab = labels + (int64_t)b * n;
    const int64_t i0 = (int64_t)blockIdx.x * (SCAN_PIX * RELABEL_Q) + threadIdx.x * 4;
    int4 pq[RELABEL_Q];
#pragma unroll
    for (int q = 0; q < RELABEL_Q; ++q) {
        const int64_t i = i0 + (int64_t)q * SCAN_PIX;
        pq[q] = i < n ? *reinterpret_cast<const int4 *>(par + i) : make_int4(-1, -1, -1, -1);
    }
    // the entries themselves are fenced first (common.h, walk_ok): an entry above the index it is stored at -- or past the
    // frame -- is not a union-find entry; it becomes background and the frame's count -1
    int bad = 0;
#pragma unroll
    for (int q = 0; q < RELABEL_Q; ++q) {  // (selects, not branches: sixteen `if`s in front of the chains cost the pass 15 us)
        const int i = (int)(i0 + (int64_t)q * SCAN_PIX);
        const int ox = pq[q].x >= 0 && !walk_ok(i, pq[q].x), oy = pq[q].y >= 0 && !walk_ok(i + 1, pq[q].y);
        const int oz = pq[q].z >= 0 && !walk_ok(i + 2, pq[q].z), ow = pq[q].w >= 0 && !walk_ok(i + 3, pq[q].w);
        bad |= ox | oy | oz | ow;
        pq[q].x = ox ? -1 : pq[q].x;
        pq[q].y = oy ? -1 : pq[q].y;
        pq[q].z = oz ? -1 : pq[q].z;
        pq[q].w = ow ? -1 : pq[q].w;
    }
    // First batch: the quads' first entries.  Neighbouring lanes mostly carry the same entry: only the first lane of each
    // run of equal entries walks, the others take its answer with one cross-lane read.
    // Second batch: a quad that straddles a component border holds a second entry.  Walked one pixel at a time these were
    // most of the pass (a copy of the two images takes 100 us, the pass without its gathers took 224): up to twelve
    // dependent chains per lane, taken by the whole wave whenever one of its 256 pixels sat at a border.
    // (all eight chains in ONE lockstep loop: 209 us against 182 for the two batches -- a wave without a border quad skips
    // the second batch altogether; the watershed's label pass, where borders are everywhere, is the other way round)
    int lead[RELABEL_Q], lead2[RELABEL_Q], root[RELABEL_Q], val[RELABEL_Q], val2[RELABEL_Q], head_lane[RELABEL_Q];
    bool third = false;  // a quad with three different entries: its odd pixels walk on their own below
    const int lane = lane_id();
#pragma unroll
    for (int q = 0; q < RELABEL_Q; ++q) {
        const int pv[4] = {pq[q].x, pq[q].y, pq[q].z, pq[q].w};
        lead[q] = pv[0] >= 0 ? pv[0] : (pv[1] >= 0 ? pv[1] : (pv[2] >= 0 ? pv[2] : pv[3]));
        const int left = __shfl_up(lead[q], 1);
        const bool head = lane == 0 || lead[q] != left;
        const unsigned long long heads = __ballot(head);
        head_lane[q] = 63 - __clzll((long long)(heads & (~0ull >> (63 - lane))));
        root[q] = head ? lead[q] : -1;
        lead2[q] = -1;
#pragma unroll
        for (int j = 1; j < 4; ++j) {
            if (pv[j] < 0 || pv[j] == lead[q]) continue;
            if (lead2[q] < 0) lead2[q] = pv[j];
            else if (pv[j] != lead2[q]) third = true;
        }
    }
    relabel_chains(par, lab, blockoff, pred, b, n, nblk, chase, root, val, &bad);
    bool any2 = false;
#pragma unroll
    for (int q = 0; q < RELABEL_Q; ++q) {
        val[q] = __shfl(val[q], head_lane[q]);
        root[q] = lead2[q];
        any2 = any2 || lead2[q] >= 0;
    }
    if (__any(any2)) relabel_chains(par, lab, blockoff, pred, b, n, nblk, chase, root, val2, &bad);
#pragma unroll
    for (int q = 0; q < RELABEL_Q; ++q) {
        const int64_t i = i0 + (int64_t)q * SCAN_PIX;
        if (i >= n) continue;
        const int pv[4] = {pq[q].x, pq[q].y, pq[q].z, pq[q].w};
        int out[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            out[j] = 0;
            if (pv[j] < 0) continue;
            if (pv[j] == lead[q]) out[j] = val[q];
            else if (pv[j] == lead2[q]) out[j] = val2[q];
            else if (third) out[j] = relabel_decode(par, lab, blockoff, pred, b, n, nblk, pv[j], chase, &bad);
        }
        *reinterpret_cast<int4 *>(lab + i) = make_int4(out[0], out[1], out[2], out[3]);
    }
    if (bad && counts) counts[b] = -1;
}

// ---- host-side drivers ---------------------------------------------------
struct CclWs {
    int *parent;
    int *blockcount;
    int nblk;
    int *corrupt;  // [B], cleared by ccl_roots / the callers that bring their own parents
};

static size_t ccl_ws_bytes(int B, int H, int W)
{
    int64_t n = (int64_t)H * W;
    int nblk = (int)((n + SCAN_PIX - 1) / SCAN_PIX);
    return align_up(sizeof(int) * (size_t)B * n) + align_up(sizeof(int) * (size_t)B * nblk) + align_up(sizeof(int) * (size_t)B);
}

static CclWs ccl_carve(Carver &cv, int B, int H, int W)
{
    int64_t n = (int64_t)H * W;
    CclWs ws;
    ws.nblk = (int)((n + SCAN_PIX - 1) / SCAN_PIX);
    ws.parent = cv.take<int>((size_t)B * n);
    ws.blockcount = cv.take<int>((size_t)B * ws.nblk);
    ws.corrupt = cv.take<int>((size_t)B);
    return ws;
}

template <typename KeyFn, bool CONN8>
static int ccl_roots(KeyFn keyfn, int *parent, int B, int H, int W, hipStream_t s, bool tile_pass_done = false, int *corrupt = nullptr)
{
    // `corrupt` ([B], may be null): cleared here, raised by a fenced walk of the border pass or of a later pass (walk_ok)
    if (corrupt) PCSEG_CHECK_HIP(hipMemsetAsync(corrupt, 0, sizeof(int) * (size_t)B, s));
    dim3 tgrid((W + CCL_TW - 1) / CCL_TW, (H + CCL_TH - 1) / CCL_TH, B);
    if (!tile_pass_done) {
        PCSEG_LAUNCH((ccl_tile_kernel<KeyFn, CONN8>), tgrid, dim3(256), 0, s, keyfn, parent, H, W);
        PCSEG_CHECK_LAUNCH();
    }
    if (tgrid.x > 1 || tgrid.y > 1) {
        const int64_t border_px = (int64_t)((H - 1) / CCL_TH) * W + (int64_t)(CONN8 ? 2 : 1) * ((W - 1) / CCL_TW) * H;
        dim3 bgrid((unsigned)((border_px + 255) / 256), B);
        PCSEG_LAUNCH((ccl_border_kernel<KeyFn, CONN8>), bgrid, dim3(256), 0, s, keyfn, parent, H, W, corrupt);
        PCSEG_CHECK_LAUNCH();
    }
    return PCSEG_OK;
}

// parent must hold roots that are NOT yet flattened when flatten == true
template <typename Pred>
static int ccl_compact(int *parent, int *blockcount, int nblk, int *labels, int *counts, Pred pred, bool flatten,
                       int B, int H, int W, hipStream_t s, int *corrupt = nullptr)
{
    int64_t n = (int64_t)H * W;
    dim3 grid(nblk, B);
    // roots are exact after the border pass (parent[i] == i), so they can be counted and ranked without flattening;
    // `flatten` (the parents are not flat yet) only tells the relabel pass to walk to the roots itself
    PCSEG_LAUNCH((ccl_flatten_count_kernel<Pred>), grid, dim3(256), 0, s, parent, labels, blockcount, pred, n, nblk, false, corrupt);
    PCSEG_CHECK_LAUNCH();
    PCSEG_LAUNCH(ccl_scan_blocks_kernel, dim3(B), dim3(256), 0, s, blockcount, counts, nblk, (const int *)corrupt);
    PCSEG_CHECK_LAUNCH();
    if ((n & 3) == 0 && (((uintptr_t)parent | (uintptr_t)labels) & 15) == 0) {
        const dim3 qgrid((unsigned)((n + SCAN_PIX * RELABEL_Q - 1) / (SCAN_PIX * RELABEL_Q)), B);
        PCSEG_LAUNCH((ccl_relabel_quads_kernel<Pred>), qgrid, dim3(256), 0, s, (const int *)parent, labels, (const int *)blockcount, pred, n,
                     nblk, flatten, counts);
    } else {
        PCSEG_LAUNCH((ccl_relabel_kernel<Pred>), grid, dim3(256), 0, s, (const int *)parent, labels, (const int *)blockcount, pred, n, nblk,
                     flatten, counts);
    }
    PCSEG_CHECK_LAUNCH();
    return PCSEG_OK;
}

template <typename KeyFn, bool CONN8>
static int ccl_full(KeyFn keyfn, int32_t *labels, int32_t *counts, int B, int H, int W, void *workspace,
                    size_t workspace_bytes, hipStream_t s)
{
    Carver cv(workspace, workspace_bytes);
    CclWs ws = ccl_carve(cv, B, H, W);
    if (!cv.ok()) {
        set_error("ccl: workspace too small (%zu < %zu)", workspace_bytes, cv.off);
        return PCSEG_ERR_WORKSPACE;
    }
    int rc = ccl_roots<KeyFn, CONN8>(keyfn, ws.parent, B, H, W, s, false, ws.corrupt);
    if (rc) return rc;
    return ccl_compact(ws.parent, ws.blockcount, ws.nblk, labels, counts, PredAll(), true, B, H, W, s, ws.corrupt);
}

int ccl_plan(void *workspace, size_t workspace_bytes, int B, int H, int W, CclPlan *plan, const char *who)
{
    Carver cv(workspace, workspace_bytes);
    CclWs ws = ccl_carve(cv, B, H, W);
    if (!cv.ok()) {
        set_error("%s: workspace too small (%zu < %zu)", who, workspace_bytes, cv.off);
        return PCSEG_ERR_WORKSPACE;
    }
    plan->parent = ws.parent;
    plan->blockcount = ws.blockcount;
    plan->nblk = ws.nblk;
    plan->corrupt = ws.corrupt;
    return PCSEG_OK;
}

int ccl_equal_u8_finish(const uint8_t *in, const CclPlan &plan, bool tile_pass_done, int *labels, int *counts, int B, int H, int W,
                        hipStream_t s)
{
    const KeyEqU8 keyfn{in, W, (int64_t)H * W};
    PCSEG_CHECK_HIP(hipMemsetAsync(plan.corrupt, 0, sizeof(int) * (size_t)B, s));
    dim3 tgrid((W + CCL_TW - 1) / CCL_TW, (H + CCL_TH - 1) / CCL_TH, B);
    if (!tile_pass_done) {
        PCSEG_LAUNCH((ccl_tile_kernel<KeyEqU8, true>), tgrid, dim3(256), 0, s, keyfn, plan.parent, H, W);
        PCSEG_CHECK_LAUNCH();
    }
    if (tgrid.x > 1 || tgrid.y > 1) {
        const int64_t border_px = (int64_t)((H - 1) / CCL_TH) * W + (int64_t)2 * ((W - 1) / CCL_TW) * H;
        dim3 bgrid((unsigned)((border_px + 255) / 256), B);
        PCSEG_LAUNCH((ccl_border_kernel<KeyEqU8, true>), bgrid, dim3(256), 0, s, keyfn, plan.parent, H, W, plan.corrupt);
        PCSEG_CHECK_LAUNCH();
    }
    return ccl_compact(plan.parent, plan.blockcount, plan.nblk, labels, counts, PredAll(), true, B, H, W, s, plan.corrupt);
}

// ---- roots(+1) image -> parent(-1 bg) conversion for pcseg_compact_labels
__global__ void __launch_bounds__(256) roots_to_parent_kernel(const int *__restrict__ roots, int *__restrict__ parent, int64_t total)
{
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < total) parent[i] = roots[i] - 1;
}

// ---- disk(r) dilation on the 1-bit image + components of the result, roots only (A6) -------------------------
// set bits: 1 where ((value_bits >> in) & 1), rows beyond H are 0
__global__ void __launch_bounds__(256) set_bits_kernel(const uint8_t *__restrict__ in, unsigned long long value_bits,
                                                        unsigned *__restrict__ bits, int H, int W, int nch)
{
    const int c = blockIdx.x * 256 + threadIdx.x;
    const int ch = blockIdx.y, b = blockIdx.z;
    if (c >= W) return;
    const uint8_t *src = in + (int64_t)b * H * W;
    unsigned word = 0;
#pragma unroll 8
    for (int j = 0; j < 32; ++j) {
        int r = ch * 32 + j;
        if (r < H) {
            unsigned v = src[rowoff(r, W) + c];
            if (v < 64 && ((value_bits >> v) & 1ull)) word |= 1u << j;
        }
    }
    bits[((int64_t)b * nch + ch) * W + c] = word;
}

// the same for W % 4 == 0 and a 4-byte aligned input: a lane takes four adjacent columns (one 4-byte load per row)
__global__ void __launch_bounds__(256) set_bits4_kernel(const uint8_t *__restrict__ in, unsigned long long value_bits,
                                                         unsigned *__restrict__ bits, int H, int W, int nch)
{
    const int c = (blockIdx.x * 256 + threadIdx.x) * 4;
    const int ch = blockIdx.y, b = blockIdx.z;
    if (c >= W) return;
    const uint8_t *src = in + (int64_t)b * H * W + c;
    unsigned w0 = 0, w1 = 0, w2 = 0, w3 = 0;
#pragma unroll 8
    for (int j = 0; j < 32; ++j) {
        const int r = ch * 32 + j;
        if (r < H) {
            const unsigned v = *reinterpret_cast<const unsigned *>(src + rowoff(r, W));
            const unsigned a = v & 255u, bb = (v >> 8) & 255u, cc = (v >> 16) & 255u, d = v >> 24;
            if (a < 64 && ((value_bits >> a) & 1ull)) w0 |= 1u << j;
            if (bb < 64 && ((value_bits >> bb) & 1ull)) w1 |= 1u << j;
            if (cc < 64 && ((value_bits >> cc) & 1ull)) w2 |= 1u << j;
            if (d < 64 && ((value_bits >> d) & 1ull)) w3 |= 1u << j;
        }
    }
    *reinterpret_cast<uint4 *>(bits + ((int64_t)b * nch + ch) * W + c) = make_uint4(w0, w1, w2, w3);
}

// several masks of one class map at once (the proximity merge wants one mask per cell type plus the union of all types):
// the map is read ONCE, mask m goes to bits + m * (B * nch * W) -- behind it the dilation / run passes simply see
// n_masks * B frames
struct MaskSet {
    unsigned long long bits[4];
    int n;
};
__global__ void __launch_bounds__(256) set_bits4_multi_kernel(const uint8_t *__restrict__ in, MaskSet masks, unsigned *__restrict__ bits,
                                                               int H, int W, int nch, int B)
{
    const int c = (blockIdx.x * 256 + threadIdx.x) * 4;
    const int ch = blockIdx.y, b = blockIdx.z;
    if (c >= W) return;
    const uint8_t *src = in + (int64_t)b * H * W + c;
    unsigned w[4][4];
#pragma unroll
    for (int m = 0; m < 4; ++m)
#pragma unroll
        for (int j = 0; j < 4; ++j) w[m][j] = 0;
    // eight rows' loads in flight at a time (a row test around each load made them 32 dependent round trips): rows past
    // the frame's end re-read its last row and are masked out of the words
#pragma unroll 1
    for (int j0 = 0; j0 < 32; j0 += 8) {
        unsigned v8[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) v8[j] = *reinterpret_cast<const unsigned *>(src + rowoff(min(ch * 32 + j0 + j, H - 1), W));
#pragma unroll
        for (int j = 0; j < 8; ++j) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const unsigned a = (v8[j] >> (8 * q)) & 255u;
                const unsigned long long sel = a < 64 ? (1ull << a) : 0ull;
#pragma unroll
                for (int m = 0; m < 4; ++m)
                    if (m < masks.n && (masks.bits[m] & sel)) w[m][q] |= 1u << (j0 + j);
            }
        }
    }
    {
        const int rows = min(32, H - ch * 32);
        const unsigned valid = rows >= 32 ? 0xFFFFFFFFu : ((1u << rows) - 1u);
#pragma unroll
        for (int m = 0; m < 4; ++m)
#pragma unroll
            for (int q = 0; q < 4; ++q) w[m][q] &= valid;
    }
    const int64_t plane = (int64_t)B * nch * W;
#pragma unroll
    for (int m = 0; m < 4; ++m)
        if (m < masks.n)
            *reinterpret_cast<uint4 *>(bits + m * plane + ((int64_t)b * nch + ch) * W + c) = make_uint4(w[m][0], w[m][1], w[m][2], w[m][3]);
}

// out = dilate(in, disk(radius)): for every row offset dy the columns within half(dy) = floor(sqrt(r^2 - dy^2)) are
// OR-ed, then shifted by dy rows across the 32-row words (skimage disk: x^2 + y^2 <= r^2; outside the image = 0)
__global__ void __launch_bounds__(256) dilate_bits_kernel(const unsigned *__restrict__ in, unsigned *__restrict__ out,
                                                           int radius, int H, int W, int nch)
{
    const int c = blockIdx.x * 256 + threadIdx.x;
    const int ch = blockIdx.y, b = blockIdx.z;
    if (c >= W) return;
    const unsigned *base = in + (int64_t)b * nch * W;
    unsigned acc = 0;
    for (int dy = 0; dy <= radius; ++dy) {
        int half = 0;
        while ((half + 1) * (half + 1) + dy * dy <= radius * radius) ++half;
        unsigned cur = 0, prev = 0, next = 0;  // OR over the columns c-half..c+half of this word / the word above / below
        for (int dx = -half; dx <= half; ++dx) {
            int cc = c + dx;
            if (cc < 0 || cc >= W) continue;
            cur |= base[rowoff(ch, W) + cc];
            if (dy > 0) {
                if (ch > 0) prev |= base[rowoff((ch - 1), W) + cc];
                if (ch + 1 < nch) next |= base[rowoff((ch + 1), W) + cc];
            }
        }
        if (dy == 0) acc |= cur;
        else {
            acc |= (cur << dy) | (prev >> (32 - dy));  // source rows above move down by dy
            acc |= (cur >> dy) | (next << (32 - dy));  // source rows below move up by dy
        }
    }
    const int rows = min(32, H - ch * 32);
    if (rows < 32) acc &= (1u << rows) - 1u;
    out[((int64_t)b * nch + ch) * W + c] = acc;
}

// disk(2) -- the only radius the reference dilates its masks with (tiff_analysis.py:829) -- with its eleven words
// (columns c-2..c+2 of the row word, c-1..c+1 of the words above and below) as ONE batch of loads at clamped addresses;
// the general kernel above tests every load and works the half widths out with a loop of multiplications per row offset
// (59 us a launch for 6 M words).  Offsets: dy = 0: |dx| <= 2, |dy| = 1: |dx| <= 1, |dy| = 2: dx = 0.
__global__ void __launch_bounds__(256) dilate_bits_disk2_kernel(const unsigned *__restrict__ in, unsigned *__restrict__ out, int H, int W,
                                                                 int nch)
{
    const int c = blockIdx.x * 256 + threadIdx.x;
    const int ch = blockIdx.y, b = blockIdx.z;
    if (c >= W) return;
    const unsigned *base = in + (int64_t)b * nch * W;
    const int chu = max(ch - 1, 0), chd = min(ch + 1, nch - 1);
    const unsigned *rc = base + rowoff(ch, W), *ru = base + rowoff(chu, W), *rd = base + rowoff(chd, W);
    const int c1l = max(c - 1, 0), c2l = max(c - 2, 0), c1r = min(c + 1, W - 1), c2r = min(c + 2, W - 1);
    // (a clamped column repeats a word that is OR-ed in anyway: the OR does not change)
    const unsigned w0 = rc[c], w1l = rc[c1l], w1r = rc[c1r], w2l = rc[c2l], w2r = rc[c2r];
    unsigned u0 = ru[c], u1l = ru[c1l], u1r = ru[c1r], d0 = rd[c], d1l = rd[c1l], d1r = rd[c1r];
    if (ch == 0) u0 = u1l = u1r = 0;          // outside the image = 0
    if (ch + 1 >= nch) d0 = d1l = d1r = 0;
    const unsigned cur1 = w0 | w1l | w1r, prev1 = u0 | u1l | u1r, next1 = d0 | d1l | d1r;
    unsigned acc = cur1 | w2l | w2r;
    acc |= (cur1 << 1) | (prev1 >> 31) | (cur1 >> 1) | (next1 << 31);
    acc |= (w0 << 2) | (u0 >> 30) | (w0 >> 2) | (d0 << 30);
    const int rows = min(32, H - ch * 32);
    if (rows < 32) acc &= (1u << rows) - 1u;
    out[((int64_t)b * nch + ch) * W + c] = acc;
}

// ---- components of a 1-bit image from its vertical runs (A6: label(dilated mask), tiff_analysis.py:829) ----------
// The dilated masks of the merge step are only ever LOOKED UP at a few hundred centroid pixels per frame, so no label
// image is made: the nodes of the union-find are the vertical runs of set bits (a run never leaves its 32-row word; it
// is named by its top pixel; runs are a few per cent of the pixel count) and only their entries of the pixel-indexed
// parent array are ever touched.  Two passes: an LDS union-find per tile of 64 columns x 128 rows (four words per
// column), then the links that cross a tile edge with agent-scope atomics.  Links (8-connectivity), always seen from
// the right / lower pixel: a run to the runs of the column to its left that overlap rows [top - 1, bottom + 1]; a
// run that starts in row 0 of its word to the run that ends in row 31 of the word above -- or, if that pixel is not
// set, to the two diagonal ones (with the vertical link present the diagonals are implied by the left links of the row
// above).
constexpr int BR_TW = 64, BR_CH = 4, BR_ROWS = 32 * BR_CH;

__device__ __forceinline__ int bitrun_start(unsigned word, int p)  // first row of the run of `word` that contains bit p
{
    const unsigned below = ~word & ((1u << p) - 1u);
    return below ? 32 - __clz(below) : 0;
}

// calls link(top row of a run of w, top row of a run of wl) for every pair of runs that touch across the column edge
template <typename Link>
__device__ __forceinline__ void bitrun_left_links(unsigned w, unsigned wl, Link &&link)
{
    unsigned rest = wl ? w : 0u;
    while (rest) {
        const int a = __ffs(rest) - 1;  // top of this run of w
        const unsigned from_a = w >> a;
        const int len = from_a == 0xFFFFFFFFu ? 32 : __ffs(~from_a) - 1;
        const int e = a + len - 1;  // bottom
        rest = e >= 31 ? 0u : rest & ~((2u << e) - 1u);
        const int lo = max(a - 1, 0), hi = min(e + 1, 31);
        unsigned m = wl & (hi == 31 ? 0xFFFFFFFFu : ((2u << hi) - 1u)) & ~((1u << lo) - 1u);
        while (m) {
            const int p = __ffs(m) - 1;
            const int st = bitrun_start(wl, p);
            const unsigned from_p = wl >> p;
            const int run_len = from_p == 0xFFFFFFFFu ? 32 : __ffs(~from_p) - 1;
            const int en = p + run_len - 1;
            m = en >= 31 ? 0u : m & ~((2u << en) - 1u);
            link(a, st);
        }
    }
}

__global__ void __launch_bounds__(256) bitrun_tile_kernel(const unsigned *__restrict__ bits, int *__restrict__ parent, int H, int W,
                                                           int nch)
{
    __shared__ unsigned sw[BR_CH][BR_TW];
    __shared__ int lpar[BR_ROWS * BR_TW];  // node = (row inside the tile) * 64 + column; only run tops are used
    const int col = threadIdx.x & (BR_TW - 1), q = threadIdx.x >> 6;
    const int c0 = blockIdx.x * BR_TW, ch0 = blockIdx.y * BR_CH, b = blockIdx.z;
    const int c = c0 + col, ch = ch0 + q;
    const unsigned w = (c < W && ch < nch) ? bits[((int64_t)b * nch + ch) * W + c] : 0u;
    sw[q][col] = w;
    for (unsigned heads = w & ~(w << 1); heads; heads &= heads - 1) {
        const int node = (q * 32 + (__ffs(heads) - 1)) * BR_TW + col;
        lpar[node] = node;
    }
    __syncthreads();
    if (w) {
        if (col > 0)
            bitrun_left_links(w, sw[q][col - 1], [&](int a, int st) {
                unite_lds(lpar, (q * 32 + a) * BR_TW + col, (q * 32 + st) * BR_TW + col - 1);
            });
        if ((w & 1u) && q > 0) {
            const int node = (q * 32) * BR_TW + col;
            const unsigned wu = sw[q - 1][col];
            if (wu >> 31) {
                unite_lds(lpar, node, ((q - 1) * 32 + bitrun_start(wu, 31)) * BR_TW + col);
            } else {
                if (col > 0 && (sw[q - 1][col - 1] >> 31))
                    unite_lds(lpar, node, ((q - 1) * 32 + bitrun_start(sw[q - 1][col - 1], 31)) * BR_TW + col - 1);
                if (col + 1 < BR_TW && (sw[q - 1][col + 1] >> 31))
                    unite_lds(lpar, node, ((q - 1) * 32 + bitrun_start(sw[q - 1][col + 1], 31)) * BR_TW + col + 1);
            }
        }
    }
    __syncthreads();
    int *par = parent + (int64_t)b * H * W;
    for (unsigned heads = w & ~(w << 1); heads; heads &= heads - 1) {
        const int node = (q * 32 + (__ffs(heads) - 1)) * BR_TW + col;
        const int root = find_lds(lpar, node);
        par[(ch0 * 32 + node / BR_TW) * W + c0 + node % BR_TW] = (ch0 * 32 + root / BR_TW) * W + c0 + root % BR_TW;
    }
}

// links that cross a tile edge: the left links of a tile's first column, the upward links of a tile's first word
// (all three columns c - 1, c, c + 1 of the word above lie in other tiles' rows), and the diagonal upward links of the
// first / last column of a tile whose neighbour column lies in the tile to the left / right
__global__ void __launch_bounds__(256) bitrun_border_kernel(const unsigned *__restrict__ bits, int *__restrict__ parent, int H, int W,
                                                             int nch)
{
    const int c = blockIdx.x * 256 + threadIdx.x;
    const int ch = blockIdx.y, b = blockIdx.z;
    if (c >= W) return;
    const int col = c & (BR_TW - 1);
    const bool tile_left = col == 0 && c > 0, tile_top = (ch % BR_CH) == 0 && ch > 0;
    const bool edge_col = col == 0 || col == BR_TW - 1;
    if (!tile_left && !tile_top && !(edge_col && ch > 0)) return;
    const unsigned *wb = bits + (int64_t)b * nch * W;
    // the five words as one batch of loads (clamped at the frame's edge, where the value is not used) instead of up to
    // four dependent round trips
    const int cl = c > 0 ? c - 1 : c, cr = c + 1 < W ? c + 1 : c, chu = ch > 0 ? ch - 1 : ch;
    const unsigned w = wb[rowoff(ch, W) + c], wl = wb[rowoff(ch, W) + cl];
    const unsigned wu = wb[rowoff(chu, W) + c], wul = wb[rowoff(chu, W) + cl], wur = wb[rowoff(chu, W) + cr];
    if (w == 0) return;
    int *par = parent + (int64_t)b * H * W;
    const int row0 = ch * 32;
    if (tile_left)
        bitrun_left_links(w, wl, [&](int a, int st) { unite_glb(par, (row0 + a) * W + c, (row0 + st) * W + c - 1); });
    if ((w & 1u) && ch > 0) {
        const int node = row0 * W + c;
        if (wu >> 31) {
            if (tile_top) unite_glb(par, node, (row0 - 32 + bitrun_start(wu, 31)) * W + c);
        } else {
            // diagonals: made in LDS unless the word above is in another tile row, or the neighbour column in another tile
            if (c > 0 && (tile_top || col == 0) && (wul >> 31)) unite_glb(par, node, (row0 - 32 + bitrun_start(wul, 31)) * W + c - 1);
            if (c + 1 < W && (tile_top || col == BR_TW - 1) && (wur >> 31))
                unite_glb(par, node, (row0 - 32 + bitrun_start(wur, 31)) * W + c + 1);
        }
    }
}

// ---- fill holes ------------------------------------------------------------
__global__ void __launch_bounds__(256) border_flag_kernel(const int *__restrict__ parent, uint8_t *__restrict__ flag, int H, int W)
{
    // one thread per border pixel: t in [0, 2W + 2H)
    int t = blockIdx.x * blockDim.x + threadIdx.x;
    int r, c;
    if (t < W) { r = 0; c = t; }
    else if (t < 2 * W) { r = H - 1; c = t - W; }
    else if (t < 2 * W + H) { r = t - 2 * W; c = 0; }
    else if (t < 2 * W + 2 * H) { r = t - 2 * W - H; c = W - 1; }
    else return;
    int64_t fbase = (int64_t)blockIdx.y * H * W;
    int p = parent[fbase + rowoff(r, W) + c];
    if (p >= 0) flag[fbase + p] = 1;
}

__global__ void __launch_bounds__(256) fill_holes_out_kernel(const uint8_t *__restrict__ mask, const int *__restrict__ parent,
                                                              const uint8_t *__restrict__ flag, uint8_t *__restrict__ out, int64_t n)
{
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    int64_t fbase = (int64_t)blockIdx.y * n;
    int p = parent[fbase + i];
    out[fbase + i] = (mask[fbase + i] != 0) || (p >= 0 && flag[fbase + p] == 0);
}

// plain flatten (no counting)
__global__ void __launch_bounds__(256) ccl_flatten_kernel(int *__restrict__ parent, int64_t n)
{
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    int *par = parent + (int64_t)blockIdx.y * n;
    int p = par[i];
    if (p < 0 || !walk_ok((int)i, p)) return;
    const int x = walk_root(par, p);  // fenced (common.h, walk_ok)
    if (x != p) par[i] = x;
}

// ---- local maxima ----------------------------------------------------------
// One stencil pass over a 64x32 tile staged in LDS with a 2-pixel halo: candidates (no higher 8-neighbour) are
// determined for the tile and the ring around it, so the same pass can also tell which candidate pixels touch an
// equal-valued NON-candidate -- the pixels that spoil their plateau.  That flag is left at the pixel's own slot of
// `bad`; after the components of the candidates are known, locmax_propagate_kernel flattens the parents and raises
// bad[root] for every flagged pixel (own-slot flags of non-root pixels are never read as root flags).
constexpr int LM_TW = CCL_TW, LM_TH = CCL_TH, LM_SW = LM_TW + 4, LM_SH = LM_TH + 4, LM_CW = LM_TW + 2, LM_CH = LM_TH + 2;

// ... and the union-find tile pass of the candidates' plateaus (equal-valued 8-connected candidates) runs in the same
// kernel on the keys while they are still in LDS: the key image is written for the border pass but never read back
// by a tile pass of its own.
// What leaves the kernel is SPARSE: candidates are a few per cent of a distance map, so instead of a key image, a flag
// image and a parent image (9 bytes per pixel, each read back by one or two later passes) the kernel writes one bit per
// pixel (flat bit array, see cand_bit) and the parent / flag entries of the candidate pixels only; every later pass
// (border links, flag propagation, root count, relabel) walks the bit words and touches candidates only.  With
// `markers` the kernel also zero-fills the marker image at the non-candidates: the relabel pass then writes the
// candidates, and every pixel of the output is written exactly once.
__global__ void __launch_bounds__(256) locmax_candidates_kernel(const int *__restrict__ img, unsigned long long *__restrict__ cbits,
                                                                 uint8_t *__restrict__ bad, int *__restrict__ nonconst,
                                                                 int *__restrict__ parent, int *__restrict__ markers, int H, int W)
{
    __shared__ int tile[LM_SH * LM_SW];
    __shared__ uint8_t cand[LM_CH * LM_CW];  // tile + 1 ring: 0 inside and not a candidate, 1 / 3 candidate (bit 0), 2 outside the image
    __shared__ int key[CCL_TILE];
    static_assert(LM_SH * LM_SW >= CCL_TILE, "the parents reuse the value tile");
    int *par = tile;  // the values are dead once the keys are out (20 KB per block instead of 28: 8 blocks per CU)
    const int OUTSIDE = (int)0x80000000;  // image values are > INT_MIN by contract: never higher, never "different"
    const TileIndex ti = xcd_tile_index();
    const int r0 = ti.y * LM_TH, c0 = ti.x * LM_TW;
    const int64_t fbase = (int64_t)ti.z * H * W;
    if ((W & 3) == 0 && c0 + LM_TW <= W && (((uintptr_t)img) & 15) == 0) {
        // full-width tile of a frame with 16-byte aligned rows: the 64 interior columns of the 36 rows as 16-byte quads
        // (three trips), the four halo columns as one scalar trip -- all of them one batch of loads at clamped addresses
        constexpr int QUADS = LM_SH * (LM_TW / 4), TRIPS = (QUADS + 255) / 256;
        int4 tq[TRIPS];
#pragma unroll
        for (int t = 0; t < TRIPS; ++t) {
            const int i = min((int)threadIdx.x + 256 * t, QUADS - 1);
            const int r = r0 + i / (LM_TW / 4) - 2;
            tq[t] = *reinterpret_cast<const int4 *>(img + fbase + rowoff(min(max(r, 0), H - 1), W) + c0 + 4 * (i % (LM_TW / 4)));
        }
        const int hl = min((int)threadIdx.x, LM_SH * 4 - 1);
        const int hr = r0 + (hl >> 2) - 2, hx = (hl & 3) < 2 ? (hl & 3) - 2 : LM_TW + (hl & 3) - 2, hc = c0 + hx;
        const int hv = img[fbase + rowoff(min(max(hr, 0), H - 1), W) + min(max(hc, 0), W - 1)];
#pragma unroll
        for (int t = 0; t < TRIPS; ++t) {
            const int i = (int)threadIdx.x + 256 * t;
            if (i < QUADS) {
                const int lr = i / (LM_TW / 4), r = r0 + lr - 2;
                const bool in = r >= 0 && r < H;
                int *dst = tile + lr * LM_SW + 2 + 4 * (i % (LM_TW / 4));
                dst[0] = in ? tq[t].x : OUTSIDE; dst[1] = in ? tq[t].y : OUTSIDE; dst[2] = in ? tq[t].z : OUTSIDE; dst[3] = in ? tq[t].w : OUTSIDE;
            }
        }
        if (threadIdx.x < LM_SH * 4) tile[(hl >> 2) * LM_SW + hx + 2] = (hr >= 0 && hr < H && hc >= 0 && hc < W) ? hv : OUTSIDE;
    } else {
        // one batch of loads instead of a loop of round trips (clamped addresses, no branch around the loads)
        constexpr int TRIPS = (LM_SH * LM_SW + 255) / 256;
        int tv[TRIPS];
#pragma unroll
        for (int t = 0; t < TRIPS; ++t) {
            const int i = min((int)threadIdx.x + 256 * t, LM_SH * LM_SW - 1);
            const int r = r0 + i / LM_SW - 2, c = c0 + i % LM_SW - 2;
            tv[t] = img[fbase + rowoff(min(max(r, 0), H - 1), W) + min(max(c, 0), W - 1)];
        }
#pragma unroll
        for (int t = 0; t < TRIPS; ++t) {
            const int i = (int)threadIdx.x + 256 * t;
            const int r = r0 + i / LM_SW - 2, c = c0 + i % LM_SW - 2;
            if (i < LM_SH * LM_SW) tile[i] = (r >= 0 && r < H && c >= 0 && c < W) ? tv[t] : OUTSIDE;
        }
    }
    __syncthreads();
    // A pixel is a candidate when none of its 8 neighbours is higher, i.e. when the maximum of the eight does not exceed it:
    // three v_max3 and one compare instead of eight compares and their ands (the kernel is VALU-bound: 13 % VALU-active per
    // wave at eight waves per SIMD).  "The frame is not constant" (a constant image has no maxima, extrema.py:388-391) used to
    // be collected from the same eight neighbours; on a connected grid it is the same as "some pixel differs from the frame's
    // first pixel": one compare per pixel.
    const int first_value = img[fbase];
    bool any_differs = false;
    for (int t = threadIdx.x; t < LM_CH * LM_CW; t += 256) {
        const int lr = t / LM_CW, lc = t % LM_CW;  // ring coordinates: image pixel (r0 + lr - 1, c0 + lc - 1)
        const int r = r0 + lr - 1, c = c0 + lc - 1;
        const int i = (lr + 1) * LM_SW + lc + 1;
        const int v = tile[i];
        uint8_t state = 2;
        if (r >= 0 && r < H && c >= 0 && c < W) {
            const int *up = tile + i - LM_SW, *dn = tile + i + LM_SW;
#if defined(PCSEG_EXP_LOCMAX) && (PCSEG_EXP_LOCMAX & 4)
            const int m = max(up[0], dn[0]) + 40;
#else
            const int m = max(max(max(up[-1], up[0]), max(up[1], tile[i - 1])), max(max(tile[i + 1], dn[-1]), max(dn[0], dn[1])));
#endif
            // 1: higher than all eight (nothing equal around it: neither a spoilt plateau nor a link), 3: a plateau pixel
            state = m < v ? 1 : (m == v ? 3 : 0);  // (OUTSIDE = INT_MIN is never higher)
            any_differs = any_differs || v != first_value;
        }
        cand[t] = state;
    }
    __syncthreads();
    // Equal-valued neighbours only exist around PLATEAU candidates (state 3), a small part of the few per cent of pixels that
    // are candidates at all: only they look at their eight neighbours -- for an equal-valued NON-candidate (the plateau is
    // spoilt) and for equal-valued candidates among the four neighbours that precede them (W, NW, N, NE: the links of the
    // tile's union-find, one bit each, kept in a register until the value tile has become the parent array).
    unsigned links = 0;  // 4 bits per pixel of this thread
#pragma unroll
    for (int kk = 0; kk < LM_TH * LM_TW / 256; ++kk) {
        const int t = threadIdx.x + 256 * kk;
        const int lr = t / LM_TW, lc = t % LM_TW;
        const int r = r0 + lr, c = c0 + lc;
        int k = 0;
        if (r < H && c < W) {
            const int i = (lr + 2) * LM_SW + lc + 2, j = (lr + 1) * LM_CW + lc + 1;
            const int v = tile[i];
            const uint8_t st = cand[j];
            const bool is_cand = (st & 1) != 0;
            bool touches = false;
#if defined(PCSEG_EXP_LOCMAX) && (PCSEG_EXP_LOCMAX & 1)  // (ablation builds, profiles/r04/time_ops.py locmax: the pass without a phase)
            if (false) {
#else
            if (st == 3) {
#endif
#pragma unroll
                for (int dr = -1; dr <= 1; ++dr)
#pragma unroll
                    for (int dc = -1; dc <= 1; ++dc) {
                        if (dr == 0 && dc == 0) continue;
                        const bool eq = tile[i + dr * LM_SW + dc] == v;
                        const uint8_t sn = cand[j + dr * LM_CW + dc];
                        touches = touches || (eq && sn == 0);
                        // a link to a PRECEDING neighbour inside the tile (the border pass makes the links across tiles)
                        if ((dr < 0 || (dr == 0 && dc < 0)) && eq && (sn & 1) && lr + dr >= 0 && lc + dc >= 0 && lc + dc < LM_TW)
                            links |= 1u << (4 * kk + (dr == 0 ? 0 : dc + 2));  // bit 0: W, 1: NW, 2: N, 3: NE
                    }
            }
            // key must be non-zero for candidates and equal exactly when the values are equal (values > INT_MIN)
            k = is_cand ? (v == 0 ? (int)0x80000000 : v) : 0;
            if (is_cand) bad[fbase + rowoff(r, W) + c] = touches ? 1 : 0;
            else if (markers) markers[fbase + rowoff(r, W) + c] = 0;
        }
        key[t] = k;
        // a wave covers one 64-pixel tile row per trip: its ballot is the row's candidate bits, 64 consecutive bits of the
        // flat array (they straddle two words unless the row starts on a word boundary; the array is zeroed by the caller)
        const unsigned long long rowbits = __ballot(k != 0);
        if (lane_id() == 0 && r < H && rowbits) {
            const int64_t g = fbase + rowoff(r, W) + c0;
            const int sh = (int)(g & 63);
            atomicOr(cbits + (g >> 6), rowbits << sh);
            if (sh) atomicOr(cbits + (g >> 6) + 1, rowbits >> (64 - sh));
        }
    }
    if (__any(any_differs) && lane_id() == 0 && nonconst[ti.z] == 0) nonconst[ti.z] = 1;
    __syncthreads();
    // tile pass of the candidates' plateaus: every pixel its own node, then the recorded links (candidates are sparse and their
    // plateaus small: the run-based tile pass of the dense labellings -- ballots and link tests for all 2 048 pixels -- cost 46 us
    // of this kernel's 256, profiles/r04/ab_logs/r4n_*)
    for (int i = threadIdx.x; i < CCL_TILE; i += 256) par[i] = i;
    __syncthreads();
#if !(defined(PCSEG_EXP_LOCMAX) && (PCSEG_EXP_LOCMAX & 2))
    if (links) {
#pragma unroll
        for (int kk = 0; kk < LM_TH * LM_TW / 256; ++kk) {
            const unsigned m = (links >> (4 * kk)) & 15u;
            if (!m) continue;
            const int i = threadIdx.x + 256 * kk;
            if (m & 1u) unite_lds_pair(par, i, i - 1);
            if (m & 2u) unite_lds_pair(par, i, i - CCL_TW - 1);
            if (m & 4u) unite_lds_pair(par, i, i - CCL_TW);
            if (m & 8u) unite_lds_pair(par, i, i - CCL_TW + 1);
        }
    }
    __syncthreads();
#endif
    // tile-local roots -> frame-wide parent entries, candidates only
    for (int i = threadIdx.x; i < CCL_TILE; i += 256) {
        const int r = r0 + i / CCL_TW, c = c0 + i % CCL_TW;
        if (r >= H || c >= W || key[i] == 0) continue;
        const int root = find_lds(par, i);
        parent[fbase + rowoff(r, W) + c] = (r0 + root / CCL_TW) * W + c0 + root % CCL_TW;
    }
}

// The passes after the border links walk the candidate bit words: one thread per 64-pixel word, most words are empty.
// flatten the candidates' parents and move every pixel's own flag to its root
__global__ void __launch_bounds__(256) locmax_propagate_kernel(int *__restrict__ parent, uint8_t *bad,
                                                                const unsigned long long *__restrict__ cbits, int64_t n, int64_t total)
{
    const int64_t w = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (w * 64 >= total) return;
    for (unsigned long long m = cbits[w]; m; m &= m - 1) {
        const int64_t g = w * 64 + (__ffsll((long long)m) - 1);
        const int64_t fbase = (g / n) * n;
        const int i = (int)(g - fbase);
        int *par = parent + fbase;
        const int p = par[i];
        if (!walk_ok(i, p)) continue;   // fenced (common.h, walk_ok)
        const int x = walk_root(par, p);
        if (x != p) par[i] = x;
        if (x != i && bad[g]) bad[fbase + x] = 1;
    }
}

__global__ void __launch_bounds__(256) locmax_out_kernel(const int *__restrict__ parent, const uint8_t *__restrict__ bad,
                                                          const int *__restrict__ nonconst, const unsigned long long *__restrict__ cbits,
                                                          uint8_t *__restrict__ is_max, int64_t n)
{
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int64_t fbase = (int64_t)blockIdx.y * n;
    uint8_t v = 0;
    if (cand_bit(cbits, fbase + i)) v = bad[fbase + parent[fbase + i]] == 0 && nonconst[blockIdx.y] != 0;
    is_max[fbase + i] = v;
}

// the 64 candidate bits of pixels i0 .. i0 + 63 of frame b (bits of pixels >= n cleared; the bit array ends in a spare
// zero word, so word + 1 always exists)
__device__ __forceinline__ unsigned long long cand_word(const unsigned long long *cbits, int b, int64_t i0, int64_t n)
{
    if (i0 >= n) return 0ull;
    const int64_t g0 = (int64_t)b * n + i0;
    const int sh = (int)(g0 & 63);
    unsigned long long w = cbits[g0 >> 6] >> sh;
    if (sh) w |= cbits[(g0 >> 6) + 1] << (64 - sh);
    if (n - i0 < 64) w &= (1ull << (n - i0)) - 1ull;
    return w;
}

// ccl_flatten_count_kernel / ccl_relabel_kernel for candidate components (parents already flat, see the propagate pass):
// same SCAN_PIX blocks, same rank codes, but one THREAD per 64-pixel word of candidate bits (most are empty) instead of
// one per four pixels: sixteen consecutive lanes are one SCAN_PIX block, their root counts are ranked by a 16-wide scan.
static_assert(SCAN_PIX == 1024, "sixteen 64-bit words per count block");
template <typename Pred>
__global__ void __launch_bounds__(256) locmax_count_kernel(const int *__restrict__ parent, int *__restrict__ labels,
                                                            int *__restrict__ blockcount, const unsigned long long *__restrict__ cbits,
                                                            Pred pred, int64_t n, int nblk)
{
    const int b = blockIdx.y;
    const int *par = parent + (int64_t)b * n;
    const int64_t word = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int64_t i0 = word * 64;
    unsigned long long roots = 0;
    for (unsigned long long m = cand_word(cbits, b, i0, n); m; m &= m - 1) {
        const int j = __ffsll((long long)m) - 1;
        const int64_t i = i0 + j;
        if (par[i] == (int)i && pred((int64_t)b * n + i)) roots |= 1ull << j;
    }
    const int cnt = __popcll(roots);
    int inc = cnt;
    for (int off = 1; off < 16; off <<= 1) {
        const int t = __shfl_up(inc, off, 16);
        if ((threadIdx.x & 15) >= off) inc += t;
    }
    int rank = inc - cnt;
    for (unsigned long long m = roots; m; m &= m - 1) labels[(int64_t)b * n + i0 + (__ffsll((long long)m) - 1)] = -(++rank);
    if ((threadIdx.x & 15) == 15 && word / 16 < nblk) blockcount[b * nblk + (int)(word / 16)] = inc;
}

template <typename Pred>
__global__ void __launch_bounds__(256) locmax_relabel_kernel(const int *__restrict__ parent, int *labels, const int *__restrict__ blockoff,
                                                              const unsigned long long *__restrict__ cbits, Pred pred, int64_t n, int nblk)
{
    const int b = blockIdx.y;
    const int *par = parent + (int64_t)b * n;
    int *lab = labels + (int64_t)b * n;
    const int64_t i0 = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 64;
    for (unsigned long long m = cand_word(cbits, b, i0, n); m; m &= m - 1) {
        const int64_t i = i0 + (__ffsll((long long)m) - 1);
        const int p = par[i];  // the component's root
        int v = 0;
        if (pred((int64_t)b * n + p)) {
            v = __hip_atomic_load(lab + p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);  // rank code or final label
            if (v < 0) v = blockoff[b * nblk + p / SCAN_PIX] - v;
        }
        lab[i] = v;
    }
}

// ---- overlap removal (C6) --------------------------------------------------
__global__ void __launch_bounds__(256) overlap_count_kernel(const int *__restrict__ parent, const uint8_t *__restrict__ other,
                                                             int *__restrict__ area, int *__restrict__ ov, int64_t n)
{
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    int64_t fbase = (int64_t)blockIdx.y * n;
    int p = parent[fbase + i];
    if (p < 0) return;
    atomicAdd(&area[fbase + p], 1);
    if (other[fbase + i] == 1) atomicAdd(&ov[fbase + p], 1);
}

__global__ void __launch_bounds__(256) overlap_out_kernel(const uint8_t *__restrict__ dapi, const int *__restrict__ parent,
                                                           const int *__restrict__ area, const int *__restrict__ ov,
                                                           double threshold, uint8_t *__restrict__ out, int64_t n)
{
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    int64_t fbase = (int64_t)blockIdx.y * n;
    int p = parent[fbase + i];
    uint8_t v = dapi[fbase + i];
    if (p >= 0) {
        double frac = (double)ov[fbase + p] / (double)area[fbase + p];
        if (frac > threshold) v = 2;
    }
    out[fbase + i] = v;
}

}  // namespace pcseg

using namespace pcseg;

extern "C" {

size_t pcseg_ccl_workspace_bytes(int B, int H, int W)
{
    if (!check_shape(B, H, W)) return 0;
    return ccl_ws_bytes(B, H, W);
}

int pcseg_ccl8_equal_u8(const uint8_t *in, int32_t *labels, int32_t *counts, int B, int H, int W, void *workspace,
                        size_t workspace_bytes, pcseg_stream_t stream)
{
    PCSEG_REQUIRE(in && labels && counts && workspace && check_shape(B, H, W), "bad arguments");
    return ccl_full<KeyEqU8, true>(KeyEqU8{in, W, (int64_t)H * W}, labels, counts, B, H, W, workspace, workspace_bytes, (hipStream_t)stream);
}

int pcseg_ccl8_bool(const uint8_t *in, int32_t *labels, int32_t *counts, int B, int H, int W, void *workspace,
                    size_t workspace_bytes, pcseg_stream_t stream)
{
    PCSEG_REQUIRE(in && labels && counts && workspace && check_shape(B, H, W), "bad arguments");
    return ccl_full<KeyNzU8, true>(KeyNzU8{in, W, (int64_t)H * W}, labels, counts, B, H, W, workspace, workspace_bytes, (hipStream_t)stream);
}

int pcseg_ccl4_bool(const uint8_t *in, int32_t *labels, int32_t *counts, int B, int H, int W, void *workspace,
                    size_t workspace_bytes, pcseg_stream_t stream)
{
    PCSEG_REQUIRE(in && labels && counts && workspace && check_shape(B, H, W), "bad arguments");
    return ccl_full<KeyNzU8, false>(KeyNzU8{in, W, (int64_t)H * W}, labels, counts, B, H, W, workspace, workspace_bytes, (hipStream_t)stream);
}

size_t pcseg_dilate_ccl_workspace_bytes(int B, int H, int W)
{
    if (!check_shape(B, H, W)) return 0;
    int nch = (H + 31) / 32;
    return 2 * align_up(sizeof(unsigned) * (size_t)B * nch * W);
}

int pcseg_dilate_ccl_roots_u8(const uint8_t *in, uint64_t value_bits, int radius, int32_t *roots, int B, int H, int W,
                              void *workspace, size_t workspace_bytes, pcseg_stream_t stream)
{
    PCSEG_REQUIRE(in && roots && workspace && radius >= 0 && radius <= 15 && check_shape(B, H, W), "bad arguments (radius <= 15)");
    hipStream_t s = (hipStream_t)stream;
    const int nch = (H + 31) / 32;
    Carver cv(workspace, workspace_bytes);
    unsigned *bits = cv.take<unsigned>((size_t)B * nch * W);
    unsigned *dil = cv.take<unsigned>((size_t)B * nch * W);
    if (!cv.ok()) {
        set_error("dilate_ccl_roots: workspace too small (%zu < %zu)", workspace_bytes, cv.off);
        return PCSEG_ERR_WORKSPACE;
    }
    dim3 g((W + 255) / 256, nch, B);
    if ((W & 3) == 0 && ((uintptr_t)in & 3) == 0 && ((uintptr_t)bits & 15) == 0) {
        PCSEG_LAUNCH(set_bits4_kernel, dim3((W / 4 + 255) / 256, g.y, g.z), dim3(256), 0, s, in, (unsigned long long)value_bits, bits, H,
                     W, nch);
    } else {
        PCSEG_LAUNCH(set_bits_kernel, g, dim3(256), 0, s, in, (unsigned long long)value_bits, bits, H, W, nch);
    }
    PCSEG_CHECK_LAUNCH();
    if (radius == 2) PCSEG_LAUNCH(dilate_bits_disk2_kernel, g, dim3(256), 0, s, (const unsigned *)bits, dil, H, W, nch);
    else PCSEG_LAUNCH(dilate_bits_kernel, g, dim3(256), 0, s, (const unsigned *)bits, dil, radius, H, W, nch);
    PCSEG_CHECK_LAUNCH();
    return ccl_roots<KeyBits, true>(KeyBits{dil, W, nch}, roots, B, H, W, s);
}

size_t pcseg_dilate_ccl_runs_workspace_bytes(int B, int H, int W)
{
    if (!check_shape(B, H, W)) return 0;
    int nch = (H + 31) / 32;
    return align_up(sizeof(unsigned) * (size_t)B * nch * W);
}

int pcseg_dilate_ccl_runs_u8(const uint8_t *in, uint64_t value_bits, int radius, uint32_t *dilated_bits, int32_t *run_parent,
                             int B, int H, int W, void *workspace, size_t workspace_bytes, pcseg_stream_t stream)
{
    PCSEG_REQUIRE(in && dilated_bits && run_parent && workspace && radius >= 0 && radius <= 15 && check_shape(B, H, W),
                  "bad arguments (radius <= 15)");
    hipStream_t s = (hipStream_t)stream;
    const int nch = (H + 31) / 32;
    Carver cv(workspace, workspace_bytes);
    unsigned *bits = cv.take<unsigned>((size_t)B * nch * W);
    if (!cv.ok()) {
        set_error("dilate_ccl_runs: workspace too small (%zu < %zu)", workspace_bytes, cv.off);
        return PCSEG_ERR_WORKSPACE;
    }
    dim3 g((W + 255) / 256, nch, B);
    if ((W & 3) == 0 && ((uintptr_t)in & 3) == 0 && ((uintptr_t)bits & 15) == 0) {
        PCSEG_LAUNCH(set_bits4_kernel, dim3((W / 4 + 255) / 256, g.y, g.z), dim3(256), 0, s, in, (unsigned long long)value_bits, bits, H,
                     W, nch);
    } else {
        PCSEG_LAUNCH(set_bits_kernel, g, dim3(256), 0, s, in, (unsigned long long)value_bits, bits, H, W, nch);
    }
    PCSEG_CHECK_LAUNCH();
    if (radius == 2) PCSEG_LAUNCH(dilate_bits_disk2_kernel, g, dim3(256), 0, s, (const unsigned *)bits, (unsigned *)dilated_bits, H, W, nch);
    else PCSEG_LAUNCH(dilate_bits_kernel, g, dim3(256), 0, s, (const unsigned *)bits, (unsigned *)dilated_bits, radius, H, W, nch);
    PCSEG_CHECK_LAUNCH();
    PCSEG_LAUNCH(bitrun_tile_kernel, dim3((W + BR_TW - 1) / BR_TW, (nch + BR_CH - 1) / BR_CH, B), dim3(256), 0, s,
                 (const unsigned *)dilated_bits, (int *)run_parent, H, W, nch);
    PCSEG_CHECK_LAUNCH();
    PCSEG_LAUNCH(bitrun_border_kernel, g, dim3(256), 0, s, (const unsigned *)dilated_bits, (int *)run_parent, H, W, nch);
    PCSEG_CHECK_LAUNCH();
    return PCSEG_OK;
}

int pcseg_dilate_ccl_runs_multi_u8(const uint8_t *in, const uint64_t *value_bits, int n_masks, int radius, uint32_t *dilated_bits,
                                   int32_t *run_parent, int B, int H, int W, void *workspace, size_t workspace_bytes,
                                   pcseg_stream_t stream)
{
    PCSEG_REQUIRE(in && value_bits && dilated_bits && run_parent && workspace && n_masks >= 1 && n_masks <= 4 && radius >= 0 &&
                      radius <= 15 && check_shape(B, H, W) && check_shape(B * n_masks, H, W),
                  "bad arguments (1..4 masks, radius <= 15)");
    PCSEG_REQUIRE((W & 3) == 0 && ((uintptr_t)in & 3) == 0, "W must be a multiple of 4 (use pcseg_dilate_ccl_runs_u8 per mask otherwise)");
    hipStream_t s = (hipStream_t)stream;
    const int nch = (H + 31) / 32;
    const int BM = B * n_masks;
    Carver cv(workspace, workspace_bytes);
    unsigned *bits = cv.take<unsigned>((size_t)BM * nch * W);
    if (!cv.ok() || ((uintptr_t)bits & 15)) {
        set_error("dilate_ccl_runs_multi: workspace too small or misaligned (%zu < %zu)", workspace_bytes, cv.off);
        return PCSEG_ERR_WORKSPACE;
    }
    MaskSet masks;
    masks.n = n_masks;
    for (int m = 0; m < 4; ++m) masks.bits[m] = m < n_masks ? (unsigned long long)value_bits[m] : 0ull;
    PCSEG_LAUNCH(set_bits4_multi_kernel, dim3((W / 4 + 255) / 256, nch, B), dim3(256), 0, s, in, masks, bits, H, W, nch, B);
    PCSEG_CHECK_LAUNCH();
    // everything behind the bit planes sees n_masks * B independent frames
    dim3 g((W + 255) / 256, nch, BM);
    if (radius == 2) PCSEG_LAUNCH(dilate_bits_disk2_kernel, g, dim3(256), 0, s, (const unsigned *)bits, (unsigned *)dilated_bits, H, W, nch);
    else PCSEG_LAUNCH(dilate_bits_kernel, g, dim3(256), 0, s, (const unsigned *)bits, (unsigned *)dilated_bits, radius, H, W, nch);
    PCSEG_CHECK_LAUNCH();
    PCSEG_LAUNCH(bitrun_tile_kernel, dim3((W + BR_TW - 1) / BR_TW, (nch + BR_CH - 1) / BR_CH, BM), dim3(256), 0, s,
                 (const unsigned *)dilated_bits, (int *)run_parent, H, W, nch);
    PCSEG_CHECK_LAUNCH();
    PCSEG_LAUNCH(bitrun_border_kernel, g, dim3(256), 0, s, (const unsigned *)dilated_bits, (int *)run_parent, H, W, nch);
    PCSEG_CHECK_LAUNCH();
    return PCSEG_OK;
}

int pcseg_compact_labels(const int32_t *roots, int32_t *labels, int32_t *counts, int B, int H, int W, void *workspace,
                         size_t workspace_bytes, pcseg_stream_t stream)
{
    PCSEG_REQUIRE(roots && labels && counts && workspace && roots != labels && check_shape(B, H, W), "bad arguments");
    hipStream_t s = (hipStream_t)stream;
    Carver cv(workspace, workspace_bytes);
    CclWs ws = ccl_carve(cv, B, H, W);
    if (!cv.ok()) {
        set_error("compact_labels: workspace too small (%zu < %zu)", workspace_bytes, cv.off);
        return PCSEG_ERR_WORKSPACE;
    }
    int64_t total = (int64_t)B * H * W;
    // the caller's roots are walked like the library's own parents: fenced (common.h, walk_ok) -- an entry that is not "index
    // of an earlier-or-equal pixel of the same frame, + 1" makes counts[b] = -1 instead of a wild load
    PCSEG_CHECK_HIP(hipMemsetAsync(ws.corrupt, 0, sizeof(int) * (size_t)B, s));
    PCSEG_LAUNCH(roots_to_parent_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, roots, ws.parent, total);
    PCSEG_CHECK_LAUNCH();
    return ccl_compact(ws.parent, ws.blockcount, ws.nblk, labels, counts, PredAll(), true, B, H, W, s, ws.corrupt);
}

size_t pcseg_fill_holes_workspace_bytes(int B, int H, int W)
{
    if (!check_shape(B, H, W)) return 0;
    return ccl_ws_bytes(B, H, W) + align_up((size_t)B * H * W);
}

int pcseg_fill_holes(const uint8_t *mask, uint8_t *out, int B, int H, int W, void *workspace, size_t workspace_bytes,
                     pcseg_stream_t stream)
{
    PCSEG_REQUIRE(mask && out && workspace && check_shape(B, H, W), "bad arguments");
    hipStream_t s = (hipStream_t)stream;
    int64_t n = (int64_t)H * W;
    Carver cv(workspace, workspace_bytes);
    CclWs ws = ccl_carve(cv, B, H, W);
    uint8_t *flag = cv.take<uint8_t>((size_t)B * n);
    if (!cv.ok()) {
        set_error("fill_holes: workspace too small (%zu < %zu)", workspace_bytes, cv.off);
        return PCSEG_ERR_WORKSPACE;
    }
    int rc = ccl_roots<KeyZeroU8, false>(KeyZeroU8{mask, W, (int64_t)H * W}, ws.parent, B, H, W, s);
    if (rc) return rc;
    dim3 g1((unsigned)((n + 255) / 256), B);
    PCSEG_LAUNCH(ccl_flatten_kernel, g1, dim3(256), 0, s, ws.parent, n);
    PCSEG_CHECK_LAUNCH();
    PCSEG_CHECK_HIP(hipMemsetAsync(flag, 0, (size_t)B * n, s));
    PCSEG_LAUNCH(border_flag_kernel, dim3((2 * W + 2 * H + 255) / 256, B), dim3(256), 0, s, ws.parent, flag, H, W);
    PCSEG_CHECK_LAUNCH();
    PCSEG_LAUNCH(fill_holes_out_kernel, g1, dim3(256), 0, s, mask, ws.parent, flag, out, n);
    PCSEG_CHECK_LAUNCH();
    return PCSEG_OK;
}

size_t pcseg_local_maxima_workspace_bytes(int B, int H, int W)
{
    if (!check_shape(B, H, W)) return 0;
    size_t n = (size_t)H * W;
    return ccl_ws_bytes(B, H, W) + align_up(8 * (((size_t)B * n + 63) / 64 + 1)) + align_up((size_t)B * n) + align_up(sizeof(int) * B);
}

int pcseg_local_maxima_i32(const int32_t *img, uint8_t *is_max, int32_t *markers, int32_t *counts, int B, int H, int W,
                           void *workspace, size_t workspace_bytes, pcseg_stream_t stream)
{
    PCSEG_REQUIRE(img && workspace && check_shape(B, H, W), "bad arguments");
    PCSEG_REQUIRE(!markers || counts, "markers need counts");
    hipStream_t s = (hipStream_t)stream;
    int64_t n = (int64_t)H * W;
    const int64_t total = (int64_t)B * n;
    const size_t nwords = (size_t)((total + 63) / 64);
    Carver cv(workspace, workspace_bytes);
    CclWs ws = ccl_carve(cv, B, H, W);
    unsigned long long *cbits = cv.take<unsigned long long>(nwords + 1);
    uint8_t *bad = cv.take<uint8_t>((size_t)B * n);
    int *nonconst = cv.take<int>(B);
    if (!cv.ok()) {
        set_error("local_maxima: workspace too small (%zu < %zu)", workspace_bytes, cv.off);
        return PCSEG_ERR_WORKSPACE;
    }
    PCSEG_CHECK_HIP(hipMemsetAsync(nonconst, 0, sizeof(int) * B, s));
    PCSEG_CHECK_HIP(hipMemsetAsync(cbits, 0, sizeof(unsigned long long) * (nwords + 1), s));  // one bit per pixel
    dim3 g2((W + LM_TW - 1) / LM_TW, (H + LM_TH - 1) / LM_TH, B);
    PCSEG_LAUNCH(locmax_candidates_kernel, g2, dim3(256), 0, s, img, cbits, bad, nonconst, ws.parent, (int *)markers, H, W);
    PCSEG_CHECK_LAUNCH();
    // cross-tile links of the plateaus (tile pass: done above); keys come from the candidate bits and the image itself
    int rc = ccl_roots<KeyCandBits, true>(KeyCandBits{img, cbits, W, n}, ws.parent, B, H, W, s, true);
    if (rc) return rc;
    const dim3 gw((unsigned)((nwords + 255) / 256));
    PCSEG_LAUNCH(locmax_propagate_kernel, gw, dim3(256), 0, s, ws.parent, bad, (const unsigned long long *)cbits, n, total);
    PCSEG_CHECK_LAUNCH();
    if (is_max) {
        dim3 g1((unsigned)((n + 255) / 256), B);
        PCSEG_LAUNCH(locmax_out_kernel, g1, dim3(256), 0, s, (const int *)ws.parent, (const uint8_t *)bad, (const int *)nonconst,
                     (const unsigned long long *)cbits, is_max, n);
        PCSEG_CHECK_LAUNCH();
    }
    if (markers) {
        // raster-order numbering of the accepted plateaus: count / scan / relabel over the candidate bits
        const PredNotFlagged pred{bad, nonconst, n};
        dim3 grid((unsigned)((ws.nblk + 15) / 16), B);  // a thread per 64-pixel word, sixteen words per count block
        PCSEG_LAUNCH((locmax_count_kernel<PredNotFlagged>), grid, dim3(256), 0, s, (const int *)ws.parent, markers, ws.blockcount,
                     (const unsigned long long *)cbits, pred, n, ws.nblk);
        PCSEG_CHECK_LAUNCH();
        PCSEG_LAUNCH(ccl_scan_blocks_kernel, dim3(B), dim3(256), 0, s, ws.blockcount, counts, ws.nblk, (const int *)nullptr);
        PCSEG_CHECK_LAUNCH();
        PCSEG_LAUNCH((locmax_relabel_kernel<PredNotFlagged>), grid, dim3(256), 0, s, (const int *)ws.parent, markers,
                     (const int *)ws.blockcount, (const unsigned long long *)cbits, pred, n, ws.nblk);
        PCSEG_CHECK_LAUNCH();
    }
    return PCSEG_OK;
}

size_t pcseg_overlap_workspace_bytes(int B, int H, int W)
{
    if (!check_shape(B, H, W)) return 0;
    size_t n = (size_t)H * W;
    return ccl_ws_bytes(B, H, W) + 2 * align_up(sizeof(int) * B * n);
}

int pcseg_remove_overlapping(const uint8_t *dapi, const uint8_t *other, double threshold, uint8_t *out, int B, int H, int W,
                             void *workspace, size_t workspace_bytes, pcseg_stream_t stream)
{
    PCSEG_REQUIRE(dapi && other && out && workspace && check_shape(B, H, W), "bad arguments");
    hipStream_t s = (hipStream_t)stream;
    int64_t n = (int64_t)H * W;
    Carver cv(workspace, workspace_bytes);
    CclWs ws = ccl_carve(cv, B, H, W);
    int *area = cv.take<int>((size_t)B * n);
    int *ov = cv.take<int>((size_t)B * n);
    if (!cv.ok()) {
        set_error("remove_overlapping: workspace too small (%zu < %zu)", workspace_bytes, cv.off);
        return PCSEG_ERR_WORKSPACE;
    }
    int rc = ccl_roots<KeyIsOneU8, true>(KeyIsOneU8{dapi, W, (int64_t)H * W}, ws.parent, B, H, W, s);
    if (rc) return rc;
    dim3 g1((unsigned)((n + 255) / 256), B);
    PCSEG_LAUNCH(ccl_flatten_kernel, g1, dim3(256), 0, s, ws.parent, n);
    PCSEG_CHECK_LAUNCH();
    PCSEG_CHECK_HIP(hipMemsetAsync(area, 0, sizeof(int) * (size_t)B * n, s));
    PCSEG_CHECK_HIP(hipMemsetAsync(ov, 0, sizeof(int) * (size_t)B * n, s));
    PCSEG_LAUNCH(overlap_count_kernel, g1, dim3(256), 0, s, ws.parent, other, area, ov, n);
    PCSEG_CHECK_LAUNCH();
    PCSEG_LAUNCH(overlap_out_kernel, g1, dim3(256), 0, s, dapi, ws.parent, area, ov, threshold, out, n);
    PCSEG_CHECK_LAUNCH();
    return PCSEG_OK;
}

}  // extern "C"
