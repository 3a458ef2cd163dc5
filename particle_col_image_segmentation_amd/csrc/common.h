// Shared helpers of libpcseg (gfx950 only, wave64).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include "../../include/pcseg.h"

namespace pcseg {

constexpr int WAVE = 64;

void set_error(const char *fmt, ...);

#define PCSEG_CHECK_HIP(expr)                                                              \
    do {                                                                                   \
        hipError_t _e = (expr);                                                            \
        if (_e != hipSuccess) {                                                            \
            pcseg::set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, __LINE__); \
            return PCSEG_ERR_HIP;                                                          \
        }                                                                                  \
    } while (0)

#define PCSEG_CHECK_LAUNCH()                                                               \
    do {                                                                                   \
        hipError_t _e = hipGetLastError();                                                 \
        if (_e != hipSuccess) {                                                            \
            pcseg::set_error("kernel launch failed: %s (%s:%d)", hipGetErrorString(_e), __FILE__, __LINE__); \
            return PCSEG_ERR_HIP;                                                          \
        }                                                                                  \
    } while (0)

#define PCSEG_REQUIRE(cond, msg)                                                           \
    do {                                                                                   \
        if (!(cond)) {                                                                     \
            pcseg::set_error("%s: %s", __func__, msg);                                     \
            return PCSEG_ERR_ARG;                                                          \
        }                                                                                  \
    } while (0)

// optional per-kernel timing (pcseg_timing_enable): hipEvents recorded around every launch ON THE LAUNCH STREAM
struct LaunchTimer {
    hipStream_t stream;
    void *stop;  // hipEvent_t recorded by the destructor (nullptr: timing off)
    LaunchTimer(const char *kernel, const char *where, hipStream_t s);  // where = __PRETTY_FUNCTION__ (template arguments)
    ~LaunchTimer();
};

#define PCSEG_LAUNCH(kernel, grid, block, lds, stream, ...)                 \
    do {                                                                    \
        pcseg::LaunchTimer _timer(#kernel, __PRETTY_FUNCTION__, stream);                         \
        hipLaunchKernelGGL(kernel, grid, block, lds, stream, __VA_ARGS__);  \
    } while (0)

inline int check_shape(int B, int H, int W)
{
    // linear indices are int32 per frame; rows/cols are stored as uint16 in the EDT scratch
    if (B < 1 || H < 1 || W < 1) return 0;
    if ((int64_t)H * W >= ((int64_t)1 << 30)) return 0;
    if (H > 32768 || W > 32768) return 0;
    return 1;
}

inline size_t align_up(size_t x, size_t a = 256) { return (x + a - 1) / a * a; }

// carve typed regions out of the caller's workspace
struct Carver {
    char *base;
    size_t off = 0;
    size_t cap;
    Carver(void *p, size_t bytes) : base((char *)p), cap(bytes) {}
    template <typename T>
    T *take(size_t n)
    {
        size_t bytes = align_up(n * sizeof(T));
        T *p = (T *)(base + off);
        off += bytes;
        return p;
    }
    bool ok() const { return off <= cap && base != nullptr; }
};

__device__ __forceinline__ int lane_id() { return threadIdx.x & (WAVE - 1); }

// row * W as a FULL-RATE 24-bit multiply, widened afterwards: rows, word rows and widths are below 2^15 (check_shape) and a
// frame has fewer than 2^30 pixels, so the product fits 32 bits.  Written as (int64_t)r * W the compiler emits
// v_mad_u64_u32 -- a quarter-rate instruction -- for every pixel address of the tile kernels (87 of them in the union-find
// tile pass, whose VALU pipe is the bottleneck).
__device__ __forceinline__ int64_t rowoff(int r, int W) { return (int64_t)__mul24(r, W); }

// Inclusive prefix maximum / minimum over the 64 lanes of a wave with DPP moves only (no LDS round trip per step, which
// is what __shfl_up costs: ds_bpermute + a wait): four shifts inside the 16-lane rows, then lane 15 of rows 0 and 2 is
// broadcast into rows 1 and 3, then lane 31 into rows 2 and 3.  Lanes without a source keep `ident` (bound_ctrl off).
template <bool MAX>
__device__ __forceinline__ int wave_prefix_extreme(int x)
{
    constexpr int ident = MAX ? (int)0x80000000 : 0x7FFFFFFF;
#define PCSEG_DPP_STEP(ctrl, row_mask)                                                              \
    {                                                                                               \
        const int t = __builtin_amdgcn_update_dpp(ident, x, ctrl, row_mask, 0xF, false);            \
        x = MAX ? max(x, t) : min(x, t);                                                            \
    }
    PCSEG_DPP_STEP(0x111, 0xF)  // row_shr:1
    PCSEG_DPP_STEP(0x112, 0xF)  // row_shr:2
    PCSEG_DPP_STEP(0x114, 0xF)  // row_shr:4
    PCSEG_DPP_STEP(0x118, 0xF)  // row_shr:8
    PCSEG_DPP_STEP(0x142, 0xA)  // row_bcast:15 into rows 1 and 3
    PCSEG_DPP_STEP(0x143, 0xC)  // row_bcast:31 into rows 2 and 3
#undef PCSEG_DPP_STEP
    return x;
}
__device__ __forceinline__ int wave_prefix_max(int x) { return wave_prefix_extreme<true>(x); }
__device__ __forceinline__ int wave_prefix_min(int x) { return wave_prefix_extreme<false>(x); }
__device__ __forceinline__ int wave_last_lane(int x) { return __builtin_amdgcn_readlane(x, WAVE - 1); }

// XCD-aware tile order for kernels whose blocks read a halo of their neighbours' pixels.  Workgroups are dealt
// round-robin over the 8 XCDs (blocks b and b + 8 share one; MI355X_MICROARCH.md, "Workgroup dispatch"), each XCD with
// an L2 of its own: with the plain blockIdx -> tile map the left / right neighbour of a tile runs on ANOTHER XCD, so
// every halo column costs its full cache line from HBM a second time (for a 64-column tile of 4-byte pixels: 6 lines
// fetched per 4 used).  This map hands XCD k the k-th eighth of the tile sequence (x fastest, then y, then frame), in
// order: neighbouring tiles of a frame are issued together on one XCD and share their halo lines in its L2.  Speed
// only -- any block -> tile bijection is correct.
struct TileIndex {
    int x, y, z;
};
__device__ __forceinline__ TileIndex xcd_tile_index()
{
    const unsigned gx = gridDim.x, gy = gridDim.y;
    const unsigned total = gx * gy * gridDim.z;
    const unsigned b = blockIdx.x + gx * (blockIdx.y + gy * blockIdx.z);
    const unsigned per = total / 8u;
    const unsigned t = b < per * 8u ? (b % 8u) * per + b / 8u : b;  // (a remainder of < 8 blocks keeps its place)
    TileIndex ti;
    ti.x = (int)(t % gx);
    ti.y = (int)((t / gx) % gy);
    ti.z = (int)(t / (gx * gy));
    return ti;
}

// relaxed agent-scope accesses: served by L2, never by a stale L1 line
__device__ __forceinline__ int ld_agent(const int *p)
{
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ unsigned ld_agent(const unsigned *p)
{
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// ---- LDS union-find
// (relaxed workgroup-scope atomic accesses, not `volatile`: the compiler keeps a volatile access through a pointer
// argument in the generic address space -- flat_load / flat_store instead of ds_read / ds_write, 68 of them in the
// watershed tile pass -- while it does infer LDS for these)
__device__ __forceinline__ int ld_lds(const int *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
__device__ __forceinline__ void st_lds(int *p, int v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
__device__ __forceinline__ int find_lds(int *par, int x)
{
    int p;
    while ((p = ld_lds(par + x)) != x) x = p;
    return x;
}
__device__ __forceinline__ void unite_lds(int *par, int a, int b)
{
    for (;;) {
        a = find_lds(par, a);
        b = find_lds(par, b);
        if (a == b) return;
        if (a < b) { int t = a; a = b; b = t; }
        int old = atomicMin(&par[a], b);
        if (old == a) return;
        a = old;
    }
}
// the same with both walks in lockstep (two independent LDS reads per step).  Pays in the class-map tile pass (451 -> 434
// us); the run-based tile pass got slower with it (155 -> 168 us) and so did the watershed's, which is bound by its
// vector instructions, not by LDS latency (360 -> 424 us): they keep the plain form.
__device__ __forceinline__ void unite_lds_pair(int *par, int a, int b)
{
    for (;;) {
        for (;;) {
            if (a == b) return;
            const int pa = ld_lds(par + a), pb = ld_lds(par + b);
            if (pa == a && pb == b) break;
            a = pa;
            b = pb;
        }
        if (a < b) { int t = a; a = b; b = t; }
        int old = atomicMin(&par[a], b);
        if (old == a) return;
        a = old;
    }
}

// with path halving (long chains: the lake components of the watershed's second level wind through a tile)
__device__ __forceinline__ int find_lds_halving(int *par, int x)
{
    int p;
    while ((p = ld_lds(par + x)) != x) {
        const int g = ld_lds(par + p);
        if (g != p) st_lds(par + x, g);  // re-pointing x at its grandparent keeps it in its set (parents only ever decrease)
        x = g;
    }
    return x;
}
__device__ __forceinline__ void unite_lds_halving(int *par, int a, int b)
{
    for (;;) {
        a = find_lds_halving(par, a);
        b = find_lds_halving(par, b);
        if (a == b) return;
        if (a < b) { int t = a; a = b; b = t; }
        const int old = atomicMin(&par[a], b);
        if (old == a) return;
        a = old;
    }
}

// ---- global union-find (parents only ever decrease; stale reads cost iterations, never correctness)
//
// WALKS ARE FENCED.  In a well-formed union-find image the entry stored at index x is a smaller-or-equal index of the same
// set and never negative (roots are minima; unions only ever lower an entry), so a walk is strictly decreasing: finite and
// inside [0, x].  `walk_ok` is ONE unsigned compare that holds a walk to exactly that, whatever the array contains -- an
// image that is being rewritten by another instance of the same chain (one captured graph replayed on several streams at
// once: profiles/r03/exp_graph_r3a.log) or roots handed in by a caller (pcseg_compact_labels) then
// ends in an error flag (`corrupt`, may be null: the walk just stops) instead of a load from a wild address or a loop
// that never ends.
__device__ __forceinline__ bool walk_ok(int x, int p) { return (unsigned)p <= (unsigned)x; }
__device__ __forceinline__ void walk_corrupt(int *corrupt)
{
    if (corrupt) __hip_atomic_store(corrupt, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// plain (read-only) walk to the root of x; x must be a valid index
__device__ __forceinline__ int walk_root(const int *par, int x, int *corrupt = nullptr)
{
    int q;
    while ((q = par[x]) != x) {
        if (!walk_ok(x, q)) {
            walk_corrupt(corrupt);
            break;
        }
        x = q;
    }
    return x;
}

__device__ __forceinline__ int find_glb(int *par, int x, int *corrupt = nullptr)
{
    // path halving: re-pointing x at its grandparent is always valid (any ancestor is) and races are benign
    int p;
    while ((p = ld_agent(par + x)) != x) {
        if (!walk_ok(x, p)) { walk_corrupt(corrupt); return x; }
        int g = ld_agent(par + p);
        if (!walk_ok(p, g)) { walk_corrupt(corrupt); return p; }
        if (g != p) __hip_atomic_store(par + x, g, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        x = g;
    }
    return x;
}
// the two walks to the roots advance in LOCKSTEP (path halving on both, as in find_glb): a step of the pair is two
// independent loads in flight instead of one, and the border passes are nothing but these dependent walks.
// Returns false (and raises `corrupt`) when an entry breaks the fence: the caller must not link anything then.
__device__ __forceinline__ bool find2_glb(int *par, int &a, int &b, int *corrupt = nullptr)
{
    for (;;) {
        if (a == b) return true;
        const int pa = ld_agent(par + a), pb = ld_agent(par + b);
        if (pa == a && pb == b) return true;
        if (!walk_ok(a, pa) || !walk_ok(b, pb)) break;
        const int ga = ld_agent(par + pa), gb = ld_agent(par + pb);
        if (!walk_ok(pa, ga) || !walk_ok(pb, gb)) break;
        if (pa != a) {
            if (ga != pa) __hip_atomic_store(par + a, ga, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            a = ga;
        }
        if (pb != b) {
            if (gb != pb) __hip_atomic_store(par + b, gb, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            b = gb;
        }
    }
    walk_corrupt(corrupt);
    return false;
}
__device__ __forceinline__ void unite_glb(int *par, int a, int b, int *corrupt = nullptr)
{
    for (;;) {
        if (!find2_glb(par, a, b, corrupt)) return;
        if (a == b) return;
        if (a < b) { int t = a; a = b; b = t; }
        int old = atomicMin(par + a, b);
        if (old == a) return;
        if (!walk_ok(a, old)) { walk_corrupt(corrupt); return; }
        a = old;
    }
}

// ccl.hip, for the fused class-map front end (frontend.hip): where the union-find image and the per-block root counts
// live in a pcseg_ccl_workspace_bytes workspace, and the passes after the tile pass for equal-valued 8-connected
// components of a uint8 image (border links, flatten + count, scan, relabel -> raster-order labels)
struct CclPlan {
    int *parent;
    int *blockcount;
    int nblk;
    int *corrupt;  // [B] raised by a fenced walk (see walk_ok): the frame's count comes out as -1
};
int ccl_plan(void *workspace, size_t workspace_bytes, int B, int H, int W, CclPlan *plan, const char *who);
int ccl_equal_u8_finish(const uint8_t *in, const CclPlan &plan, bool tile_pass_done, int *labels, int *counts, int B, int H, int W,
                        hipStream_t s);

}  // namespace pcseg
