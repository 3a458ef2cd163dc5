// Fused front end of the class-map chain: class map (argmax over the probability planes + 1, what ilastik's "Simple
// Segmentation" export holds, tiff_analysis.py:639-642) -> 5x5 median (A1, :643) -> the LDS tile pass of label()
// (A2, :743) in ONE kernel.  The raw class map is never written: a block computes the classes of its 64x32 tile plus
// the 2-pixel reflected halo straight from the planes, takes the medians from one-hot histogram words in LDS, and
// runs the tile-local union-find on the medians while they are still there.  HBM: the planes once (x 1.2 for the
// halo), the denoised map and the union-find image written once -- instead of planes -> class map -> denoised map ->
// union-find image through three launches.
#include "common.h"
#include "tile_ops.h"

namespace pcseg {

template <int CT>  // CT > 0: compile-time plane count (loads of all planes in flight together); 0: runtime C
__device__ __forceinline__ unsigned classes4(const float *__restrict__ fr, int C, int64_t n, int64_t off)
{
    const int NC = CT > 0 ? CT : C;
    float4 v[CT > 0 ? CT : 1];
    if (CT > 0) {
#pragma unroll
        for (int k = 0; k < CT; ++k) v[k] = *reinterpret_cast<const float4 *>(fr + (int64_t)k * n + off);
    }
    float4 best = CT > 0 ? v[0] : *reinterpret_cast<const float4 *>(fr + off);
    unsigned a0 = 1, a1 = 1, a2 = 1, a3 = 1;
#pragma unroll
    for (int k = 1; k < NC; ++k) {
        const float4 x = CT > 0 ? v[CT > 0 ? k : 0] : *reinterpret_cast<const float4 *>(fr + (int64_t)k * n + off);
        if (x.x > best.x) { best.x = x.x; a0 = k + 1; }  // first maximum wins, like numpy.argmax
        if (x.y > best.y) { best.y = x.y; a1 = k + 1; }
        if (x.z > best.z) { best.z = x.z; a2 = k + 1; }
        if (x.w > best.w) { best.w = x.w; a3 = k + 1; }
    }
    return a0 | (a1 << 8) | (a2 << 16) | (a3 << 24);
}

__device__ __forceinline__ unsigned class1(const float *__restrict__ fr, int C, int64_t n, int64_t off)
{
    float best = fr[off];
    unsigned arg = 1;
    for (int k = 1; k < C; ++k) {
        const float x = fr[(int64_t)k * n + off];
        if (x > best) { best = x; arg = k + 1; }
    }
    return arg;
}

template <int CT>
#ifndef PCSEG_FRONTEND_OCC
#define PCSEG_FRONTEND_OCC 1  // A/B: 8 = 64 registers (20 bytes of scratch) and eight waves per SIMD instead of 67 and seven
#endif
__global__ void __launch_bounds__(256, PCSEG_FRONTEND_OCC) classmap_median_ccl_kernel(const float *__restrict__ stack, int C, uint8_t *__restrict__ z,
                                                                   int *__restrict__ parent, int H, int W)
{
    // the histogram words are dead once the medians are out: the union-find parents take their place (18 KB per block
    // instead of 26: 8 blocks per CU -- the kernel waits on memory most of the time and wants the waves)
    __shared__ __attribute__((aligned(16))) uint32_t hot[MED_LH * MED_LW];  // 1 << (6 * (class - 1)) of the tile + halo
    __shared__ int key[CCL_TILE];
    static_assert(sizeof(uint32_t) * MED_LH * MED_LW >= sizeof(int) * CCL_TILE, "parents fit the histogram words");
    int *par = reinterpret_cast<int *>(hot);
    const TileIndex ti = xcd_tile_index();  // the 2-pixel halo is the neighbours' rim: keep them on one XCD's L2
    const int b = ti.z;
    const int r0 = ti.y * MED_TH, c0 = ti.x * MED_TW;
    const int64_t n = (int64_t)H * W;
    const float *fr = stack + (int64_t)b * C * n;
    // (1) classes of the 36 x 68 pixels around the tile.  Interior columns of a full-width tile: 16-byte loads
    const bool wide = c0 + MED_TW <= W && (W & 3) == 0 && ((uintptr_t)stack & 15) == 0;
    if (wide) {
        // (unrolled, branch-free row reflection for H >= 2: the loads of all trips are issued as one batch)
        constexpr int QUADS = MED_LH * (MED_TW / 4), TRIPS = (QUADS + 255) / 256;
        unsigned cls4[TRIPS];
#pragma unroll
        for (int t = 0; t < TRIPS; ++t) {
            const int i = min((int)threadIdx.x + 256 * t, QUADS - 1);
            const int lr = i / (MED_TW / 4), q = i % (MED_TW / 4);
            int rr = r0 + lr - 2;
            rr = H >= 4 ? (rr < 0 ? -1 - rr : (rr >= H ? 2 * H - 1 - rr : rr)) : reflect_idx(rr, H);
            rr = min(max(rr, 0), H - 1);  // (rows of a ragged last tile far below the frame: any valid row, unused)
            cls4[t] = classes4<CT>(fr, C, n, rowoff(rr, W) + c0 + 4 * q);
        }
#pragma unroll
        for (int t = 0; t < TRIPS; ++t) {
            const int i = (int)threadIdx.x + 256 * t;
            if (i < QUADS) {
                const int lr = i / (MED_TW / 4), q = i % (MED_TW / 4);
                const unsigned a = cls4[t];
                uint32_t *dst = hot + lr * MED_LW + 2 + 4 * q;
                dst[0] = 1u << (6 * (a & 255u) - 6);
                dst[1] = 1u << (6 * ((a >> 8) & 255u) - 6);
                dst[2] = 1u << (6 * ((a >> 16) & 255u) - 6);
                dst[3] = 1u << (6 * (a >> 24) - 6);
            }
        }
        for (int i = threadIdx.x; i < MED_LH * 4; i += 256) {  // the two halo columns either side
            const int lr = i >> 2, k = i & 3, lc = k < 2 ? k : MED_TW + k;
            int rr = r0 + lr - 2, cc = c0 + lc - 2;
            if (rr < 0 || rr >= H) rr = reflect_idx(rr, H);
            if (cc < 0 || cc >= W) cc = reflect_idx(cc, W);
            hot[lr * MED_LW + lc] = 1u << (6 * class1(fr, C, n, rowoff(rr, W) + cc) - 6);
        }
    } else {
        for (int i = threadIdx.x; i < MED_LH * (MED_TW + 4); i += 256) {
            const int lr = i / (MED_TW + 4), lc = i % (MED_TW + 4);
            int rr = r0 + lr - 2, cc = c0 + lc - 2;
            if (rr < 0 || rr >= H) rr = reflect_idx(rr, H);
            if (cc < 0 || cc >= W) cc = reflect_idx(cc, W);
            hot[lr * MED_LW + lc] = 1u << (6 * class1(fr, C, n, rowoff(rr, W) + cc) - 6);
        }
    }
    __syncthreads();
    // (2) medians -> the denoised map (global) and the union-find keys (LDS; 0 = outside the frame, classes are >= 1).
    // A thread takes two 4-pixel strips, one below the other (MED_TW / 4 strips x MED_TH / 2 row pairs = 256 threads).
    uint8_t *dst = z + (int64_t)b * n;
    static_assert((MED_TW / 4) * (MED_TH / 2) == 256, "one pair of strips per thread");
    {
        const int s = threadIdx.x;
        const int lr = 2 * (s / (MED_TW / 4)), lc = (s % (MED_TW / 4)) * 4;
        const int c = c0 + lc;
        uint32_t med[2][4] = {{0, 0, 0, 0}, {0, 0, 0, 0}};
#if defined(PCSEG_EXP_FRONTEND) && (PCSEG_EXP_FRONTEND & 1)  // (ablation builds, profiles/r04/time_ops.py: the kernel's time without a phase)
        for (int j = 0; j < 4; ++j) med[0][j] = med[1][j] = 1 + ((lr + lc) & 1);
#else
        if (r0 + lr < H && c < W) median5_hot6_strip2(hot, lr, lc, med[0], med[1]);
#endif
#pragma unroll
        for (int half = 0; half < 2; ++half) {
            const int r = r0 + lr + half;
#pragma unroll
            for (int j = 0; j < 4; ++j) key[(lr + half) * CCL_TW + lc + j] = (r < H && c + j < W) ? (int)med[half][j] : 0;
            if (r < H && c < W) {
                if (c + 3 < W && (W & 3) == 0) {
                    *reinterpret_cast<uint32_t *>(dst + rowoff(r, W) + c) =
                        med[half][0] | (med[half][1] << 8) | (med[half][2] << 16) | (med[half][3] << 24);
                } else {
                    for (int j = 0; j < 4 && c + j < W; ++j) dst[rowoff(r, W) + c + j] = (uint8_t)med[half][j];
                }
            }
        }
    }
    __syncthreads();
    // (3) tile pass of the equal-value 8-connected labelling
#if defined(PCSEG_EXP_FRONTEND) && (PCSEG_EXP_FRONTEND & 2)
    for (int i = threadIdx.x; i < CCL_TILE; i += 256) par[i] = i;
    __syncthreads();
#else
    ccl_tile_unions<true>(key, par);
#endif
    ccl_tile_store(key, par, parent, (int64_t)b * n, r0, c0, H, W);
}

}  // namespace pcseg

using namespace pcseg;

extern "C" {

size_t pcseg_classmap_label_workspace_bytes(int B, int H, int W)
{
    if (!check_shape(B, H, W)) return 0;
    return pcseg_ccl_workspace_bytes(B, H, W) + align_up((size_t)B * H * W);  // + a raw class map for the unfused path
}

int pcseg_classmap_label_f32(const float *stack, int C, uint8_t *denoised, int32_t *labels, int32_t *counts, int B, int H, int W,
                             void *workspace, size_t workspace_bytes, pcseg_stream_t stream)
{
    PCSEG_REQUIRE(stack && denoised && labels && counts && workspace && C >= 1 && C <= 255 && check_shape(B, H, W), "bad arguments");
    static_assert(MED_TW == CCL_TW && MED_TH == CCL_TH, "one tile shape for the median and the union-find");
    hipStream_t s = (hipStream_t)stream;
    const size_t ccl_bytes = pcseg_ccl_workspace_bytes(B, H, W);
    PCSEG_REQUIRE(workspace_bytes >= ccl_bytes + align_up((size_t)B * H * W), "workspace too small");
    if (C > 5) {
        // more than five planes: class values above 5 do not fit the one-hot histogram words -- the separate kernels
        uint8_t *raw = (uint8_t *)workspace + ccl_bytes;
        int rc = pcseg_argmax_planes_f32(stack, raw, B, C, H, W, stream);
        if (rc) return rc;
        rc = pcseg_median5_u8(raw, denoised, B, H, W, stream);
        if (rc) return rc;
        return pcseg_ccl8_equal_u8(denoised, labels, counts, B, H, W, workspace, ccl_bytes, stream);
    }
    CclPlan plan;
    int rc = ccl_plan(workspace, ccl_bytes, B, H, W, &plan, "classmap_label");
    if (rc) return rc;
    const dim3 grid((W + MED_TW - 1) / MED_TW, (H + MED_TH - 1) / MED_TH, B);
    if (C == 5)
        PCSEG_LAUNCH(classmap_median_ccl_kernel<5>, grid, dim3(256), 0, s, stack, C, denoised, plan.parent, H, W);
    else if (C == 4)
        PCSEG_LAUNCH(classmap_median_ccl_kernel<4>, grid, dim3(256), 0, s, stack, C, denoised, plan.parent, H, W);
    else if (C == 3)
        PCSEG_LAUNCH(classmap_median_ccl_kernel<3>, grid, dim3(256), 0, s, stack, C, denoised, plan.parent, H, W);
    else
        PCSEG_LAUNCH(classmap_median_ccl_kernel<0>, grid, dim3(256), 0, s, stack, C, denoised, plan.parent, H, W);
    PCSEG_CHECK_LAUNCH();
    return ccl_equal_u8_finish(denoised, plan, true, labels, counts, B, H, W, s);
}

}  // extern "C"
