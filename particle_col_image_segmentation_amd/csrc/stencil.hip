// Pointwise and small-stencil kernels: class-map ingest (argmax), 5x5 median
// (A1), threshold (R1), 3x3 morphology (X2), Otsu histogram (X1).
// HBM-bound byte work: coalesced row-major loads, LDS tile with halo for the
// median, no MFMA (nothing here is a contraction).
#include <stdarg.h>

#include <atomic>
#include <deque>
#include <map>
#include <mutex>
#include <string>
#include <vector>

#include "common.h"
#include "tile_ops.h"

namespace pcseg {

static thread_local char g_err[512] = "";

void set_error(const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof g_err, fmt, ap);
    va_end(ap);
}

// ---------------------------------------------------------------- launch timing
struct TimingSlot {
    const char *name, *where;
    hipEvent_t a, b;
};
static std::mutex g_tmutex;
static std::atomic<bool> g_timing{false};
static std::deque<TimingSlot> g_slots;  // a deque: growing it never moves the slots other threads hold handles into
static size_t g_used = 0;

LaunchTimer::LaunchTimer(const char *n, const char *where, hipStream_t s) : stream(s), stop(nullptr)
{
    if (!g_timing.load(std::memory_order_relaxed)) return;
    {   // a stream that is being captured into a hipGraph takes no timing events (they would become graph nodes)
        hipStreamCaptureStatus st = hipStreamCaptureStatusNone;
        if (hipStreamIsCapturing(s, &st) != hipSuccess || st != hipStreamCaptureStatusNone) return;
    }
    hipEvent_t start;
    {
        std::lock_guard<std::mutex> lk(g_tmutex);
        if (g_used == g_slots.size()) {
            TimingSlot t{n, where, nullptr, nullptr};
            if (hipEventCreate(&t.a) != hipSuccess || hipEventCreate(&t.b) != hipSuccess) return;
            g_slots.push_back(t);
        }
        TimingSlot &slot = g_slots[g_used++];
        slot.name = n;
        slot.where = where;
        start = slot.a;
        stop = slot.b;  // the destructor records this handle without looking at the table again
    }
    (void)hipEventRecord(start, s);
}

LaunchTimer::~LaunchTimer()
{
    if (stop) (void)hipEventRecord((hipEvent_t)stop, stream);
}

// ---------------------------------------------------------------- argmax
// 4 pixels per thread when the plane size allows 16-byte loads.
template <bool VEC>
__global__ void __launch_bounds__(256) argmax_kernel(const float *__restrict__ stack, uint8_t *__restrict__ cls,
                                                      int C, int64_t n /* H*W */)
{
    const int b = blockIdx.y;
    const float *fr = stack + (int64_t)b * C * n;
    uint8_t *out = cls + (int64_t)b * n;
    if (VEC) {
        int64_t i4 = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
        if (i4 * 4 >= n) return;
        float4 best = *reinterpret_cast<const float4 *>(fr + i4 * 4);
        uchar4 arg = make_uchar4(1, 1, 1, 1);
        for (int k = 1; k < C; ++k) {
            float4 v = *reinterpret_cast<const float4 *>(fr + k * n + i4 * 4);
            if (v.x > best.x) { best.x = v.x; arg.x = k + 1; }
            if (v.y > best.y) { best.y = v.y; arg.y = k + 1; }
            if (v.z > best.z) { best.z = v.z; arg.z = k + 1; }
            if (v.w > best.w) { best.w = v.w; arg.w = k + 1; }
        }
        *reinterpret_cast<uchar4 *>(out + i4 * 4) = arg;
    } else {
        int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
        if (i >= n) return;
        float best = fr[i];
        int arg = 1;
        for (int k = 1; k < C; ++k) {
            float v = fr[k * n + i];
            if (v > best) { best = v; arg = k + 1; }
        }
        out[i] = (uint8_t)arg;
    }
}

// ---------------------------------------------------------------- median
// Tile 64x32 outputs per 256-thread block, 2-pixel reflected halo, each thread
// produces a 4-pixel horizontal strip for 2 rows.  The median of the 25 window
// bytes is built bit by bit: med = max t with #(v >= t) >= 13; the number of
// bit rounds is block-uniform (from the tile maximum), so a class map with
// values <= 7 costs 3 rounds.
__global__ void __launch_bounds__(256) median5_kernel(const uint8_t *__restrict__ in, uint8_t *__restrict__ out, int H, int W)
{
    __shared__ __attribute__((aligned(16))) uint8_t tile[MED_LH * MED_LW];
    __shared__ int tile_max;
    const int b = blockIdx.z;
    const int r0 = blockIdx.y * MED_TH, c0 = blockIdx.x * MED_TW;
    const uint8_t *src = in + (int64_t)b * H * W;
    uint8_t *dst = out + (int64_t)b * H * W;
    if (threadIdx.x == 0) tile_max = 0;
    __syncthreads();
    int local_max = 0;
    for (int i = threadIdx.x; i < MED_LH * (MED_TW + 4); i += 256) {
        int lr = i / (MED_TW + 4), lc = i % (MED_TW + 4);
        int rr = r0 + lr - 2, cc = c0 + lc - 2;
        if (rr < 0 || rr >= H) rr = reflect_idx(rr, H);  // only tiles on the frame's rim pay for the modulo
        if (cc < 0 || cc >= W) cc = reflect_idx(cc, W);
        uint8_t v = src[rowoff(rr, W) + cc];
        tile[lr * MED_LW + lc] = v;
        local_max = max(local_max, (int)v);
    }
    for (int off = 32; off; off >>= 1) local_max = max(local_max, __shfl_xor(local_max, off));
    if (lane_id() == 0) atomicMax(&tile_max, local_max);
    __syncthreads();
    const int nbits = 32 - __clz(tile_max | 1);
    if (tile_max <= 5) {
        // Small alphabets (class maps): every pixel becomes a one-hot word with one 5-bit counter per value
        // (1 << 5 v); the 25 words of a window then ADD up to the histogram of the window (a count is at most 25 < 32,
        // so the fields never carry into each other), and the median is the first value whose cumulative count
        // reaches 13.  Column sums of 5 rows are shared by the 4 outputs of a strip: 42 adds instead of 300 compares.
        __shared__ __attribute__((aligned(16))) uint32_t hot[MED_LH * MED_LW];
        for (int i = threadIdx.x; i < MED_LH * MED_LW; i += 256) hot[i] = 1u << (5 * tile[i]);  // (pad columns: unused)
        __syncthreads();
        for (int s = threadIdx.x; s < (MED_TW / 4) * MED_TH; s += 256) {
            const int lr = s / (MED_TW / 4), lc = (s % (MED_TW / 4)) * 4;
            const int r = r0 + lr;
            if (r >= H || c0 + lc >= W) continue;
            uint32_t med[4];
            median5_hot_strip(hot, lr, lc, med);
            const int c = c0 + lc;
            if (c + 3 < W && (W & 3) == 0) {
                *reinterpret_cast<uint32_t *>(dst + rowoff(r, W) + c) = med[0] | (med[1] << 8) | (med[2] << 16) | (med[3] << 24);
            } else {
                for (int j = 0; j < 4 && c + j < W; ++j) dst[rowoff(r, W) + c + j] = (uint8_t)med[j];
            }
        }
        return;
    }
    // strip id: 16 strips per row, 32 rows -> 512 strips, 2 per thread
    for (int s = threadIdx.x; s < (MED_TW / 4) * MED_TH; s += 256) {
        int lr = s / (MED_TW / 4), lc = (s % (MED_TW / 4)) * 4;
        int r = r0 + lr;
        if (r >= H || c0 + lc >= W) continue;
        uint32_t v[5][8];
#pragma unroll
        for (int dr = 0; dr < 5; ++dr) {
            const uint32_t *row = reinterpret_cast<const uint32_t *>(tile + (lr + dr) * MED_LW + lc);
            uint32_t w0 = row[0], w1 = row[1];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                v[dr][k] = (w0 >> (8 * k)) & 0xFF;
                v[dr][4 + k] = (w1 >> (8 * k)) & 0xFF;
            }
        }
        uint32_t med[4] = {0, 0, 0, 0};
        for (int bit = nbits - 1; bit >= 0; --bit) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                uint32_t t = med[j] | (1u << bit);
                int cnt = 0;
#pragma unroll
                for (int dr = 0; dr < 5; ++dr)
#pragma unroll
                    for (int dc = 0; dc < 5; ++dc) cnt += (v[dr][j + dc] >= t) ? 1 : 0;
                if (cnt >= 13) med[j] = t;
            }
        }
        int c = c0 + lc;
        if (c + 3 < W && (W & 3) == 0) {
            *reinterpret_cast<uint32_t *>(dst + rowoff(r, W) + c) =
                med[0] | (med[1] << 8) | (med[2] << 16) | (med[3] << 24);
        } else {
            for (int j = 0; j < 4 && c + j < W; ++j) dst[rowoff(r, W) + c + j] = (uint8_t)med[j];
        }
    }
}

// ---------------------------------------------------------------- threshold
__global__ void __launch_bounds__(256) threshold_lt_kernel(const float *__restrict__ img, float thr,
                                                            uint8_t *__restrict__ mask, int64_t total)
{
    int64_t i4 = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) * 4;
    if (i4 + 3 < total && (total & 3) == 0) {
        float4 v = *reinterpret_cast<const float4 *>(img + i4);
        *reinterpret_cast<uchar4 *>(mask + i4) = make_uchar4(v.x < thr, v.y < thr, v.z < thr, v.w < thr);
    } else {
        for (int j = 0; j < 4 && i4 + j < total; ++j) mask[i4 + j] = img[i4 + j] < thr;
    }
}

// ---------------------------------------------------------------- 3x3 morphology
__global__ void __launch_bounds__(256) morph3x3_kernel(const uint8_t *__restrict__ in, uint8_t *__restrict__ out,
                                                        int erode, int H, int W)
{
    int c = blockIdx.x * 64 + (threadIdx.x & 63);
    int r = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (r >= H || c >= W) return;
    const uint8_t *src = in + (int64_t)blockIdx.z * H * W;
    int acc = erode ? 1 : 0;
    for (int dr = -1; dr <= 1; ++dr)
        for (int dc = -1; dc <= 1; ++dc) {
            int rr = r + dr, cc = c + dc;
            int v = (rr < 0 || rr >= H || cc < 0 || cc >= W) ? (erode ? 1 : 0) : (src[rowoff(rr, W) + cc] != 0);
            acc = erode ? (acc & v) : (acc | v);
        }
    out[(int64_t)blockIdx.z * H * W + (int64_t)r * W + c] = (uint8_t)acc;
}

// ---------------------------------------------------------------- Otsu histogram
__device__ __forceinline__ unsigned f32_ordered(float f)
{
    unsigned u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float ordered_f32(unsigned k)
{
    unsigned u = (k & 0x80000000u) ? (k & 0x7FFFFFFFu) : ~k;
    return __uint_as_float(u);
}

__global__ void __launch_bounds__(256) minmax_kernel(const float *__restrict__ img, unsigned *__restrict__ lohi_key, int64_t n)
{
    const float *fr = img + (int64_t)blockIdx.y * n;
    unsigned lo = 0xFFFFFFFFu, hi = 0;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        unsigned k = f32_ordered(fr[i]);
        lo = min(lo, k);
        hi = max(hi, k);
    }
    for (int off = 32; off; off >>= 1) {
        lo = min(lo, (unsigned)__shfl_xor((int)lo, off));
        hi = max(hi, (unsigned)__shfl_xor((int)hi, off));
    }
    if (lane_id() == 0) {
        atomicMin(&lohi_key[blockIdx.y * 2 + 0], lo);
        atomicMax(&lohi_key[blockIdx.y * 2 + 1], hi);
    }
}

__global__ void keys_to_float_kernel(unsigned *keys, int count)
{
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < count) keys[i] = __float_as_uint(ordered_f32(keys[i]));
}

// numpy.histogram(float32 image, 256) as skimage.exposure.histogram calls it (scikit-image 0.18.3 / numpy 1.26.4, pinned
// by tests/golden/extensions.npz): edges = float32(i * ((hi - lo) / 256) + lo) computed in float64 (last edge = hi), the
// index trunc(((x - lo) / (hi - lo)) * 256) in float32 arithmetic, 256 -> 255, then one step down / up against the
// float32 edges.  No contraction into FMAs anywhere: the library rounds after every operation.
__device__ __forceinline__ float otsu_edge(int i, float lo, float hi)
{
    if (i >= 256) return hi;
    const double step = __ddiv_rn(__dsub_rn((double)hi, (double)lo), 256.0);
    return (float)__dadd_rn(__dmul_rn((double)i, step), (double)lo);
}

__global__ void __launch_bounds__(256) otsu_hist_kernel(const float *__restrict__ img, const unsigned *__restrict__ lohi_key,
                                                         unsigned long long *__restrict__ hist, int64_t n)
{
    __shared__ unsigned h[256];
    __shared__ float edges[257];
    const int b = blockIdx.y;
    const float *fr = img + (int64_t)b * n;
    const float lo = ordered_f32(lohi_key[b * 2]), hi = ordered_f32(lohi_key[b * 2 + 1]);
    h[threadIdx.x] = 0;
    edges[threadIdx.x] = otsu_edge(threadIdx.x, lo, hi);
    if (threadIdx.x == 0) edges[256] = hi;
    __syncthreads();
    const float denom = __fsub_rn(hi, lo);
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const float x = fr[i];
        int bin = 0;
        if (hi > lo) {
            bin = (int)__fmul_rn(__fdiv_rn(__fsub_rn(x, lo), denom), 256.0f);
            if (bin == 256) bin = 255;
            if (x < edges[bin]) --bin;
            if (x >= edges[bin + 1] && bin != 255) ++bin;
        }
        atomicAdd(&h[bin], 1u);
    }
    __syncthreads();
    if (h[threadIdx.x]) atomicAdd(&hist[b * 256 + threadIdx.x], (unsigned long long)h[threadIdx.x]);
}

// skimage.filters.threshold_otsu on that histogram, one frame per block: bin centres (edge[i] + edge[i + 1]) / 2 in
// float32, class sums as numpy.cumsum builds them -- sequentially, class 1 from the left, class 2 from the RIGHT, in
// float64 -- variance[i] = (w1[i] * w2[i + 1]) * (mean1[i] - mean2[i + 1])^2, first maximum, threshold = that bin's
// centre; a constant frame returns its value.  The two 256-step chains are sequential by definition of the library's
// rounding, so one lane walks them (3 x 256 steps on 256 numbers: microseconds, off any critical path).
__global__ void __launch_bounds__(64) otsu_threshold_kernel(const unsigned long long *__restrict__ hist, const unsigned *__restrict__ lohi_key,
                                                             double *__restrict__ thr)
{
    __shared__ double p[256], cs2[256];
    __shared__ float centers[256];
    __shared__ long long w2[256];
    const int b = blockIdx.x;
    const float lo = ordered_f32(lohi_key[b * 2]), hi = ordered_f32(lohi_key[b * 2 + 1]);
    const unsigned long long *h = hist + (int64_t)b * 256;
    for (int i = threadIdx.x; i < 256; i += 64) {
        centers[i] = __fdiv_rn(__fadd_rn(otsu_edge(i, lo, hi), otsu_edge(i + 1, lo, hi)), 2.0f);
        p[i] = __dmul_rn((double)(long long)h[i], (double)centers[i]);
    }
    __syncthreads();
    if (threadIdx.x != 0) return;
    if (!(hi > lo)) {
        thr[b] = (double)lo;
        return;
    }
    long long acc = 0;
    double facc = 0.0;
    for (int i = 255; i >= 0; --i) {
        acc += (long long)h[i];
        w2[i] = acc;
        facc = __dadd_rn(facc, p[i]);
        cs2[i] = facc;
    }
    acc = 0;
    facc = 0.0;
    double best = 0.0;
    int arg = 0;
    for (int i = 0; i < 255; ++i) {
        acc += (long long)h[i];
        facc = __dadd_rn(facc, p[i]);
        const double m1 = __ddiv_rn(facc, (double)acc), m2 = __ddiv_rn(cs2[i + 1], (double)w2[i + 1]);
        const double d = __dsub_rn(m1, m2);
        const double var = __dmul_rn((double)(acc * w2[i + 1]), __dmul_rn(d, d));
        if (i == 0 || var > best) {
            best = var;
            arg = i;
        }
    }
    thr[b] = (double)centers[arg];
}

}  // namespace pcseg

using namespace pcseg;

extern "C" {

int pcseg_version(void) { return 100; }

void pcseg_timing_enable(int on)
{
    std::lock_guard<std::mutex> lk(g_tmutex);
    if (on) {
        // the event pool is made here, outside any timed region: launches only take slots
        while (g_slots.size() < 4096) {
            TimingSlot t{nullptr, nullptr, nullptr, nullptr};
            if (hipEventCreate(&t.a) != hipSuccess || hipEventCreate(&t.b) != hipSuccess) break;
            g_slots.push_back(t);
        }
    }
    g_timing.store(on != 0);
    g_used = 0;
}

int pcseg_timing_report(char *buf, size_t buf_bytes)
{
    std::lock_guard<std::mutex> lk(g_tmutex);
    std::map<std::string, std::pair<long, double>> agg;
    for (size_t i = 0; i < g_used; ++i) {
        if (hipEventSynchronize(g_slots[i].b) != hipSuccess) continue;
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, g_slots[i].a, g_slots[i].b) != hipSuccess) continue;
        // kernel name + the template arguments of the launching function ("[Fg = ..., Epi = ...]")
        std::string key = g_slots[i].name;
        const char *br = strchr(g_slots[i].where, '[');
        if (br) key += std::string(" ") + br;
        auto &e = agg[key];
        e.first += 1;
        e.second += ms;
    }
    std::string out;
    char line[1024];
    for (auto &kv : agg) {
        snprintf(line, sizeof line, "%s\t%ld\t%.6f\n", kv.first.c_str(), kv.second.first, kv.second.second);
        out += line;
    }
    if (!buf || buf_bytes == 0) return (int)out.size() + 1;
    g_used = 0;  // records are consumed only when they are actually written out
    size_t nbytes = out.size() < buf_bytes - 1 ? out.size() : buf_bytes - 1;
    memcpy(buf, out.data(), nbytes);
    buf[nbytes] = 0;
    return (int)nbytes;
}

const char *pcseg_last_error(void) { return g_err; }

int pcseg_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

int pcseg_argmax_planes_f32(const float *stack, uint8_t *cls, int B, int C, int H, int W, pcseg_stream_t stream)
{
    PCSEG_REQUIRE(stack && cls && C >= 1 && C <= 255 && check_shape(B, H, W), "bad arguments");
    int64_t n = (int64_t)H * W;
    hipStream_t s = (hipStream_t)stream;
    if ((n & 3) == 0) {
        dim3 grid((unsigned)((n / 4 + 255) / 256), B);
        PCSEG_LAUNCH(argmax_kernel<true>, grid, dim3(256), 0, s, stack, cls, C, n);
    } else {
        dim3 grid((unsigned)((n + 255) / 256), B);
        PCSEG_LAUNCH(argmax_kernel<false>, grid, dim3(256), 0, s, stack, cls, C, n);
    }
    PCSEG_CHECK_LAUNCH();
    return PCSEG_OK;
}

int pcseg_median5_u8(const uint8_t *in, uint8_t *out, int B, int H, int W, pcseg_stream_t stream)
{
    PCSEG_REQUIRE(in && out && in != out && check_shape(B, H, W), "bad arguments");
    dim3 grid((W + MED_TW - 1) / MED_TW, (H + MED_TH - 1) / MED_TH, B);
    PCSEG_LAUNCH(median5_kernel, grid, dim3(256), 0, (hipStream_t)stream, in, out, H, W);
    PCSEG_CHECK_LAUNCH();
    return PCSEG_OK;
}

int pcseg_threshold_lt_f32(const float *img, float threshold, uint8_t *mask, int B, int H, int W, pcseg_stream_t stream)
{
    PCSEG_REQUIRE(img && mask && check_shape(B, H, W), "bad arguments");
    int64_t total = (int64_t)B * H * W;
    unsigned blocks = (unsigned)((total + 1023) / 1024);
    PCSEG_LAUNCH(threshold_lt_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, img, threshold, mask, total);
    PCSEG_CHECK_LAUNCH();
    return PCSEG_OK;
}

int pcseg_morph3x3(const uint8_t *mask, uint8_t *out, int erode, int B, int H, int W, pcseg_stream_t stream)
{
    PCSEG_REQUIRE(mask && out && mask != out && check_shape(B, H, W), "bad arguments");
    dim3 grid((W + 63) / 64, (H + 3) / 4, B);
    PCSEG_LAUNCH(morph3x3_kernel, grid, dim3(256), 0, (hipStream_t)stream, mask, out, erode, H, W);
    PCSEG_CHECK_LAUNCH();
    return PCSEG_OK;
}

int pcseg_otsu_hist_f32(const float *img, int64_t *hist, float *lohi, int B, int H, int W, pcseg_stream_t stream)
{
    PCSEG_REQUIRE(img && hist && lohi && check_shape(B, H, W), "bad arguments");
    hipStream_t s = (hipStream_t)stream;
    int64_t n = (int64_t)H * W;
    // the float32 lo/hi buffer doubles as the ordered-key scratch of the min/max pass
    unsigned *keys = reinterpret_cast<unsigned *>(lohi);
    PCSEG_CHECK_HIP(hipMemsetAsync(hist, 0, sizeof(int64_t) * 256 * B, s));
    // init lo keys to 0xFFFFFFFF and hi keys to 0: memset pattern per 8 bytes
    PCSEG_CHECK_HIP(hipMemsetAsync(keys, 0, sizeof(unsigned) * 2 * B, s));
    for (int b = 0; b < B; ++b) PCSEG_CHECK_HIP(hipMemsetAsync(keys + 2 * b, 0xFF, sizeof(unsigned), s));
    unsigned gx = (unsigned)((n + 256 * 16 - 1) / (256 * 16));
    if (gx > 1024) gx = 1024;
    PCSEG_LAUNCH(minmax_kernel, dim3(gx, B), dim3(256), 0, s, img, keys, n);
    PCSEG_CHECK_LAUNCH();
    PCSEG_LAUNCH(otsu_hist_kernel, dim3(gx, B), dim3(256), 0, s, img, keys, (unsigned long long *)hist, n);
    PCSEG_CHECK_LAUNCH();
    // keys -> float32 lo/hi in place, after every histogram block has read them (stream order)
    PCSEG_LAUNCH(keys_to_float_kernel, dim3((2 * B + 63) / 64), dim3(64), 0, s, keys, 2 * B);
    PCSEG_CHECK_LAUNCH();
    return PCSEG_OK;
}

int pcseg_otsu_f32(const float *img, double *threshold, int64_t *hist, float *lohi, int B, int H, int W, pcseg_stream_t stream)
{
    PCSEG_REQUIRE(img && threshold && hist && lohi && check_shape(B, H, W), "bad arguments");
    hipStream_t s = (hipStream_t)stream;
    int64_t n = (int64_t)H * W;
    unsigned *keys = reinterpret_cast<unsigned *>(lohi);
    PCSEG_CHECK_HIP(hipMemsetAsync(hist, 0, sizeof(int64_t) * 256 * B, s));
    PCSEG_CHECK_HIP(hipMemsetAsync(keys, 0, sizeof(unsigned) * 2 * B, s));
    for (int b = 0; b < B; ++b) PCSEG_CHECK_HIP(hipMemsetAsync(keys + 2 * b, 0xFF, sizeof(unsigned), s));
    unsigned gx = (unsigned)((n + 256 * 16 - 1) / (256 * 16));
    if (gx > 1024) gx = 1024;
    PCSEG_LAUNCH(minmax_kernel, dim3(gx, B), dim3(256), 0, s, img, keys, n);
    PCSEG_CHECK_LAUNCH();
    PCSEG_LAUNCH(otsu_hist_kernel, dim3(gx, B), dim3(256), 0, s, img, keys, (unsigned long long *)hist, n);
    PCSEG_CHECK_LAUNCH();
    PCSEG_LAUNCH(otsu_threshold_kernel, dim3(B), dim3(64), 0, s, (const unsigned long long *)hist, (const unsigned *)keys, threshold);
    PCSEG_CHECK_LAUNCH();
    PCSEG_LAUNCH(keys_to_float_kernel, dim3((2 * B + 63) / 64), dim3(64), 0, s, keys, 2 * B);
    PCSEG_CHECK_LAUNCH();
    return PCSEG_OK;
}

}  // extern "C"
