// What bounds a plain int4 copy of a 64 x 1024^2 int32 image: access shapes of the label passes, timed with HIP events.
//   hipcc --offload-arch=gfx950 -O3 profiles/r03/copy_shapes.hip -o ab/copy_shapes (build container), then ab/copy_shapes on the GPU box
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

template <int Q>
__global__ void __launch_bounds__(256) copy_quads(const int *__restrict__ in, int *out, int64_t n)
{
    const int b = blockIdx.y;
    const int *src = in + (int64_t)b * n;
    int *dst = out + (int64_t)b * n;
    const int64_t i0 = (int64_t)blockIdx.x * (1024 * Q) + threadIdx.x * 4;
    int4 v[Q];
#pragma unroll
    for (int q = 0; q < Q; ++q) v[q] = *reinterpret_cast<const int4 *>(src + i0 + q * 1024);
#pragma unroll
    for (int q = 0; q < Q; ++q) {
        v[q].x += 1;
        *reinterpret_cast<int4 *>(dst + i0 + q * 1024) = v[q];
    }
}
// the same with the label passes' extra: every lane also reads ONE word of `out` at a data-dependent place first
template <int Q>
__global__ void __launch_bounds__(256) copy_quads_gather(const int *__restrict__ in, int *out, int64_t n)
{
    const int b = blockIdx.y;
    const int *src = in + (int64_t)b * n;
    int *dst = out + (int64_t)b * n;
    const int64_t i0 = (int64_t)blockIdx.x * (1024 * Q) + threadIdx.x * 4;
    int4 v[Q];
#pragma unroll
    for (int q = 0; q < Q; ++q) v[q] = *reinterpret_cast<const int4 *>(src + i0 + q * 1024);
    int g[Q];
#pragma unroll
    for (int q = 0; q < Q; ++q) g[q] = __hip_atomic_load(dst + (v[q].x & 0xFFFFF), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
#pragma unroll
    for (int q = 0; q < Q; ++q) {
        v[q].x += g[q];
        *reinterpret_cast<int4 *>(dst + i0 + q * 1024) = v[q];
    }
}

int main()
{
    const int B = 64;
    const int64_t n = 1024 * 1024;
    int *a, *b;
    CK(hipMalloc(&a, sizeof(int) * B * n));
    CK(hipMalloc(&b, sizeof(int) * B * n));
    CK(hipMemset(a, 0, sizeof(int) * B * n));
    CK(hipMemset(b, 0, sizeof(int) * B * n));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    auto run = [&](const char *name, auto launch) {
        for (int i = 0; i < 3; ++i) launch();
        hipEventRecord(e0, 0);
        for (int i = 0; i < 20; ++i) launch();
        hipEventRecord(e1, 0);
        hipEventSynchronize(e1);
        float ms = 0;
        hipEventElapsedTime(&ms, e0, e1);
        printf("%-44s %7.1f us  %.2f TB/s\n", name, ms / 20 * 1e3, 2.0 * B * n * 4 / (ms / 20 * 1e-3) / 1e12);
        return 0;
    };
    run("1 quad / lane, grid (1024, 64)", [&] { hipLaunchKernelGGL(copy_quads<1>, dim3(1024, B), dim3(256), 0, 0, a, b, n); });
    run("2 quads / lane, grid (512, 64)", [&] { hipLaunchKernelGGL(copy_quads<2>, dim3(512, B), dim3(256), 0, 0, a, b, n); });
    run("4 quads / lane, grid (256, 64)", [&] { hipLaunchKernelGGL(copy_quads<4>, dim3(256, B), dim3(256), 0, 0, a, b, n); });
    run("8 quads / lane, grid (128, 64)", [&] { hipLaunchKernelGGL(copy_quads<8>, dim3(128, B), dim3(256), 0, 0, a, b, n); });
    run("4 quads + a gather from the output image", [&] { hipLaunchKernelGGL(copy_quads_gather<4>, dim3(256, B), dim3(256), 0, 0, a, b, n); });
    run("1 quad + a gather from the output image", [&] { hipLaunchKernelGGL(copy_quads_gather<1>, dim3(1024, B), dim3(256), 0, 0, a, b, n); });
    return 0;
}
