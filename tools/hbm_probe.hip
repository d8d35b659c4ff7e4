// HBM probe for the rollout kernels' traffic shape: what does MI355X sustain for write-only streams, compared with reads?
// Build: hipcc --offload-arch=gfx950 -O3 -o gpurun_out/hbm_probe tools/hbm_probe.hip      Run: gpurun_out/hbm_probe
// Every kernel moves the same number of bytes as one rollout launch of T steps over N envs with F+2 output planes
// ([T][F][N] float32 obs, [T][N] float32 rewards, [T][N] bytes) unless stated otherwise.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

// one lane per env, T steps, F planes per step: dword stores, lanes consecutive (the rollout kernels' store pattern)
__global__ void __launch_bounds__(512) k_planes_dword(float* __restrict__ out, int N, int F, int T, float v)
{
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= N) return;
    for (int t = 0; t < T; t++) {
        float* o = out + (size_t)t * F * N + e;
#pragma unroll 4
        for (int q = 0; q < F; q++) o[(size_t)q * N] = v + q;
    }
}

// same bytes, but each lane owns 4 consecutive envs: dwordx4 stores
__global__ void __launch_bounds__(512) k_planes_x4(float4* __restrict__ out, int N4, int F, int T, float v)
{
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= N4) return;
    for (int t = 0; t < T; t++) {
        float4* o = out + (size_t)t * F * N4 + e;
#pragma unroll 4
        for (int q = 0; q < F; q++) o[(size_t)q * N4] = make_float4(v + q, v, v, v);
    }
}

// wave-uniform pseudo-random delay: desynchronises the waves the way data-dependent work does in the real kernels
__device__ __forceinline__ void drift(int t, int spread)
{
    if (spread <= 0) return;
    unsigned h = (unsigned)(blockIdx.x * 8 + threadIdx.x / 64) * 2654435761u + (unsigned)t * 40503u;
    h ^= h >> 15; h *= 2246822519u; h ^= h >> 13;
    const int n = (int)(h % (unsigned)spread);
    for (int q = 0; q < n; q++) __builtin_amdgcn_s_sleep(8);      // 8 * 64 cycles
}

// the plane pattern with drifting waves
__global__ void __launch_bounds__(512) k_planes_drift(float* __restrict__ out, int N, int F, int T, float v, int spread)
{
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    for (int t = 0; t < T; t++) {
        drift(t, spread);
        float* o = out + (size_t)t * F * N + e;
#pragma unroll 4
        for (int q = 0; q < F; q++) __builtin_nontemporal_store(v + q, &o[(size_t)q * N]);
    }
}

// env-major rows: every wave owns 64 envs = one contiguous block of 64 * F floats per step, written as dwordx4
__global__ void __launch_bounds__(512) k_rows_drift(float* __restrict__ out, int N, int F, int T, float v, int spread)
{
    const int wave = (blockIdx.x * blockDim.x + threadIdx.x) / 64, lane = threadIdx.x & 63;
    const int n4 = 64 * F / 4;                                    // float4 per wave block
    for (int t = 0; t < T; t++) {
        drift(t, spread);
        typedef float vf4 __attribute__((ext_vector_type(4)));
        vf4* o = (vf4*)(out + (size_t)t * F * N + (size_t)wave * 64 * F);
        const vf4 x = {v, v, v, v};
        for (int q = lane; q < n4; q += 64) __builtin_nontemporal_store(x, &o[q]);
    }
}

// grid-stride linear fill with dwordx4
__global__ void __launch_bounds__(256) k_fill_x4(float4* __restrict__ out, size_t n4, float v)
{
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x)
        out[i] = make_float4(v, v, v, v);
}

// grid-stride linear read with dwordx4 (sum kept alive through a never-taken store)
__global__ void __launch_bounds__(256) k_read_x4(const float4* __restrict__ in, size_t n4, float* __restrict__ sink)
{
    float acc = 0.f;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
        const float4 x = in[i];
        acc += x.x + x.y + x.z + x.w;
    }
    if (acc == 123456.789f) *sink = acc;
}

__global__ void __launch_bounds__(256) k_copy_x4(const float4* __restrict__ in, float4* __restrict__ out, size_t n4)
{
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) out[i] = in[i];
}

template <typename F>
static double time_ms(F&& launch, int reps)
{
    hipEvent_t a, b;
    CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    launch();
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(a, 0));
    for (int r = 0; r < reps; r++) launch();
    CK(hipEventRecord(b, 0));
    CK(hipEventSynchronize(b));
    float ms = 0.f;
    CK(hipEventElapsedTime(&ms, a, b));
    CK(hipEventDestroy(a)); CK(hipEventDestroy(b));
    return ms / reps;
}

int main(int argc, char** argv)
{
    const int F = 37, T = 200;
    const int Ns[] = {65536, 262144, 1048576};
    size_t max_bytes = (size_t)60 * F * 1048576 * 4;     // >= 200 steps x 262144 envs as well
    float *buf, *buf2, *sink;
    CK(hipMalloc(&buf, max_bytes)); CK(hipMalloc(&buf2, max_bytes)); CK(hipMalloc(&sink, 4));
    CK(hipMemset(buf, 0, max_bytes)); CK(hipMemset(buf2, 0, max_bytes));
    printf("%-34s %10s %10s %10s\n", "kernel", "N", "ms", "TB/s");
    for (int N : Ns) {
        const int Tn = (N == 1048576) ? 60 : T;          // keep the largest case inside the buffer and the run short
        const size_t bytes = (size_t)Tn * F * N * 4;
        const int reps = 5;
        double ms;
        for (int bs : {256, 512}) {
            ms = time_ms([&] { hipLaunchKernelGGL(k_planes_dword, dim3((N + bs - 1) / bs), dim3(bs), 0, 0, buf, N, F, Tn, 1.f); }, reps);
            printf("planes dword  (block %3d)          %10d %10.3f %10.3f\n", bs, N, ms, bytes / ms * 1e-9);
        }
        for (int spread : {0, 4, 16}) {
            ms = time_ms([&] { hipLaunchKernelGGL(k_planes_drift, dim3(N / 256), dim3(256), 0, 0, buf, N, 36, Tn, 1.f, spread); }, reps);
            printf("planes nt dword, drift %2d           %10d %10.3f %10.3f\n", spread, N, ms, (size_t)Tn * 36 * N * 4 / ms * 1e-9);
            ms = time_ms([&] { hipLaunchKernelGGL(k_rows_drift, dim3(N / 256), dim3(256), 0, 0, buf, N, 36, Tn, 1.f, spread); }, reps);
            printf("rows nt dwordx4,  drift %2d           %10d %10.3f %10.3f\n", spread, N, ms, (size_t)Tn * 36 * N * 4 / ms * 1e-9);
        }
        if (N > 65536) {   // a 65 536-env slice of the wider batch: every plane is written only in part (plane stride N * 4 bytes)
            const size_t sl = (size_t)Tn * F * 65536 * 4;
            ms = time_ms([&] { hipLaunchKernelGGL(k_planes_dword, dim3(65536 / 256), dim3(256), 0, 0, buf, N, F, Tn, 1.f); }, reps);
            printf("planes dword, first 65536 envs     %10d %10.3f %10.3f\n", N, ms, sl / ms * 1e-9);
            ms = time_ms([&] { for (int c = 0; c < N / 65536; c++) hipLaunchKernelGGL(k_planes_dword, dim3(65536 / 256), dim3(256), 0, 0, buf + (size_t)c * 65536, N, F, Tn, 1.f); }, reps);
            printf("planes dword, 65536-env slices     %10d %10.3f %10.3f\n", N, ms, bytes / ms * 1e-9);
        }
        ms = time_ms([&] { hipLaunchKernelGGL(k_planes_x4, dim3((N / 4 + 255) / 256), dim3(256), 0, 0, (float4*)buf, N / 4, F, Tn, 1.f); }, reps);
        printf("planes dwordx4 (4 envs per lane)   %10d %10.3f %10.3f\n", N, ms, bytes / ms * 1e-9);
        for (int g : {1024, 4096, 16384}) {
            ms = time_ms([&] { hipLaunchKernelGGL(k_fill_x4, dim3(g), dim3(256), 0, 0, (float4*)buf, bytes / 16, 1.f); }, reps);
            printf("linear fill dwordx4 (grid %5d)    %10d %10.3f %10.3f\n", g, N, ms, bytes / ms * 1e-9);
        }
        ms = time_ms([&] { CK(hipMemsetAsync(buf, 0, bytes, 0)); }, reps);
        printf("hipMemsetAsync                     %10d %10.3f %10.3f\n", N, ms, bytes / ms * 1e-9);
        ms = time_ms([&] { hipLaunchKernelGGL(k_read_x4, dim3(4096), dim3(256), 0, 0, (const float4*)buf, bytes / 16, sink); }, reps);
        printf("linear read dwordx4 (grid  4096)   %10d %10.3f %10.3f\n", N, ms, bytes / ms * 1e-9);
        ms = time_ms([&] { hipLaunchKernelGGL(k_copy_x4, dim3(4096), dim3(256), 0, 0, (const float4*)buf, (float4*)buf2, bytes / 16); }, reps);
        printf("linear copy dwordx4 (read+write)   %10d %10.3f %10.3f\n", N, ms, 2.0 * bytes / ms * 1e-9);
    }
    CK(hipFree(buf)); CK(hipFree(buf2)); CK(hipFree(sink));
    return 0;
}
