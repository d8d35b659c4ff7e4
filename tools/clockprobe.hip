// clockprobe.hip -- diagnostic helpers for tools/coldstart.py (NOT part of the product library).
//   clk_probe   one 64-lane workgroup per CU spins for `spin_cycles` shader cycles and records (s_memtime, s_memrealtime)
//               before and after: shader clock = d(memtime) / d(memrealtime) x 100 MHz (MI355X_MICROARCH.md, DVFS item 6).
//   bw_write    a pure write stream (dwordx4 per lane, non-temporal) over `bytes`;  bw_read  a pure dwordx4 read stream.
// Build: hipcc --offload-arch=gfx950 -O3 -shared -fPIC -o tools/libclockprobe.so tools/clockprobe.hip
#include <hip/hip_runtime.h>
#include <cstdint>

namespace {

__global__ void __launch_bounds__(64) k_clk(unsigned long long* out, int spin_cycles)
{
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    const unsigned long long r0 = __builtin_amdgcn_s_memrealtime();
    unsigned long long t1 = t0;
    while ((long long)(t1 - t0) < (long long)spin_cycles) {
        __builtin_amdgcn_s_sleep(2);
        t1 = __builtin_amdgcn_s_memtime();
    }
    const unsigned long long r1 = __builtin_amdgcn_s_memrealtime();
    if (threadIdx.x == 0) {
        out[blockIdx.x * 4 + 0] = t0; out[blockIdx.x * 4 + 1] = t1;
        out[blockIdx.x * 4 + 2] = r0; out[blockIdx.x * 4 + 3] = r1;
    }
}

typedef float vf4 __attribute__((ext_vector_type(4)));

__global__ void __launch_bounds__(256) k_write(vf4* p, size_t n4, float v)
{
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    const vf4 x = {v, v, v, v};
    for (size_t g = (size_t)blockIdx.x * blockDim.x + threadIdx.x; g < n4; g += stride) __builtin_nontemporal_store(x, p + g);
}

__global__ void __launch_bounds__(256) k_read(const vf4* p, size_t n4, float* sink)
{
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    vf4 acc = {0.f, 0.f, 0.f, 0.f};
    for (size_t g = (size_t)blockIdx.x * blockDim.x + threadIdx.x; g < n4; g += stride) acc += __builtin_nontemporal_load(p + g);
    if (acc.x + acc.y + acc.z + acc.w == 12345.678f) *sink = acc.x;
}

// entry skew of a grid: every workgroup stamps the 100 MHz clock when its first wave starts (and uses `vregs`-ish registers / LDS as asked)
__global__ void k_entry(unsigned long long* out)
{
    extern __shared__ unsigned char dyn[];
    if (threadIdx.x == 0) { out[blockIdx.x] = __builtin_amdgcn_s_memrealtime(); if (dyn[0] == 123) out[blockIdx.x] = 0; }
}

// a chain of dependent launches of the step kernel's shape (256 x 256 threads): what does ONE launch cost at least when it
// moves `n16` 16-byte pieces (kind 1: written; kind 2: one sixth read first, the rest written) -- or nothing at all (kind 0)?
__global__ void __launch_bounds__(256) k_chain(vf4* p, const vf4* q, size_t n16, int kind)
{
    if (kind == 0) return;
    const size_t stride = (size_t)gridDim.x * blockDim.x, g0 = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    vf4 x = {1.f, 2.f, 3.f, 4.f};
    if (kind == 2) for (size_t g = g0; g < n16 / 6; g += stride) x += q[g];
    for (size_t g = g0; g < n16; g += stride) __builtin_nontemporal_store(x, p + g);
}

// feature-major store pattern of the fused rollout: every wave owns 64 envs and writes, per step, F columns of 64 x `W` bytes at a
// column stride of N x W bytes ([T][F][N] output).  rot = 0: every workgroup walks the columns in the same order (0, 1, 2, ...);
// rot > 0: workgroup b starts at column (b x rot) % F.  Is the order what a power-of-two column stride trips over?
template <typename V>
__global__ void __launch_bounds__(256) k_fm(V* p, int N, int F, int T, int rot)
{
    const int e = blockIdx.x * 256 + threadIdx.x;
    const int q0 = rot ? (int)((blockIdx.x * (unsigned)rot) % (unsigned)F) : 0;
    V v; __builtin_memset(&v, 0, sizeof v);
    for (int t = 0; t < T; t++) {
        V* row = p + (size_t)t * F * N + e;
        int q = q0;
        for (int i = 0; i < F; i++) {
            __builtin_nontemporal_store(v, row + (size_t)q * N);
            q = q + 1 == F ? 0 : q + 1;
        }
    }
}

}  // namespace

extern "C" {

// us per step of the pattern above (W = 4 or 8 bytes per env and column), T steps in one launch of N / 256 workgroups
double fm_probe(void* buf, int N, int F, int T, int W, int rot)
{
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    float best = 1e30f;
    for (int rep = 0; rep < 4; rep++) {
        (void)hipEventRecord(e0, 0);
        if (W == 8) hipLaunchKernelGGL(k_fm<double>, dim3(N / 256), dim3(256), 0, 0, (double*)buf, N, F, T, rot);
        else hipLaunchKernelGGL(k_fm<float>, dim3(N / 256), dim3(256), 0, 0, (float*)buf, N, F, T, rot);
        (void)hipEventRecord(e1, 0); (void)hipEventSynchronize(e1);
        float ms = 0.f; (void)hipEventElapsedTime(&ms, e0, e1);
        if (rep && ms < best) best = ms;
    }
    (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
    return (double)best * 1e3 / T;
}

double chain_probe(void* buf, void* buf2, size_t bytes, int kind, int n_launch)
{
    hipStream_t st;
    if (hipStreamCreate(&st) != hipSuccess) return -1.0;
    hipGraph_t g; hipGraphExec_t ge; hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    (void)hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal);
    for (int i = 0; i < n_launch; i++) hipLaunchKernelGGL(k_chain, dim3(256), dim3(256), 0, st, (vf4*)buf, (const vf4*)buf2, bytes / 16, kind);
    if (hipStreamEndCapture(st, &g) != hipSuccess || hipGraphInstantiate(&ge, g, nullptr, nullptr, 0) != hipSuccess) return -2.0;
    (void)hipGraphLaunch(ge, st); (void)hipStreamSynchronize(st);       // warm
    (void)hipEventRecord(e0, st); (void)hipGraphLaunch(ge, st); (void)hipEventRecord(e1, st); (void)hipStreamSynchronize(st);
    float ms = 0.f;
    (void)hipEventElapsedTime(&ms, e0, e1);
    (void)hipGraphExecDestroy(ge); (void)hipGraphDestroy(g); (void)hipEventDestroy(e0); (void)hipEventDestroy(e1); (void)hipStreamDestroy(st);
    return (double)ms * 1e3 / n_launch;
}

int entry_probe(void* stream, unsigned long long* out_dev, int n_wg, int block, int lds_bytes)
{
    (void)hipFuncSetAttribute((const void*)k_entry, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 512);
    hipLaunchKernelGGL(k_entry, dim3(n_wg), dim3(block), lds_bytes, (hipStream_t)stream, out_dev);
    return (int)hipGetLastError();
}


int clk_probe(void* stream, unsigned long long* out_dev, int n_wg, int spin_cycles)
{
    hipLaunchKernelGGL(k_clk, dim3(n_wg), dim3(64), 0, (hipStream_t)stream, out_dev, spin_cycles);
    return (int)hipGetLastError();
}

int bw_write(void* stream, void* buf, size_t bytes)
{
    hipLaunchKernelGGL(k_write, dim3(2048), dim3(256), 0, (hipStream_t)stream, (vf4*)buf, bytes / 16, 1.0f);
    return (int)hipGetLastError();
}

int bw_read(void* stream, const void* buf, size_t bytes, float* sink)
{
    hipLaunchKernelGGL(k_read, dim3(2048), dim3(256), 0, (hipStream_t)stream, (const vf4*)buf, bytes / 16, sink);
    return (int)hipGetLastError();
}

}
