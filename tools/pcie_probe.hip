// pcie_probe.hip -- how fast can one vector step's outputs reach pinned host memory?  (diagnostic, NOT product code)
//
// PtGVecEnv.step at 65 536 envs spends 0.40 ms on ONE hipMemcpyAsync of 9.5 MB (24 GB/s) behind a 6 us kernel.  Candidates:
//   1  hipMemcpyAsync device -> pinned host, whole block on one stream                      (what ptg_step_host does today)
//   2  the same block cut into 2 / 4 / 8 pieces on as many streams                           (several SDMA engines at once)
//   3  a copy KERNEL: reads the device block, writes the device-mapped host block            (shader copy, grid sweep)
//   4  a kernel that WRITES the host-mapped block directly                                   (zero copy: what the step kernel would do)
// Build: hipcc --offload-arch=gfx950 -O3 -o tools/bin/pcie_probe tools/pcie_probe.hip      Run: tools/bin/pcie_probe [MB]
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s failed: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
typedef float vf4 __attribute__((ext_vector_type(4)));

__global__ void __launch_bounds__(256) k_copy(const vf4* __restrict__ src, vf4* __restrict__ dst, size_t n16, int nt)
{
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t g = (size_t)blockIdx.x * blockDim.x + threadIdx.x; g < n16; g += stride) {
        const vf4 v = src[g];
        if (nt) __builtin_nontemporal_store(v, dst + g); else dst[g] = v;
    }
}

__global__ void __launch_bounds__(256) k_fill(vf4* __restrict__ dst, size_t n16, float x, int nt)
{
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    const vf4 v = {x, x + 1.f, x + 2.f, x + 3.f};
    for (size_t g = (size_t)blockIdx.x * blockDim.x + threadIdx.x; g < n16; g += stride) {
        if (nt) __builtin_nontemporal_store(v, dst + g); else dst[g] = v;
    }
}

static double now_us() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

int main(int argc, char** argv)
{
    const double mb = argc > 1 ? atof(argv[1]) : 9.77;       // 149 B x 65 536
    const size_t bytes = (size_t)(mb * 1e6) / 4096 * 4096, n16 = bytes / 16;
    CK(hipSetDevice(0));
    void *dev, *host, *host_dev;
    CK(hipMalloc(&dev, bytes));
    CK(hipHostMalloc(&host, bytes, hipHostMallocMapped));
    CK(hipHostGetDevicePointer(&host_dev, host, 0));
    CK(hipMemset(dev, 1, bytes));
    memset(host, 0, bytes);
    const int NS = 8, REP = 30;
    hipStream_t st[NS];
    for (int q = 0; q < NS; q++) CK(hipStreamCreateWithFlags(&st[q], hipStreamNonBlocking));
    printf("block %.2f MB, pinned + device-mapped host memory; wall clock per repetition (launch .. synchronise), best / median of %d\n", bytes / 1e6, REP);
    auto report = [&](const char* name, std::vector<double>& v) {
        std::sort(v.begin(), v.end());
        printf("%-78s %8.1f / %8.1f us  = %5.1f GB/s\n", name, v[0], v[v.size() / 2], bytes / v[v.size() / 2] / 1e3);
        fflush(stdout);
    };
    char name[200];
    for (int parts : {1, 2, 4, 8}) {
        std::vector<double> v;
        for (int r = 0; r < REP; r++) {
            const double t0 = now_us();
            const size_t per = (bytes / parts + 4095) / 4096 * 4096;
            for (int q = 0; q < parts; q++) {
                const size_t off = q * per, len = off >= bytes ? 0 : std::min(per, bytes - off);
                if (len) CK(hipMemcpyAsync((char*)host + off, (char*)dev + off, len, hipMemcpyDeviceToHost, st[q]));
            }
            for (int q = 0; q < parts; q++) CK(hipStreamSynchronize(st[q]));
            v.push_back(now_us() - t0);
        }
        snprintf(name, sizeof name, "hipMemcpyAsync D2H, %d piece(s) on %d stream(s)", parts, parts);
        report(name, v);
    }
    for (int nt : {0, 1})
        for (int grid : {16, 64, 256, 1024}) {
            std::vector<double> v;
            for (int r = 0; r < REP; r++) {
                const double t0 = now_us();
                hipLaunchKernelGGL(k_copy, dim3(grid), dim3(256), 0, st[0], (const vf4*)dev, (vf4*)host_dev, n16, nt);
                CK(hipStreamSynchronize(st[0]));
                v.push_back(now_us() - t0);
            }
            snprintf(name, sizeof name, "copy kernel device -> mapped host, %d x 256 threads, %s stores", grid, nt ? "non-temporal" : "plain");
            report(name, v);
        }
    for (int nt : {0, 1})
        for (int grid : {64, 256}) {
            std::vector<double> v;
            for (int r = 0; r < REP; r++) {
                const double t0 = now_us();
                hipLaunchKernelGGL(k_fill, dim3(grid), dim3(256), 0, st[0], (vf4*)host_dev, n16, (float)r, nt);
                CK(hipStreamSynchronize(st[0]));
                v.push_back(now_us() - t0);
            }
            snprintf(name, sizeof name, "kernel writes the mapped host block directly, %d x 256 threads, %s stores", grid, nt ? "non-temporal" : "plain");
            report(name, v);
        }
    {   // two halves: copy kernel for one, SDMA for the other, at the same time
        std::vector<double> v;
        const size_t half = bytes / 2 / 4096 * 4096;
        for (int r = 0; r < REP; r++) {
            const double t0 = now_us();
            hipLaunchKernelGGL(k_copy, dim3(256), dim3(256), 0, st[0], (const vf4*)dev, (vf4*)host_dev, half / 16, 1);
            CK(hipMemcpyAsync((char*)host + half, (char*)dev + half, bytes - half, hipMemcpyDeviceToHost, st[1]));
            CK(hipStreamSynchronize(st[0])); CK(hipStreamSynchronize(st[1]));
            v.push_back(now_us() - t0);
        }
        report("half by copy kernel + half by hipMemcpyAsync, concurrently", v);
    }
    // H2D of the action block (256 KB) for completeness
    {
        std::vector<double> v;
        for (int r = 0; r < REP; r++) {
            const double t0 = now_us();
            CK(hipMemcpyAsync(dev, host, 65536 * 4, hipMemcpyHostToDevice, st[0]));
            CK(hipStreamSynchronize(st[0]));
            v.push_back(now_us() - t0);
        }
        std::sort(v.begin(), v.end());
        printf("%-78s %8.1f / %8.1f us\n", "hipMemcpyAsync H2D of the 256 KB action block", v[0], v[v.size() / 2]);
    }
    return 0;
}
