// persist_probe.hip -- what would a PERSISTENT ptg_step cost?  (diagnostic, NOT part of the product library)
//
// The launch-per-step path (k_step_hot) costs 5.1 us per step at 65 536 envs; an empty dependent launch costs 1.5 us of that and the
// step's 13.96 MB written by a kernel of that shape 3.3 us (profiles/r02_step_floor.txt).  The alternative without a kernel boundary:
// a resident env kernel (state in registers, so only the 149 B per env-step of the rollout form leave it) that waits for a mailbox word
// per step, written by whoever produces the actions.  This probe times the LOWER BOUND of that form -- an env kernel that does nothing
// but wait, read its actions, write the step's output bytes and signal -- in the two shapes a policy can have:
//   A  "kernel per step": a stand-in policy kernel is launched once per step (what a torch policy is: kernels with boundaries).  It waits
//      until all 256 env workgroups have signalled the previous step (fan-in on one counter), reads one observation word per env, writes
//      the actions, and its last workgroup releases the step's mailbox word (the flag every env workgroup polls).
//   B  "tile-local": a resident stand-in policy whose workgroup j serves env workgroups 4j .. 4j+3 only -- per-tile flags both ways, no
//      grid-wide fan-in at all (the best case: a policy fused into one persistent kernel that never mixes tiles).
// Every spin loop is bounded by the 100 MHz clock (abort after `limit_ms`), so a scheduling surprise ends the kernels instead of hanging.
// Hand-off protocol (MI355X_MICROARCH.md, "Valid forms" R1 / handoff-flag): every handed-off byte is written with a 16-byte `sc1`
// (write-through) store, every storing wave drains (`s_waitcnt vmcnt(0)`), the workgroup's barrier, then ONE lane signals with a relaxed
// agent-scope atomic; the consumer polls with `sc1` loads, a workgroup barrier, then reads the bytes with `sc1` loads.  No release /
// acquire fences: polling with acquire loads and a __threadfence() per step (whole-L2 write-back + invalidate per workgroup and step)
// measured 86 / 102 us per step in the first version of this probe.  `persist_probe <steps> fence` brings that version back.
//
// Build: hipcc --offload-arch=gfx950 -O3 -o tools/bin/persist_probe tools/persist_probe.hip        Run: tools/bin/persist_probe [steps]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include <algorithm>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s failed: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

typedef float vf4 __attribute__((ext_vector_type(4)));

constexpr int N = 65536, F = 35, ENV_WG = 256, ENV_T = 256, POL_WG = 64, POL_T = 256;
constexpr unsigned long long TICKS_PER_MS = 100000ull;

struct Ctl {
    unsigned flag;            // A: step whose actions are ready (t + 1)
    unsigned done_ctr;        // A: env workgroups that have finished a step, cumulative
    unsigned pol_ctr;         // A: policy workgroups that have written their actions, cumulative
    unsigned abort_;          // set by any spin loop that ran out of time
    unsigned pad[12];
    unsigned tile_ready[ENV_WG];      // B: per env workgroup, step whose actions are ready
    unsigned tile_done[ENV_WG];       // B: per env workgroup, steps finished
};

template <bool FENCE> __device__ __forceinline__ unsigned ld_flag(const unsigned* p)
{
    return FENCE ? __hip_atomic_load(p, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) : __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
template <bool FENCE> __device__ __forceinline__ void st_flag(unsigned* p, unsigned v)
{
    if (FENCE) __hip_atomic_store(p, v, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    else __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
template <bool FENCE> __device__ __forceinline__ unsigned add_flag(unsigned* p)
{
    return FENCE ? __hip_atomic_fetch_add(p, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT) : __hip_atomic_fetch_add(p, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// the handed-off bytes: FENCE = plain / non-temporal accesses ordered by fences; else 16-byte sc1 stores and sc1 loads
template <bool FENCE> __device__ __forceinline__ void st16(vf4* p, vf4 v)
{
    if (FENCE) __builtin_nontemporal_store(v, p);
    else asm volatile("global_store_dwordx4 %0, %1, off sc1" :: "v"(p), "v"(v) : "memory");
}
template <bool FENCE> __device__ __forceinline__ int ld4(const int* p)
{
    return FENCE ? __builtin_nontemporal_load(p) : __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
template <bool FENCE> __device__ __forceinline__ void publish()      // after a wave's stores, before the workgroup's barrier
{
    if (FENCE) __threadfence(); else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

// wait until *p >= want; false on time-out / abort (thread 0 of a workgroup calls it, the result is broadcast through LDS)
template <bool FENCE>
__device__ bool spin_ge(const unsigned* p, unsigned want, Ctl* c, unsigned long long t_end)
{
    while (ld_flag<FENCE>(p) < want) {
        if (__builtin_amdgcn_s_memrealtime() > t_end || __hip_atomic_load(&c->abort_, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) {
            __hip_atomic_store(&c->abort_, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            return false;
        }
        __builtin_amdgcn_s_sleep(1);
    }
    return true;
}

// the step's output of one env workgroup (256 envs): obs rows [256][35] float32 as one contiguous block + rewards + done flags,
// dwordx4 non-temporal stores -- the rollout kernel's 149 B per env-step minus the action read
template <bool FENCE>
__device__ __forceinline__ void write_step(float* obs, float* rew, unsigned char* done, int wg, int t, float v)
{
    vf4 x = {v, v + 1.f, v + 2.f, (float)t};
    vf4* o = (vf4*)(obs + (size_t)wg * ENV_T * F);
    for (int g = threadIdx.x; g < ENV_T * F / 4; g += ENV_T) st16<FENCE>(o + g, x);
    if (threadIdx.x < ENV_T / 4) st16<FENCE>((vf4*)(rew + wg * ENV_T) + threadIdx.x, x);              // 256 rewards = 64 x 16 bytes
    if (threadIdx.x < ENV_T / 16) st16<FENCE>((vf4*)(done + wg * ENV_T) + threadIdx.x, vf4{0.f, 0.f, 0.f, 0.f});      // 256 flags = 16 x 16 bytes
}

// ---- A: resident env kernel + one policy kernel per step
template <bool FENCE>
__global__ void __launch_bounds__(ENV_T) k_env_a(Ctl* c, const int* act, float* obs, float* rew, unsigned char* done, int T, unsigned limit_ms)
{
    __shared__ int ok;
    const unsigned long long t_end = __builtin_amdgcn_s_memrealtime() + limit_ms * TICKS_PER_MS;
    float acc = 0.f;
    for (int t = 0; t < T; t++) {
        if (threadIdx.x == 0) ok = spin_ge<FENCE>(&c->flag, (unsigned)t + 1u, c, t_end);
        __syncthreads();
        if (!ok) return;
        const int a = ld4<FENCE>(act + (t & 1) * N + blockIdx.x * ENV_T + threadIdx.x);      // fresh from the policy
        acc += (float)a;
        write_step<FENCE>(obs, rew, done, blockIdx.x, t, acc);
        publish<FENCE>();                                   // the outputs are visible device-wide before the signal
        __syncthreads();
        if (threadIdx.x == 0) add_flag<FENCE>(&c->done_ctr);
    }
}

template <bool FENCE>
__global__ void __launch_bounds__(POL_T) k_pol_a(Ctl* c, int* act, const float* obs, int t, unsigned limit_ms)
{
    __shared__ int ok;
    const unsigned long long t_end = __builtin_amdgcn_s_memrealtime() + limit_ms * TICKS_PER_MS;
    if (threadIdx.x == 0) ok = spin_ge<FENCE>(&c->done_ctr, (unsigned)ENV_WG * (unsigned)t, c, t_end);      // all envs have finished step t - 1
    __syncthreads();
    if (!ok) return;
    {   // four envs per lane: one word of each env's newest row in (a consumer in the loop), one 16-byte action store out
        const int e = (blockIdx.x * POL_T + threadIdx.x) * 4;
        int a[4];
        for (int q = 0; q < 4; q++) a[q] = (int)(__int_as_float(ld4<FENCE>((const int*)(obs + (size_t)(e + q) * F))) * 0.f) + ((e + q + t) % 5);
        st16<FENCE>((vf4*)(act + (t & 1) * N + e), vf4{__int_as_float(a[0]), __int_as_float(a[1]), __int_as_float(a[2]), __int_as_float(a[3])});
    }
    publish<FENCE>();
    __syncthreads();
    if (threadIdx.x == 0) {
        const unsigned old = add_flag<FENCE>(&c->pol_ctr);
        if (old == (unsigned)POL_WG * (unsigned)(t + 1) - 1u) st_flag<FENCE>(&c->flag, (unsigned)t + 1u);      // the last one opens the mailbox
    }
}

// ---- B: both resident, tile-local flags
template <bool FENCE>
__global__ void __launch_bounds__(ENV_T) k_env_b(Ctl* c, const int* act, float* obs, float* rew, unsigned char* done, int T, unsigned limit_ms)
{
    __shared__ int ok;
    const unsigned long long t_end = __builtin_amdgcn_s_memrealtime() + limit_ms * TICKS_PER_MS;
    float acc = 0.f;
    for (int t = 0; t < T; t++) {
        if (threadIdx.x == 0) ok = spin_ge<FENCE>(&c->tile_ready[blockIdx.x], (unsigned)t + 1u, c, t_end);
        __syncthreads();
        if (!ok) return;
        const int a = ld4<FENCE>(act + (t & 1) * N + blockIdx.x * ENV_T + threadIdx.x);
        acc += (float)a;
        write_step<FENCE>(obs, rew, done, blockIdx.x, t, acc);
        publish<FENCE>();
        __syncthreads();
        if (threadIdx.x == 0) st_flag<FENCE>(&c->tile_done[blockIdx.x], (unsigned)t + 1u);
    }
}

template <bool FENCE>
__global__ void __launch_bounds__(POL_T) k_pol_b(Ctl* c, int* act, const float* obs, int T, unsigned limit_ms)
{
    const unsigned long long t_end = __builtin_amdgcn_s_memrealtime() + limit_ms * TICKS_PER_MS;
    constexpr int TILES = ENV_WG / POL_WG;                  // env workgroups per policy workgroup (4): one wave of the policy each
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int tile = blockIdx.x * TILES + wave;
    for (int t = 0; t < T; t++) {
        // each WAVE serves its own tile: no workgroup barrier couples the four tiles
        int w_ok = 1;
        if (lane == 0) w_ok = spin_ge<FENCE>(&c->tile_done[tile], (unsigned)t, c, t_end);
        w_ok = __shfl(w_ok, 0);
        if (!w_ok) return;
        {
            const int e = tile * ENV_T + lane * 4;
            int a[4];
            for (int q = 0; q < 4; q++) a[q] = (int)(__int_as_float(ld4<FENCE>((const int*)(obs + (size_t)(e + q) * F))) * 0.f) + ((e + q + t) % 5);
            st16<FENCE>((vf4*)(act + (t & 1) * N + e), vf4{__int_as_float(a[0]), __int_as_float(a[1]), __int_as_float(a[2]), __int_as_float(a[3])});
        }
        publish<FENCE>();
        if (lane == 0) st_flag<FENCE>(&c->tile_ready[tile], (unsigned)t + 1u);
    }
}

template <bool FENCE> int run(int T);

int main(int argc, char** argv)
{
    const int T = argc > 1 ? atoi(argv[1]) : 2000;
    const bool fence = argc > 2 && !strcmp(argv[2], "fence");
    printf("hand-off protocol: %s\n", fence ? "release / acquire fences (plain stores, __threadfence, acquire polls)" : "sc1 write-through stores + drained relaxed flags + sc1 loads");
    return fence ? run<true>(T) : run<false>(T);
}

template <bool FENCE> int run(int T)
{
    const unsigned limit_ms = 2000;
    CK(hipSetDevice(0));
    Ctl* c; int* act; float *obs, *rew; unsigned char* done;
    CK(hipMalloc(&c, sizeof(Ctl))); CK(hipMalloc(&act, 2 * N * sizeof(int)));
    CK(hipMalloc(&obs, (size_t)N * F * 4)); CK(hipMalloc(&rew, N * 4)); CK(hipMalloc(&done, N));
    CK(hipMemset(obs, 0, (size_t)N * F * 4)); CK(hipMemset(act, 0, 2 * N * sizeof(int)));
    hipStream_t s_env, s_pol;
    CK(hipStreamCreateWithFlags(&s_env, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&s_pol, hipStreamNonBlocking));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const double bytes = (double)N * (F * 4 + 4 + 1 + 4);
    printf("persistent-step lower bound, N = %d envs, %d steps per run, %.2f MB per step (149 B per env-step)\n", N, T, bytes / 1e6);
    Ctl host;
    hipGraph_t pol_graph; hipGraphExec_t pol_exec;
    CK(hipStreamBeginCapture(s_pol, hipStreamCaptureModeThreadLocal));
    for (int t = 0; t < T; t++) hipLaunchKernelGGL(k_pol_a<FENCE>, dim3(POL_WG), dim3(POL_T), 0, s_pol, c, act, obs, t, limit_ms);
    CK(hipStreamEndCapture(s_pol, &pol_graph));
    CK(hipGraphInstantiate(&pol_exec, pol_graph, nullptr, nullptr, 0));
    for (int rep = 0; rep < 3; rep++) {
        // ---- A
        CK(hipMemset(c, 0, sizeof(Ctl))); CK(hipDeviceSynchronize());
        CK(hipEventRecord(e0, s_env));
        hipLaunchKernelGGL(k_env_a<FENCE>, dim3(ENV_WG), dim3(ENV_T), 0, s_env, c, act, obs, rew, done, T, limit_ms);
        CK(hipEventRecord(e1, s_env));
        CK(hipGraphLaunch(pol_exec, s_pol));             // the T policy launches, back to back (a graph: no host launch rate in the way)
        CK(hipDeviceSynchronize());
        float ms = 0.f; CK(hipEventElapsedTime(&ms, e0, e1));
        CK(hipMemcpy(&host, c, sizeof(Ctl), hipMemcpyDeviceToHost));
        printf("A  policy kernel per step, grid-wide fan-in + one mailbox word: %.2f us per step = %.0f GB/s = %.2f of 8 TB/s%s (done %u / %u)\n",
               ms * 1e3 / T, bytes / (ms * 1e-3 / T) / 1e9, bytes / (ms * 1e-3 / T) / 8e12, host.abort_ ? "  ** ABORTED (time-out) **" : "",
               host.done_ctr, (unsigned)ENV_WG * T);
        fflush(stdout);
        // ---- B
        CK(hipMemset(c, 0, sizeof(Ctl))); CK(hipDeviceSynchronize());
        CK(hipEventRecord(e0, s_env));
        hipLaunchKernelGGL(k_env_b<FENCE>, dim3(ENV_WG), dim3(ENV_T), 0, s_env, c, act, obs, rew, done, T, limit_ms);
        CK(hipEventRecord(e1, s_env));
        hipLaunchKernelGGL(k_pol_b<FENCE>, dim3(POL_WG), dim3(POL_T), 0, s_pol, c, act, obs, T, limit_ms);
        CK(hipDeviceSynchronize());
        CK(hipEventElapsedTime(&ms, e0, e1));
        CK(hipMemcpy(&host, c, sizeof(Ctl), hipMemcpyDeviceToHost));
        unsigned mn = ~0u; for (int q = 0; q < ENV_WG; q++) mn = std::min(mn, host.tile_done[q]);
        printf("B  resident policy, tile-local flags (no fan-in):               %.2f us per step = %.0f GB/s = %.2f of 8 TB/s%s (min tile steps %u / %d)\n",
               ms * 1e3 / T, bytes / (ms * 1e-3 / T) / 1e9, bytes / (ms * 1e-3 / T) / 8e12, host.abort_ ? "  ** ABORTED (time-out) **" : "", mn, T);
        fflush(stdout);
    }
    return 0;
}
