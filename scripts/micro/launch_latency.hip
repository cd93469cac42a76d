// How long does one kernel of a dependent chain cost on this GPU?  (tiny kernels: the floor under every launch)
// build: hipcc --offload-arch=gfx950 -O3 scripts/micro/launch_latency.hip -o /tmp/launch_latency
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <chrono>
#include <cstdio>

__global__ void tiny(float* p) { if (threadIdx.x == 0 && blockIdx.x == 0) p[0] += 1.f; }
__global__ void mid(float* p, int n) {      // ~64 KB read-modify-write per block: a "small BatchNorm" sized kernel
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = p[i] * 1.0001f + 1.f;
}

template <typename F>
static double run(const char* name, int n, hipStream_t s, F f) {
    for (int i = 0; i < 50; ++i) f(i);
    (void)hipStreamSynchronize(s);
    auto t0 = std::chrono::high_resolution_clock::now();
    for (int i = 0; i < n; ++i) f(i);
    auto t1 = std::chrono::high_resolution_clock::now();
    (void)hipStreamSynchronize(s);
    auto t2 = std::chrono::high_resolution_clock::now();
    const double enq = std::chrono::duration<double, std::micro>(t1 - t0).count() / n;
    const double tot = std::chrono::duration<double, std::micro>(t2 - t0).count() / n;
    printf("%-46s enqueue %.2f us/launch   total %.2f us/launch\n", name, enq, tot);
    return tot;
}

int main() {
    float* p;
    (void)hipMalloc(&p, 64 << 20);
    (void)hipMemset(p, 0, 64 << 20);
    hipStream_t s, s2;
    (void)hipStreamCreate(&s);
    (void)hipStreamCreate(&s2);
    const int N = 2000;
    run("tiny, hipLaunchKernelGGL", N, s, [&](int) { hipLaunchKernelGGL(tiny, dim3(1), dim3(64), 0, s, p); });
    run("tiny, hipExtLaunchKernelGGL any-order", N, s, [&](int) {
        hipExtLaunchKernelGGL(tiny, dim3(1), dim3(64), 0, s, nullptr, nullptr, hipExtAnyOrderLaunch, p);
    });
    run("mid (4 MB), hipLaunchKernelGGL", N, s, [&](int) { hipLaunchKernelGGL(mid, dim3(4096), dim3(256), 0, s, p, 1 << 20); });
    run("mid (4 MB), any-order", N, s, [&](int) {
        hipExtLaunchKernelGGL(mid, dim3(4096), dim3(256), 0, s, nullptr, nullptr, hipExtAnyOrderLaunch, p, 1 << 20);
    });
    run("mid (64 MB), hipLaunchKernelGGL", 500, s, [&](int) { hipLaunchKernelGGL(mid, dim3(65536), dim3(256), 0, s, p, 16 << 20); });
    run("tiny, two streams alternating", N, s, [&](int i) { hipLaunchKernelGGL(tiny, dim3(1), dim3(64), 0, (i & 1) ? s : s2, p + (i & 1)); });
    (void)hipStreamSynchronize(s2);
    // hipGraph of 1000 dependent tiny kernels
    hipGraph_t g;
    hipGraphExec_t ge;
    (void)hipStreamBeginCapture(s, hipStreamCaptureModeGlobal);
    for (int i = 0; i < 1000; ++i) hipLaunchKernelGGL(tiny, dim3(1), dim3(64), 0, s, p);
    (void)hipStreamEndCapture(s, &g);
    (void)hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
    (void)hipGraphLaunch(ge, s);
    (void)hipStreamSynchronize(s);
    auto t0 = std::chrono::high_resolution_clock::now();
    for (int r = 0; r < 5; ++r) (void)hipGraphLaunch(ge, s);
    (void)hipStreamSynchronize(s);
    auto t1 = std::chrono::high_resolution_clock::now();
    printf("%-46s total %.2f us/kernel\n", "hipGraph of 1000 tiny kernels", std::chrono::duration<double, std::micro>(t1 - t0).count() / 5000);
    return 0;
}
