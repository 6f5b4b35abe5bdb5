// launch_gap_probe.hip — what does a dependent kernel boundary cost on this box, and what does it depend on?
//   hipcc --offload-arch=gfx950 -O3 -o launch_gap_probe launch_gap_probe.hip && ./launch_gap_probe
// A chain of K launches of one kernel on one stream, timed with events (eager and as a captured graph), per launch:
//   trivial      : 256 WGs x 256 threads, one store each
//   tiny         : 1 WG x 64 threads (add_int_kernel's shape)
//   dirty <MB>   : every launch rewrites <MB> of fp32 (plain 16-byte stores), the NEXT launch is the measured boundary
//   sc1   <MB>   : the same bytes written through (sc1 stores)
//   alternating  : dirty-writer followed by trivial (the step's "finish launch after a slab writer" pattern)
// MI355X_MICROARCH.md prices the boundary at 1.45 us between trivial kernels (+ B / 6 TB/s behind B dirty bytes); rocprofv3 shows
// 4.7 us for every trivial kernel inside our replayed step.  This probe separates launch floor, dirty-L2 write-back and graph effects.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

__global__ void trivial(float* p) { p[blockIdx.x * 256 + threadIdx.x] = 1.f; }
__global__ void tiny(int* p) { if (threadIdx.x == 0) p[0] += 1; }
__global__ void writer(float4* p, size_t n4) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) p[i] = make_float4(1.f, 2.f, 3.f, 4.f);
}
__global__ void writer_sc1(float4* p, size_t n4) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
        typedef float f4 __attribute__((ext_vector_type(4)));
        f4 v = {1.f, 2.f, 3.f, 4.f};
        asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(p + i), "v"(v) : "memory");
    }
}
__global__ void reader(const float4* p, size_t n4, float* out) {
    float s = 0.f;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) { float4 v = p[i]; s += v.x + v.y + v.z + v.w; }
    if (s == 12345.f) out[0] = s;
}

template <typename F> static float time_chain(hipStream_t st, int K, int reps, bool graph, F enqueue) {
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    hipGraph_t g = nullptr; hipGraphExec_t ge = nullptr;
    if (graph) {
        hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal);
        for (int i = 0; i < K; ++i) enqueue(i);
        hipStreamEndCapture(st, &g);
        hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
    }
    float best = 1e30f;
    for (int r = 0; r < reps + 2; ++r) {
        hipEventRecord(a, st);
        if (graph) hipGraphLaunch(ge, st); else for (int i = 0; i < K; ++i) enqueue(i);
        hipEventRecord(b, st);
        hipEventSynchronize(b);
        float ms; hipEventElapsedTime(&ms, a, b);
        if (r >= 2 && ms < best) best = ms;
    }
    if (ge) hipGraphExecDestroy(ge);
    if (g) hipGraphDestroy(g);
    return best * 1e3f / K;     // us per launch
}

int main() {
    hipStream_t st; CK(hipStreamCreate(&st));
    const size_t MAXB = 64ull << 20;
    float* buf; CK(hipMalloc(&buf, MAXB)); int* cnt; CK(hipMalloc(&cnt, 256)); CK(hipMemset(cnt, 0, 256));
    float* tb; CK(hipMalloc(&tb, 256 * 256 * 4));
    const int K = 64, reps = 20;
    for (int graph = 0; graph < 2; ++graph) {
        const char* mode = graph ? "graph" : "eager";
        printf("%s trivial 256x256       : %.2f us/launch\n", mode, time_chain(st, K, reps, graph, [&](int) { hipLaunchKernelGGL(trivial, dim3(256), dim3(256), 0, st, tb); }));
        printf("%s tiny 1x64             : %.2f us/launch\n", mode, time_chain(st, K, reps, graph, [&](int) { hipLaunchKernelGGL(tiny, dim3(1), dim3(64), 0, st, cnt); }));
        for (size_t mb : {1, 4, 16, 64}) {
            const size_t n4 = (mb << 20) / 16;
            const float w = time_chain(st, K, reps, graph, [&](int) { hipLaunchKernelGGL(writer, dim3(2048), dim3(256), 0, st, (float4*)buf, n4); });
            const float ws = time_chain(st, K, reps, graph, [&](int) { hipLaunchKernelGGL(writer_sc1, dim3(2048), dim3(256), 0, st, (float4*)buf, n4); });
            const float wt = time_chain(st, K, reps, graph, [&](int i) {
                if (i & 1) hipLaunchKernelGGL(tiny, dim3(1), dim3(64), 0, st, cnt); else hipLaunchKernelGGL(writer, dim3(2048), dim3(256), 0, st, (float4*)buf, n4); });
            const float wst = time_chain(st, K, reps, graph, [&](int i) {
                if (i & 1) hipLaunchKernelGGL(tiny, dim3(1), dim3(64), 0, st, cnt); else hipLaunchKernelGGL(writer_sc1, dim3(2048), dim3(256), 0, st, (float4*)buf, n4); });
            const float wr = time_chain(st, K, reps, graph, [&](int i) {
                if (i & 1) hipLaunchKernelGGL(reader, dim3(2048), dim3(256), 0, st, (const float4*)buf, n4, tb); else hipLaunchKernelGGL(writer, dim3(2048), dim3(256), 0, st, (float4*)buf, n4); });
            printf("%s writer %3zu MB         : plain %.2f us/launch (%.2f TB/s)  sc1 %.2f;  writer+tiny pair: plain %.2f sc1 %.2f us/pair;  writer+reader pair %.2f\n", mode, mb, w, mb * 1.048576 / w, ws, 2 * wt, 2 * wst, 2 * wr);
        }
    }
    // a kernel with a large by-value argument block (the WgradTable / AdamTable launches pass ~4 KB of kernargs)
    return 0;
}
