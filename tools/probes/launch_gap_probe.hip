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

// 32 DISTINCT kernels of ~1.5 KB of code each (a chain in the real step never launches the same kernel twice in a row: every launch starts with cold
// instruction-cache lines and a cold kernel descriptor); I selects a different unrolled body so the code objects do not fold into one.
template <int I> __global__ void distinct_tiny(int* p, int n) {
    int v = p[threadIdx.x & 63];
#pragma unroll
    for (int k = 0; k < 64; ++k) v = v * (I + 3 + k) + ((v >> (k & 7)) ^ (I * 7919 + k));
    if (n == 123456789) p[threadIdx.x & 63] = v;      // never true: the body stays, nothing is written
    if (threadIdx.x == 0) p[64 + I] += 1;
}
template <int I> static void launch_distinct(int which, hipStream_t st, int* p) {
    if constexpr (I < 32) {
        if (which == I) { hipLaunchKernelGGL(distinct_tiny<I>, dim3(1), dim3(64), 0, st, p, 0); return; }
        launch_distinct<I + 1>(which, st, p);
    }
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
    float* buf; CK(hipMalloc(&buf, MAXB)); int* cnt; CK(hipMalloc(&cnt, 1024)); CK(hipMemset(cnt, 0, 1024));
    float* tb; CK(hipMalloc(&tb, 256 * 256 * 4));
    const int K = 64, reps = 20;
    for (int graph = 0; graph < 2; ++graph) {
        const char* mode = graph ? "graph" : "eager";
        printf("%s trivial 256x256       : %.2f us/launch\n", mode, time_chain(st, K, reps, graph, [&](int) { hipLaunchKernelGGL(trivial, dim3(256), dim3(256), 0, st, tb); }));
        printf("%s tiny 1x64             : %.2f us/launch\n", mode, time_chain(st, K, reps, graph, [&](int) { hipLaunchKernelGGL(tiny, dim3(1), dim3(64), 0, st, cnt); }));
        printf("%s 32 distinct tiny 1x64 : %.2f us/launch\n", mode, time_chain(st, K, reps, graph, [&](int i) { launch_distinct<0>(i & 31, st, cnt); }));
        // the same with 64 MB rewritten between two passes over the chain, so the kernels' code has left the caches (one step of the model moves > 2 GB)
        printf("%s 32 distinct tiny, cold: %.2f us/launch (chain of 32 + one 64 MB writer; the writer alone %.2f us)\n", mode,
               33.f / 32.f * time_chain(st, 33, reps, graph, [&](int i) { if (i == 32) hipLaunchKernelGGL(writer, dim3(2048), dim3(256), 0, st, (float4*)buf, MAXB / 16); else launch_distinct<0>(i, st, cnt); }),
               time_chain(st, 4, reps, graph, [&](int) { hipLaunchKernelGGL(writer, dim3(2048), dim3(256), 0, st, (float4*)buf, MAXB / 16); }));
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
    // What a tiny kernel costs when everything it needs is cold — the situation inside the train step, where > 2 GB move between two launches of the same
    // small kernel: a 1 GB rewrite (far beyond the 256 MB Infinity Cache and every TLB) in front of EACH tiny launch; the boundary is the pair minus the writer.
    {
        float* huge; CK(hipMalloc(&huge, 1ull << 30));
        const size_t n4 = (1ull << 30) / 16;
        for (int graph = 0; graph < 2; ++graph) {
            const float w = time_chain(st, 8, 6, graph, [&](int) { hipLaunchKernelGGL(writer, dim3(4096), dim3(256), 0, st, (float4*)huge, n4); });
            const float p_same = time_chain(st, 16, 6, graph, [&](int i) {
                if (i & 1) hipLaunchKernelGGL(tiny, dim3(1), dim3(64), 0, st, cnt); else hipLaunchKernelGGL(writer, dim3(4096), dim3(256), 0, st, (float4*)huge, n4); });
            const float p_dist = time_chain(st, 64, 4, graph, [&](int i) {
                if (i & 1) launch_distinct<0>((i >> 1) & 31, st, cnt); else hipLaunchKernelGGL(writer, dim3(4096), dim3(256), 0, st, (float4*)huge, n4); });
            printf("%s 1 GB writer %.1f us; writer + tiny pair %.1f (tiny behind it: %.2f us); writer + one of 32 distinct tiny kernels %.1f (%.2f us)\n", graph ? "graph" : "eager", w,
                   2 * p_same, 2 * p_same - w, 2 * p_dist, 2 * p_dist - w);
        }
        hipFree(huge);
    }
    // a kernel with a large by-value argument block (the WgradTable / AdamTable launches pass ~4 KB of kernargs)
    return 0;
}
