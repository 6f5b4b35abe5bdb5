// Achievable HBM stream rates on this GPU (reference points for the HBM-bound kernels' rooflines): fill, nontemporal fill, copy, read-sum.
//   hipcc --offload-arch=gfx950 -O3 -o hbm_probe tools/probes/hbm_probe.hip && ./hbm_probe [MiB]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

template <bool NT> __global__ __launch_bounds__(256) void fill_kernel(f32x4* __restrict__ p, size_t n, float v) {
    const f32x4 x = {v, v, v, v};
    for (size_t i = blockIdx.x * 256ull + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        if (NT) __builtin_nontemporal_store(x, p + i); else p[i] = x;
    }
}
template <bool NT> __global__ __launch_bounds__(256) void copy_kernel(const f32x4* __restrict__ s, f32x4* __restrict__ d, size_t n) {
    for (size_t i = blockIdx.x * 256ull + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        const f32x4 x = NT ? __builtin_nontemporal_load(s + i) : s[i];
        if (NT) __builtin_nontemporal_store(x, d + i); else d[i] = x;
    }
}
// each lane owns 32 consecutive bytes and writes them with two 16-byte stores (the access shape of a thread that produces 8 outputs in a row)
template <bool NT> __global__ __launch_bounds__(256) void fill2_kernel(f32x4* __restrict__ p, size_t n, float v) {
    const f32x4 x = {v, v, v, v};
    for (size_t i = blockIdx.x * 256ull + threadIdx.x; 2 * i + 1 < n; i += (size_t)gridDim.x * 256) {
        if (NT) { __builtin_nontemporal_store(x, p + 2 * i); __builtin_nontemporal_store(x, p + 2 * i + 1); } else { p[2 * i] = x; p[2 * i + 1] = x; }
    }
}
__global__ __launch_bounds__(256) void sum_kernel(const f32x4* __restrict__ s, float* out, size_t n) {
    f32x4 a = {0.f, 0.f, 0.f, 0.f};
    for (size_t i = blockIdx.x * 256ull + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) a += s[i];
    if (a.x + a.y + a.z + a.w == 123.456f) *out = 1.f;
}

int main(int argc, char** argv) {
    const size_t mib = argc > 1 ? strtoull(argv[1], nullptr, 10) : 2048, bytes = mib << 20, n = bytes / 16;
    f32x4 *a, *b; float* o;
    CK(hipMalloc(&a, bytes)); CK(hipMalloc(&b, bytes)); CK(hipMalloc(&o, 4));
    CK(hipMemset(a, 0, bytes)); CK(hipMemset(b, 0, bytes));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int grids[] = {2048, 65536};
    for (int g : grids) {
        for (int k = 0; k < 7; ++k) {
            float best = 1e30f;
            for (int rep = 0; rep < 5; ++rep) {
                CK(hipEventRecord(e0));
                switch (k) {
                    case 0: hipLaunchKernelGGL(fill_kernel<false>, dim3(g), dim3(256), 0, 0, a, n, 1.f); break;
                    case 1: hipLaunchKernelGGL(fill_kernel<true>, dim3(g), dim3(256), 0, 0, a, n, 1.f); break;
                    case 2: hipLaunchKernelGGL(copy_kernel<false>, dim3(g), dim3(256), 0, 0, a, b, n); break;
                    case 3: hipLaunchKernelGGL(copy_kernel<true>, dim3(g), dim3(256), 0, 0, a, b, n); break;
                    case 5: hipLaunchKernelGGL(fill2_kernel<false>, dim3(g), dim3(256), 0, 0, a, n, 1.f); break;
                    case 6: hipLaunchKernelGGL(fill2_kernel<true>, dim3(g), dim3(256), 0, 0, a, n, 1.f); break;
                    case 4: hipLaunchKernelGGL(sum_kernel, dim3(g), dim3(256), 0, 0, a, o, n); break;
                    default: break;
                }
                CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
                float ms; CK(hipEventElapsedTime(&ms, e0, e1));
                if (ms < best) best = ms;
            }
            const char* names[] = {"fill", "fill nontemporal", "copy (read + write)", "copy nontemporal", "read", "fill 2x16B per lane", "fill 2x16B per lane nt"};
            const double moved = (k == 2 || k == 3) ? 2.0 * bytes : (double)bytes;
            printf("%-22s grid %6d  %8.1f us  %7.2f TB/s\n", names[k], g, best * 1e3, moved / (best * 1e-3) / 1e12);
        }
    }
    return 0;
}
