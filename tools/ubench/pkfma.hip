// VALU issue-rate microbenchmark: v_fma_f32 vs v_pk_fma_f32 on gfx950, one or several waves per SIMD.
//   hipcc --offload-arch=gfx950 -O3 -o pkfma tools/ubench/pkfma.hip && ./pkfma
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f2 __attribute__((ext_vector_type(2)));

template <bool PK>
__global__ void k(float* out, int iters, float a, float b) {
    f2 acc[8];
    for (int i = 0; i < 8; ++i) acc[i] = (f2){(float)threadIdx.x + i, (float)i};
    const f2 av = {a, a * 1.0001f}, bv = {b, b * 0.999f};
    long long t0 = clock64();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            if (PK) {
                acc[i] = __builtin_elementwise_fma(acc[i], av, bv);
            } else {
                acc[i].x = __builtin_fmaf(acc[i].x, av.x, bv.x);
                acc[i].y = __builtin_fmaf(acc[i].y, av.y, bv.y);
            }
        }
    }
    long long t1 = clock64();
    float s = 0;
    for (int i = 0; i < 8; ++i) s += acc[i].x + acc[i].y;
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) ((long long*)out)[1 << 20] = t1 - t0;
}

int main() {
    float* d;
    hipMalloc(&d, (1 << 23) + 64);
    const int iters = 4096;
    for (int waves = 1; waves <= 4; waves *= 2) {
        for (int pk = 0; pk < 2; ++pk) {
            dim3 grid(256), block(256 * waves);      // `waves` waves per SIMD on every CU
            hipEvent_t e0, e1;
            hipEventCreate(&e0); hipEventCreate(&e1);
            if (pk) hipLaunchKernelGGL(k<true>, grid, block, 0, 0, d, 8, 1.0001f, 0.5f); else hipLaunchKernelGGL(k<false>, grid, block, 0, 0, d, 8, 1.0001f, 0.5f);
            hipEventRecord(e0);
            if (pk) hipLaunchKernelGGL(k<true>, grid, block, 0, 0, d, iters, 1.0001f, 0.5f); else hipLaunchKernelGGL(k<false>, grid, block, 0, 0, d, iters, 1.0001f, 0.5f);
            hipEventRecord(e1);
            hipDeviceSynchronize();
            float ms; hipEventElapsedTime(&ms, e0, e1);
            long long cyc; hipMemcpy(&cyc, ((long long*)d) + (1 << 20), 8, hipMemcpyDeviceToHost);
            const double fma_per_lane = (double)iters * 16;             // 8 pairs x 2
            const double tflops = 2.0 * fma_per_lane * 256.0 * 256 * waves / (ms * 1e-3) / 1e12;
            printf("waves/SIMD %d  %s: %.3f ms  %.1f TFLOP/s  wave-0 cycles per FMA-pair-step %.2f\n", waves, pk ? "v_pk_fma_f32" : "v_fma_f32  ",
                   ms, tflops, (double)cyc / (iters * 8));
        }
    }
    return 0;
}
