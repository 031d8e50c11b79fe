// VALU issue-rate probe for gfx950: v_fma_f32 vs v_pk_fma_f32, with 1 / 2 / 4 waves per SIMD.
// build: hipcc -O3 --offload-arch=gfx950 tools/micro/valu_rate.hip -o gpurun_out/valu_rate ; run on the MI355X box
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f2 __attribute__((ext_vector_type(2)));

template <int MODE>
__global__ __launch_bounds__(256) void probe(float* out, int iters, float a, float b) {
    f2 acc[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = f2{(float)threadIdx.x + i, 1.f};
    f2 x = {a, a * 0.5f}, y = {b, b};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                if (MODE == 0) {        // 2 scalar FMAs
                    asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(acc[i].x) : "v"(x.x), "v"(y.x));
                    asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(acc[i].y) : "v"(x.y), "v"(y.y));
                } else if (MODE == 1) {                // 1 packed FMA = same work
                    asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(x), "v"(y));
                } else {                // packed FMA with op_sel broadcast of the low half of src0
                    asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[0,1,1]" : "+v"(acc[i]) : "v"(x), "v"(y));
                }
            }
        }
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) s += acc[i].x + acc[i].y;
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int MODE>
static void run(const char* name, int wgs_per_cu, float* out) {
    const int iters = 4000, cus = 256;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    probe<MODE><<<cus * wgs_per_cu, 256>>>(out, 10, 1.0001f, 0.5f);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    probe<MODE><<<cus * wgs_per_cu, 256>>>(out, iters, 1.0001f, 0.5f);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double fma_lanes = (double)cus * wgs_per_cu * 256 * iters * 4 * 16 * 2;
    const double wave_fma_pairs = (double)iters * 4 * 16;                  // per wave: pairs of FMAs (or one packed)
    const double cyc = ms * 1e-3 * 2.4e9;
    printf("%-28s waves/SIMD %d: %7.3f ms  %6.1f TFLOP/s   %.2f cycles per FMA pair per wave (at 2.4 GHz), per SIMD %.2f\n", name, wgs_per_cu, ms,
           2 * fma_lanes / ms * 1e-9, cyc / wave_fma_pairs, cyc / wave_fma_pairs / wgs_per_cu);
}

int main() {
    float* out; hipMalloc(&out, 256 * 8 * 256 * 4);
    for (int w : {1, 2, 4, 8}) {
        run<0>("2 x v_fma_f32", w, out);
        run<1>("v_pk_fma_f32", w, out);
        run<2>("v_pk_fma_f32 op_sel_hi", w, out);
    }
    return 0;
}
