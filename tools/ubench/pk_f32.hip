// microbenchmark: packed fp32 (v_pk_add_f32 / v_pk_mul_f32) against scalar v_add_f32 / v_mul_f32 on gfx950.
// hipcc -O3 --offload-arch=gfx950 -ffp-contract=off pk_f32.hip -o pk_f32 && ./pk_f32
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float v2f __attribute__((ext_vector_type(2)));
#define ITER 4096
template <int MODE>
__global__ __launch_bounds__(256) void k(float *out, float a0, float b0) {
    // 8 independent chains per lane
    v2f x[8]; v2f w = { b0, b0 * 0.5f };
    for (int i = 0; i < 8; i++) x[i] = (v2f){ a0 + i + threadIdx.x, a0 - i };
    for (int it = 0; it < ITER; it++) {
#pragma unroll
        for (int i = 0; i < 8; i++) {
            if (MODE == 0) { x[i].x = x[i].x * w.x; x[i].y = x[i].y * w.y; x[i].x = x[i].x + w.y; x[i].y = x[i].y + w.x; }   // scalar: 4 ops
            else { x[i] = x[i] * w; x[i] = x[i] + w.yx; }                                                                     // packed: 2 ops
            if (MODE == 0) { asm volatile("" : "+v"(x[i].x), "+v"(x[i].y)); } else { asm volatile("" : "+v"(x[i])); }
        }
    }
    float s = 0; for (int i = 0; i < 8; i++) s += x[i].x + x[i].y;
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
int main() {
    float *d; hipMalloc(&d, 4 * 256 * 4096);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int wgs : {256, 1024, 2048}) for (int mode = 0; mode < 2; mode++) {
        for (int rep = 0; rep < 2; rep++) {
            hipEventRecord(e0);
            if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(wgs), dim3(256), 0, 0, d, 1.0f, 1.0001f);
            else hipLaunchKernelGGL(k<1>, dim3(wgs), dim3(256), 0, 0, d, 1.0f, 1.0001f);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            if (rep) printf("wgs %d mode %s: %.3f ms  -> %.2f cycles(2.4GHz) per wave-instruction-pair-group per SIMD\n", wgs, mode ? "packed" : "scalar", ms,
                            ms * 1e-3 * 2.4e9 / ((double)ITER * 8 * (mode ? 2 : 4) * (wgs * 4.0 / 1024.0)));
        }
    }
    return 0;
}
