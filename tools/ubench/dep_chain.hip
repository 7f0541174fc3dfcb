// Microbenchmark: cycles per DEPENDENT vector instruction of one wave (what the serial recurrences of the window-control
// kernels pay per step), alone on its SIMD and with 1..7 idle-spinning neighbours.   hipcc --offload-arch=gfx950 -O3 dep_chain.hip -o dep_chain && ./dep_chain
#include <hip/hip_runtime.h>
#include <cstdio>
template <int KIND>
__global__ void k(float *out, unsigned long long *cyc, int iters, float c) {
    float env = out[threadIdx.x], v = out[64 + threadIdx.x], sum = 0.0f;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int u = 0; u < 16; u++) {
            if (KIND == 0) { asm volatile("v_add_f32 %0, %0, %1" : "+v"(env) : "v"(v)); }                       // 1 dependent op
            else if (KIND == 1) { float d; asm volatile("v_sub_f32 %0, %1, %2" : "=v"(d) : "v"(v), "v"(env));   // the recurrence: sub, mul, add
                                  asm volatile("v_mul_f32 %0, %0, %1" : "+v"(d) : "v"(c));
                                  asm volatile("v_add_f32 %0, %0, %1" : "+v"(env) : "v"(d)); }
            else { float d; asm volatile("v_sub_f32 %0, %1, %2" : "=v"(d) : "v"(v), "v"(env));
                   asm volatile("v_mul_f32 %0, %0, %1" : "+v"(d) : "v"(c));
                   asm volatile("v_add_f32 %0, %0, %1" : "+v"(env) : "v"(d));
                   asm volatile("v_add_f32 %0, %0, %1" : "+v"(sum) : "v"(env)); }                                 // + the independent sum
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    out[threadIdx.x] = env + sum;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
int main() {
    float *d; unsigned long long *c; hipMalloc(&d, 1024); hipMalloc(&c, 8 * 4096); hipMemset(d, 0, 1024);
    const int iters = 20000;
    for (int kind = 0; kind < 3; kind++) {
        for (int grid : {1, 256, 1024, 4096}) {
            for (int rep = 0; rep < 2; rep++) {
                if (kind == 0) hipLaunchKernelGGL(k<0>, dim3(grid), dim3(64), 0, 0, d, c, iters, 0.999f);
                else if (kind == 1) hipLaunchKernelGGL(k<1>, dim3(grid), dim3(64), 0, 0, d, c, iters, 0.999f);
                else hipLaunchKernelGGL(k<2>, dim3(grid), dim3(64), 0, 0, d, c, iters, 0.999f);
                hipDeviceSynchronize();
            }
            unsigned long long h; hipMemcpy(&h, c, 8, hipMemcpyDeviceToHost);
            const int ops = kind == 0 ? 1 : 3;
            printf("kind %d (%s) grid %4d waves: %.2f cycles (s_memtime ticks) per step, %.2f per dependent instruction\n", kind,
                   kind == 0 ? "add chain" : kind == 1 ? "sub-mul-add" : "sub-mul-add + sum", grid, (double)h / (iters * 16.0), (double)h / (iters * 16.0 * ops));
        }
    }
    return 0;
}
