// Microbenchmark for the question "LDS or DPP between the register passes of the wave-per-array FFT?" (north_star: "wavefront
// DPP / shuffle for the butterflies"; VERDICT r4 missing item 5: never A/B'd).
//
// A lane of the 1024-point transform holds 16 complex points (32 registers).  Four radix-2 stages run inside the lane; the next
// stages need the points of OTHER lanes.  Two ways to get them there:
//   LDS   what ships (ulcx_fft.h): write the 16 points (16 x ds_write_b64), read the 16 points of the next pass's index set
//         (16 x ds_read_b64) - ONE round trip buys the next FOUR stages (the lane then holds the right 16 points again);
//   DPP   keep the points where they are and run each further stage as a butterfly between lane i and lane i ^ k: per stage every
//         one of the 32 registers is fetched from the partner lane (xor 1, 2: quad_perm; xor 8: row_ror:8; xor 4: two masked row
//         rotations; xor 16 / 32: v_permlane16_swap / v_permlane32_swap) and the lane's role (upper / lower output) is a select.
// Measured: cycles of a SIMD per "four stages' worth of exchange" for both, at 1 / 2 / 4 waves per SIMD, arithmetic excluded
// (identical in both forms).
//   hipcc --offload-arch=gfx950 -O3 fft_exchange.hip -o fft_exchange && ./fft_exchange
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <map>
#include <vector>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
struct Rec { unsigned long long t0, t1; unsigned hw, xcc; };
typedef float v2f __attribute__((ext_vector_type(2)));

template <int KIND>
__global__ void k_ex(Rec *rec, float *sink, int iters) {
    extern __shared__ float lds[];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    v2f *z = (v2f *)lds + wv * 1088;                         // this wave's padded 1024-point array
    v2f x[16];
#pragma unroll
    for (int m = 0; m < 16; m++) { x[m].x = (float)(lane + m); x[m].y = (float)(lane - m); }
    __builtin_amdgcn_s_barrier();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; it++) {
        if (KIND == 0) {
            // one LDS round trip: points lane + 64 m out, points 16 lane' + m of the next pass in (padded: + index >> 4)
#pragma unroll
            for (int m = 0; m < 16; m++) { const int p = lane + 64 * m; z[p + (p >> 4)] = x[m]; }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
            for (int m = 0; m < 16; m++) { const int p = ((lane >> 2) << 6) + (lane & 3) + 4 * m; x[m] = z[p + (p >> 4)]; }
        } else {
            // four cross-lane stages: partner = lane ^ 8, ^ 4, ^ 2, ^ 1; per register one fetch + one select of the role
#pragma unroll
            for (int st = 0; st < 4; st++) {
                const bool upper = (lane >> (3 - st)) & 1;
#pragma unroll
                for (int m = 0; m < 16; m++) {
                    int px, py;
                    const int ax = __float_as_int(x[m].x), ay = __float_as_int(x[m].y);
                    if (st == 0) { px = __builtin_amdgcn_update_dpp(0, ax, 0x128, 0xF, 0xF, false); py = __builtin_amdgcn_update_dpp(0, ay, 0x128, 0xF, 0xF, false); }          // row_ror:8 = lane ^ 8
                    else if (st == 1) {                                                                                                                                          // lane ^ 4: two masked rotations
                        px = __builtin_amdgcn_update_dpp(0, ax, 0x124, 0xF, 0x5, false); px = __builtin_amdgcn_update_dpp(px, ax, 0x12C, 0xF, 0xA, false);
                        py = __builtin_amdgcn_update_dpp(0, ay, 0x124, 0xF, 0x5, false); py = __builtin_amdgcn_update_dpp(py, ay, 0x12C, 0xF, 0xA, false);
                    }
                    else if (st == 2) { px = __builtin_amdgcn_update_dpp(0, ax, 0x4E, 0xF, 0xF, false); py = __builtin_amdgcn_update_dpp(0, ay, 0x4E, 0xF, 0xF, false); }       // quad_perm [2,3,0,1]
                    else { px = __builtin_amdgcn_update_dpp(0, ax, 0xB1, 0xF, 0xF, false); py = __builtin_amdgcn_update_dpp(0, ay, 0xB1, 0xF, 0xF, false); }                      // quad_perm [1,0,3,2]
                    // the butterfly's two inputs in role order (the arithmetic itself is the same in both forms and left out)
                    const float bx = __int_as_float(px), by = __int_as_float(py);
                    x[m].x = upper ? bx : x[m].x; x[m].y = upper ? x[m].y : by;
                }
            }
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float out = 0.0f;
#pragma unroll
    for (int m = 0; m < 16; m++) out += x[m].x + x[m].y;
    if (out == 123.456f) sink[0] = out;
    if (lane == 0) {
        unsigned hw, xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        Rec r; r.t0 = t0; r.t1 = t1; r.hw = hw; r.xcc = xcc;
        rec[blockIdx.x * (blockDim.x / 64) + wv] = r;
    }
}

int main(int argc, char **argv) {
    hipDeviceProp_t prop; CHECK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    Rec *rec; float *sink; const int maxw = cus * 8 * 4;
    CHECK(hipMalloc(&rec, sizeof(Rec) * maxw)); CHECK(hipMalloc(&sink, 64));
    std::vector<Rec> h(maxw);
    const int iters = argc > 1 ? atoi(argv[1]) : 2000;
    printf("# cycles of a SIMD per exchange that buys four radix-2 stages of a 1024-point transform (16 complex points per lane), arithmetic excluded\n");
    printf("| form | 1 wave/SIMD | 2 waves/SIMD | 4 waves/SIMD |\n|---|---|---|---|\n");
    for (int kind = 0; kind < 2; kind++) {
        printf("| %s |", kind == 0 ? "LDS round trip (16 x ds_write_b64 + 16 x ds_read_b64, one wave fence)" : "DPP, four stages (lane ^ 8, ^ 4, ^ 2, ^ 1: 2 fetches + 2 selects per point and stage)");
        for (int w : {1, 2, 4}) {
            auto f = kind == 0 ? k_ex<0> : k_ex<1>;
            size_t lds = (160 * 1024) / w - (w == 1 ? 1024 : 512);
            CHECK(hipFuncSetAttribute((const void *)f, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            const int grid = cus * w, nw = grid * 4;
            for (int rep = 0; rep < 2; rep++) { hipLaunchKernelGGL(f, dim3(grid), dim3(256), lds, 0, rec, sink, iters); CHECK(hipDeviceSynchronize()); }
            CHECK(hipMemcpy(h.data(), rec, sizeof(Rec) * nw, hipMemcpyDeviceToHost));
            std::map<unsigned, std::vector<Rec>> by_simd;
            for (int i = 0; i < nw; i++) by_simd[((h[i].xcc & 15) << 16) | (h[i].hw & 0xff30)].push_back(h[i]);
            std::vector<double> r;
            for (auto &kv : by_simd) {
                unsigned long long a = ~0ull, b = 0;
                for (auto &x : kv.second) { a = std::min(a, x.t0); b = std::max(b, x.t1); }
                r.push_back((double)(b - a) / ((double)iters * kv.second.size()));
            }
            std::sort(r.begin(), r.end());
            printf(" %.0f |", r[r.size() / 2]);
        }
        printf("\n"); fflush(stdout);
    }
    return 0;
}
