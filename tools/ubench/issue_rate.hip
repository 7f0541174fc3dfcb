// Microbenchmark: what a SIMD / a CU of gfx950 ISSUES per cycle, as a function of the waves resident on it.
// The roof `tools/bounds_table.py` prices `valu_ms` / `salu_ms` from.  (MI355X_MICROARCH.md: a wave64 vector instruction
// retires in 2 cycles of a SIMD-32, ONE wave's stream sustains one per 4; `dep_chain.hip` is the one-wave dependent case.)
//
//   hipcc --offload-arch=gfx950 -O3 issue_rate.hip -o issue_rate && ./issue_rate
//
// Every CU gets exactly w workgroups of 256 threads (one wave per SIMD each; dynamic LDS sized so that no more fit), or
// of 64 threads for the one-SIMD rows; the placement is read back (HW_ID) and printed.  A wave runs ITERS x 64
// instructions of one kind on 8 independent registers between two s_memtime reads; per SIMD the rate is
// (instructions of its waves) / (last end - first start); the table prints the median over the SIMDs (CUs) that held
// waves, in WAVE-instructions per cycle per SIMD (vector kinds) or per CU (scalar kinds), and the same from the
// kernel's wall time at 2.4 GHz as a cross-check of the clock.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <map>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

enum { K_FMA, K_ADDU, K_PKFMA, K_CMP_BCNT, K_CMP_ONLY, K_SALU, K_SALU_BCNT, K_ADD3, K_DPP, K_ADDF64, K_FMAF64, K_MULLO, K_CNDMASK, K_MIX_V2S1, K_LDSR, K_KINDS };
static const char *kind_name[K_KINDS] = {
    "v_fma_f32", "v_add_u32", "v_pk_fma_f32", "v_cmp_gt_u32 -> s_bcnt1 -> s_add (3 per probe key)", "v_cmp_gt_u32 (to SGPR pair)",
    "s_add_u32 (SALU only)", "s_bcnt1_i32_b64 + s_add_u32 (SALU only)", "v_add3_u32", "v_add_u32 dpp row_shr:1", "v_add_f64", "v_fma_f64",
    "v_mul_lo_u32", "v_cndmask_b32 (vcc)", "2 x v_fma_f32 + 1 x s_add_u32", "ds_read_b32 (no conflicts)"};
// instructions of the counted class per unrolled body (64 statements)
struct Rec { unsigned long long t0, t1; unsigned hw, xcc; };

template <int KIND>
__global__ void k_rate(Rec *rec, float *sink, int iters) {
    extern __shared__ float lds[];
    float a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    const float c = 0.999f, d = 1e-3f;
    unsigned u0 = threadIdx.x, u1 = u0 * 3, u2 = u0 * 5, u3 = u0 * 7, u4 = u0 + 9, u5 = u0 + 11, u6 = u0 + 13, u7 = u0 + 15;
    double f0 = threadIdx.x, f1 = f0 + 1, f2 = f0 + 2, f3 = f0 + 3, f4 = f0 + 4, f5 = f0 + 5, f6 = f0 + 6, f7 = f0 + 7;
    const double dc = 0.999;
    float2 p0 = {a0, a1}, p1 = {a2, a3}, p2 = {a4, a5}, p3 = {a6, a7}, p4 = {a1, a0}, p5 = {a3, a2}, p6 = {a5, a4}, p7 = {a7, a6};
    const float2 pc = {0.999f, 0.998f};
    unsigned s0 = 1, s1 = 2, s2 = 3, s3 = 4, s4 = 5, s5 = 6, s6 = 7, s7 = 8;
    unsigned long long m0, m1, m2, m3;
    unsigned t0_, t1_, t2_, t3_;
    if (KIND == K_LDSR) { lds[threadIdx.x] = a0; __syncthreads(); u0 = (threadIdx.x & 63) * 4; }
    __builtin_amdgcn_s_barrier();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int r = 0; r < 8; r++) {
            if (KIND == K_FMA)
                asm volatile("v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n v_fma_f32 %3, %3, %8, %9\n"
                             "v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %5, %5, %8, %9\n v_fma_f32 %6, %6, %8, %9\n v_fma_f32 %7, %7, %8, %9\n"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(c), "v"(d));
            else if (KIND == K_ADDU)
                asm volatile("v_add_u32 %0, %0, %8\n v_add_u32 %1, %1, %8\n v_add_u32 %2, %2, %8\n v_add_u32 %3, %3, %8\n"
                             "v_add_u32 %4, %4, %8\n v_add_u32 %5, %5, %8\n v_add_u32 %6, %6, %8\n v_add_u32 %7, %7, %8\n"
                             : "+v"(u0), "+v"(u1), "+v"(u2), "+v"(u3), "+v"(u4), "+v"(u5), "+v"(u6), "+v"(u7) : "v"(i));
            else if (KIND == K_ADD3)
                asm volatile("v_add3_u32 %0, %0, %8, %1\n v_add3_u32 %1, %1, %8, %2\n v_add3_u32 %2, %2, %8, %3\n v_add3_u32 %3, %3, %8, %4\n"
                             "v_add3_u32 %4, %4, %8, %5\n v_add3_u32 %5, %5, %8, %6\n v_add3_u32 %6, %6, %8, %7\n v_add3_u32 %7, %7, %8, %0\n"
                             : "+v"(u0), "+v"(u1), "+v"(u2), "+v"(u3), "+v"(u4), "+v"(u5), "+v"(u6), "+v"(u7) : "v"(i));
            else if (KIND == K_MULLO)
                asm volatile("v_mul_lo_u32 %0, %0, %8\n v_mul_lo_u32 %1, %1, %8\n v_mul_lo_u32 %2, %2, %8\n v_mul_lo_u32 %3, %3, %8\n"
                             "v_mul_lo_u32 %4, %4, %8\n v_mul_lo_u32 %5, %5, %8\n v_mul_lo_u32 %6, %6, %8\n v_mul_lo_u32 %7, %7, %8\n"
                             : "+v"(u0), "+v"(u1), "+v"(u2), "+v"(u3), "+v"(u4), "+v"(u5), "+v"(u6), "+v"(u7) : "v"(i | 1));
            else if (KIND == K_DPP)
                asm volatile("s_nop 1\n v_add_u32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf\n v_add_u32_dpp %1, %1, %1 row_shr:1 row_mask:0xf bank_mask:0xf\n"
                             "v_add_u32_dpp %2, %2, %2 row_shr:1 row_mask:0xf bank_mask:0xf\n v_add_u32_dpp %3, %3, %3 row_shr:1 row_mask:0xf bank_mask:0xf\n"
                             "v_add_u32_dpp %4, %4, %4 row_shr:1 row_mask:0xf bank_mask:0xf\n v_add_u32_dpp %5, %5, %5 row_shr:1 row_mask:0xf bank_mask:0xf\n"
                             "v_add_u32_dpp %6, %6, %6 row_shr:1 row_mask:0xf bank_mask:0xf\n v_add_u32_dpp %7, %7, %7 row_shr:1 row_mask:0xf bank_mask:0xf\n"
                             : "+v"(u0), "+v"(u1), "+v"(u2), "+v"(u3), "+v"(u4), "+v"(u5), "+v"(u6), "+v"(u7));
            else if (KIND == K_CNDMASK)
                asm volatile("v_cndmask_b32 %0, %0, %8, vcc\n v_cndmask_b32 %1, %1, %8, vcc\n v_cndmask_b32 %2, %2, %8, vcc\n v_cndmask_b32 %3, %3, %8, vcc\n"
                             "v_cndmask_b32 %4, %4, %8, vcc\n v_cndmask_b32 %5, %5, %8, vcc\n v_cndmask_b32 %6, %6, %8, vcc\n v_cndmask_b32 %7, %7, %8, vcc\n"
                             : "+v"(u0), "+v"(u1), "+v"(u2), "+v"(u3), "+v"(u4), "+v"(u5), "+v"(u6), "+v"(u7) : "v"(i) : "vcc");
            else if (KIND == K_PKFMA)
                asm volatile("v_pk_fma_f32 %0, %0, %8, %8\n v_pk_fma_f32 %1, %1, %8, %8\n v_pk_fma_f32 %2, %2, %8, %8\n v_pk_fma_f32 %3, %3, %8, %8\n"
                             "v_pk_fma_f32 %4, %4, %8, %8\n v_pk_fma_f32 %5, %5, %8, %8\n v_pk_fma_f32 %6, %6, %8, %8\n v_pk_fma_f32 %7, %7, %8, %8\n"
                             : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3), "+v"(p4), "+v"(p5), "+v"(p6), "+v"(p7) : "v"(pc));
            else if (KIND == K_ADDF64)
                asm volatile("v_add_f64 %0, %0, %8\n v_add_f64 %1, %1, %8\n v_add_f64 %2, %2, %8\n v_add_f64 %3, %3, %8\n"
                             "v_add_f64 %4, %4, %8\n v_add_f64 %5, %5, %8\n v_add_f64 %6, %6, %8\n v_add_f64 %7, %7, %8\n"
                             : "+v"(f0), "+v"(f1), "+v"(f2), "+v"(f3), "+v"(f4), "+v"(f5), "+v"(f6), "+v"(f7) : "v"(dc));
            else if (KIND == K_FMAF64)
                asm volatile("v_fma_f64 %0, %0, %8, %8\n v_fma_f64 %1, %1, %8, %8\n v_fma_f64 %2, %2, %8, %8\n v_fma_f64 %3, %3, %8, %8\n"
                             "v_fma_f64 %4, %4, %8, %8\n v_fma_f64 %5, %5, %8, %8\n v_fma_f64 %6, %6, %8, %8\n v_fma_f64 %7, %7, %8, %8\n"
                             : "+v"(f0), "+v"(f1), "+v"(f2), "+v"(f3), "+v"(f4), "+v"(f5), "+v"(f6), "+v"(f7) : "v"(dc));
            else if (KIND == K_CMP_ONLY)
                asm volatile("v_cmp_gt_u32 %0, %4, %5\n v_cmp_gt_u32 %1, %5, %6\n v_cmp_gt_u32 %2, %6, %7\n v_cmp_gt_u32 %3, %7, %4\n"
                             "v_cmp_gt_u32 %0, %5, %4\n v_cmp_gt_u32 %1, %6, %5\n v_cmp_gt_u32 %2, %7, %6\n v_cmp_gt_u32 %3, %4, %7\n"
                             : "=&s"(m0), "=&s"(m1), "=&s"(m2), "=&s"(m3) : "v"(u0), "v"(u1), "v"(u2), "v"(u3));
            else if (KIND == K_CMP_BCNT)    // 8 probe keys: 8 compares, 8 counts, 8 adds; a count is used 4+ instructions behind its compare
                asm volatile("v_cmp_gt_u32 %0, %12, %13\n v_cmp_gt_u32 %1, %13, %14\n v_cmp_gt_u32 %2, %14, %15\n v_cmp_gt_u32 %3, %15, %12\n"
                             "s_bcnt1_i32_b64 %4, %0\n s_bcnt1_i32_b64 %5, %1\n s_bcnt1_i32_b64 %6, %2\n s_bcnt1_i32_b64 %7, %3\n"
                             "v_cmp_gt_u32 %0, %13, %12\n v_cmp_gt_u32 %1, %14, %13\n v_cmp_gt_u32 %2, %15, %14\n v_cmp_gt_u32 %3, %12, %15\n"
                             "s_add_u32 %8, %8, %4\n s_add_u32 %9, %9, %5\n s_add_u32 %10, %10, %6\n s_add_u32 %11, %11, %7\n"
                             "s_bcnt1_i32_b64 %4, %0\n s_bcnt1_i32_b64 %5, %1\n s_bcnt1_i32_b64 %6, %2\n s_bcnt1_i32_b64 %7, %3\n"
                             "s_add_u32 %8, %8, %4\n s_add_u32 %9, %9, %5\n s_add_u32 %10, %10, %6\n s_add_u32 %11, %11, %7\n"
                             : "=&s"(m0), "=&s"(m1), "=&s"(m2), "=&s"(m3), "=&s"(t0_), "=&s"(t1_), "=&s"(t2_), "=&s"(t3_), "+s"(s0), "+s"(s1), "+s"(s2), "+s"(s3)
                             : "v"(u0), "v"(u1), "v"(u2), "v"(u3) : "scc");
            else if (KIND == K_SALU)
                asm volatile("s_add_u32 %0, %0, %8\n s_add_u32 %1, %1, %8\n s_add_u32 %2, %2, %8\n s_add_u32 %3, %3, %8\n"
                             "s_add_u32 %4, %4, %8\n s_add_u32 %5, %5, %8\n s_add_u32 %6, %6, %8\n s_add_u32 %7, %7, %8\n"
                             : "+s"(s0), "+s"(s1), "+s"(s2), "+s"(s3), "+s"(s4), "+s"(s5), "+s"(s6), "+s"(s7) : "s"(i) : "scc");
            else if (KIND == K_SALU_BCNT)
                asm volatile("s_bcnt1_i32_b64 %4, %8\n s_bcnt1_i32_b64 %5, %9\n s_bcnt1_i32_b64 %6, %8\n s_bcnt1_i32_b64 %7, %9\n"
                             "s_add_u32 %0, %0, %4\n s_add_u32 %1, %1, %5\n s_add_u32 %2, %2, %6\n s_add_u32 %3, %3, %7\n"
                             : "+s"(s0), "+s"(s1), "+s"(s2), "+s"(s3), "=&s"(t0_), "=&s"(t1_), "=&s"(t2_), "=&s"(t3_) : "s"((unsigned long long)i * 0x9E3779B97F4A7C15ull), "s"(~(unsigned long long)i) : "scc");
            else if (KIND == K_MIX_V2S1)    // 8 statements: 6 vector? no: counted as 8 + 4: see per_body()
                asm volatile("v_fma_f32 %0, %0, %12, %13\n v_fma_f32 %1, %1, %12, %13\n s_add_u32 %8, %8, %14\n v_fma_f32 %2, %2, %12, %13\n v_fma_f32 %3, %3, %12, %13\n s_add_u32 %9, %9, %14\n"
                             "v_fma_f32 %4, %4, %12, %13\n v_fma_f32 %5, %5, %12, %13\n s_add_u32 %10, %10, %14\n v_fma_f32 %6, %6, %12, %13\n v_fma_f32 %7, %7, %12, %13\n s_add_u32 %11, %11, %14\n"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7), "+s"(s0), "+s"(s1), "+s"(s2), "+s"(s3) : "v"(c), "v"(d), "s"(i) : "scc");
            else if (KIND == K_LDSR) {
                asm volatile("ds_read_b32 %0, %8\n ds_read_b32 %1, %8 offset:256\n ds_read_b32 %2, %8 offset:512\n ds_read_b32 %3, %8 offset:768\n"
                             "ds_read_b32 %4, %8 offset:1024\n ds_read_b32 %5, %8 offset:1280\n ds_read_b32 %6, %8 offset:1536\n ds_read_b32 %7, %8 offset:1792\n s_waitcnt lgkmcnt(0)\n"
                             : "=v"(a0), "=v"(a1), "=v"(a2), "=v"(a3), "=v"(a4), "=v"(a5), "=v"(a6), "=v"(a7) : "v"(u0) : "memory");
            }
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float out = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + (float)(u0 ^ u1 ^ u2 ^ u3 ^ u4 ^ u5 ^ u6 ^ u7) + (float)(f0 + f1 + f2 + f3 + f4 + f5 + f6 + f7)
              + p0.x + p1.x + p2.x + p3.x + p4.y + p5.y + p6.y + p7.y + (float)(s0 + s1 + s2 + s3 + s4 + s5 + s6 + s7);
    if (KIND == K_CMP_ONLY) out += (float)(m0 ^ m1 ^ m2 ^ m3);
    if (out == 123.456f) sink[0] = out;
    if ((threadIdx.x & 63) == 0) {
        unsigned hw, xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        Rec r; r.t0 = t0; r.t1 = t1; r.hw = hw; r.xcc = xcc;
        rec[blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = r;
    }
}

// (vector, scalar) instructions per 8-statement asm block
static void per_block(int kind, int *v, int *s) {
    *v = 8; *s = 0;
    if (kind == K_CMP_BCNT) { *v = 8; *s = 16; }
    if (kind == K_SALU) { *v = 0; *s = 8; }
    if (kind == K_SALU_BCNT) { *v = 0; *s = 8; }
    if (kind == K_MIX_V2S1) { *v = 8; *s = 4; }
}

typedef void (*kfn)(Rec *, float *, int);
static kfn fn_of(int kind) {
    switch (kind) {
#define C(K) case K: return k_rate<K>;
        C(K_FMA) C(K_ADDU) C(K_PKFMA) C(K_CMP_BCNT) C(K_CMP_ONLY) C(K_SALU) C(K_SALU_BCNT) C(K_ADD3) C(K_DPP) C(K_ADDF64) C(K_FMAF64) C(K_MULLO) C(K_CNDMASK) C(K_MIX_V2S1) C(K_LDSR)
#undef C
    }
    return nullptr;
}

int main(int argc, char **argv) {
    hipDeviceProp_t prop; CHECK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    printf("# %s, %d CUs, clock %d MHz (reported)\n", prop.name, cus, prop.clockRate / 1000);
    printf("# rate columns: wave-instructions per cycle (s_memtime ticks) per SIMD for vector kinds (v/cyc/SIMD) and per CU for scalar (s/cyc/CU); median over units\n");
    Rec *rec; float *sink; const int maxw = cus * 8 * 4;
    CHECK(hipMalloc(&rec, sizeof(Rec) * maxw)); CHECK(hipMalloc(&sink, 64));
    std::vector<Rec> h(maxw);
    const int iters = argc > 1 ? atoi(argv[1]) : 4000;
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    printf("| kind | waves/workgroup | workgroups/CU | waves/SIMD seen (min-max) | v/cyc/SIMD | s/cyc/CU | cycles per wave-instr of ONE wave | wall-clock v/cyc/SIMD @2.4GHz | wall s/cyc/CU |\n|---|---|---|---|---|---|---|---|---|\n");
    for (int kind = 0; kind < K_KINDS; kind++) {
        for (int wpw : {4, 1}) {                       // waves per workgroup: 4 = one per SIMD; 1 = a single SIMD of the CU per workgroup
            for (int w : {1, 2, 4, 8}) {
                if (wpw == 1 && !(kind == K_SALU || kind == K_SALU_BCNT || kind == K_FMA || kind == K_CMP_BCNT) ) continue;
                if (wpw == 1 && w > 4) continue;
                kfn f = fn_of(kind);
                size_t lds = (160 * 1024) / w - (w == 1 ? 0 : 512);
                if (lds > 160 * 1024 - 1024) lds = 160 * 1024 - 1024;
                CHECK(hipFuncSetAttribute((const void *)f, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
                const int grid = cus * w, nw = grid * wpw;
                float ms = 0;
                for (int rep = 0; rep < 2; rep++) {
                    CHECK(hipEventRecord(e0));
                    hipLaunchKernelGGL(f, dim3(grid), dim3(64 * wpw), lds, 0, rec, sink, iters);
                    CHECK(hipEventRecord(e1)); CHECK(hipDeviceSynchronize());
                    CHECK(hipEventElapsedTime(&ms, e0, e1));
                }
                CHECK(hipMemcpy(h.data(), rec, sizeof(Rec) * nw, hipMemcpyDeviceToHost));
                int v, s; per_block(kind, &v, &s);
                const double nv = (double)iters * 8 * v, ns = (double)iters * 8 * s;
                // HW_ID (gfx9): wave 3:0, simd 5:4, pipe 7:6, cu 11:8, sh 12, se 15:13 ; XCC_ID 3:0
                std::map<unsigned, std::vector<Rec>> by_simd, by_cu;
                for (int i = 0; i < nw; i++) {
                    unsigned cu = ((h[i].xcc & 15) << 16) | (h[i].hw & 0xff00), simd = cu | (h[i].hw & 0x30);
                    by_simd[simd].push_back(h[i]); by_cu[cu].push_back(h[i]);
                }
                auto rate = [&](std::map<unsigned, std::vector<Rec>> &m, double n, unsigned *lo, unsigned *hi) {
                    std::vector<double> r; *lo = 1u << 30; *hi = 0;
                    for (auto &kv : m) {
                        unsigned long long a = ~0ull, b = 0;
                        for (auto &x : kv.second) { a = std::min(a, x.t0); b = std::max(b, x.t1); }
                        r.push_back(n * kv.second.size() / (double)(b - a));
                        *lo = std::min<unsigned>(*lo, kv.second.size()); *hi = std::max<unsigned>(*hi, kv.second.size());
                    }
                    std::sort(r.begin(), r.end()); return r[r.size() / 2];
                };
                unsigned lo, hi, clo, chi;
                const double rv = rate(by_simd, nv, &lo, &hi), rs = rate(by_cu, ns, &clo, &chi);
                std::vector<double> one; for (int i = 0; i < nw; i++) one.push_back((double)(h[i].t1 - h[i].t0) / (nv + ns));
                std::sort(one.begin(), one.end());
                const double wall_v = nv * nw / (ms * 1e-3 * 2.4e9) / (cus * 4), wall_s = ns * nw / (ms * 1e-3 * 2.4e9) / cus;
                printf("| %s | %d | %d | %u-%u on %zu SIMDs | %.3f | %.3f | %.2f | %.3f | %.3f |\n", kind_name[kind], wpw, w, lo, hi, by_simd.size(), rv, rs, one[one.size() / 2], wall_v, wall_s);
            }
        }
    }
    return 0;
}
