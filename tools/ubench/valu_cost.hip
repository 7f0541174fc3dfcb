// Microbenchmark: issue cost of individual gfx950 vector instructions (cycles of a SIMD per wave64 instruction), by
// ENCODING and operand kind, at 1 / 2 / 8 waves per SIMD.  Companion of issue_rate.hip (same placement / timing method):
// 8 independent destination registers per kind, 64 instructions per loop body.
//   hipcc --offload-arch=gfx950 -O3 valu_cost.hip -o valu_cost && ./valu_cost
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <map>
#include <vector>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
struct Rec { unsigned long long t0, t1; unsigned hw, xcc; };

// dst = op(dst, x)            (two VGPR sources)
#define D2(op) op " %0, %0, %8\n" op " %1, %1, %8\n" op " %2, %2, %8\n" op " %3, %3, %8\n" op " %4, %4, %8\n" op " %5, %5, %8\n" op " %6, %6, %8\n" op " %7, %7, %8\n"
// dst = op(dst, x, y)         (three VGPR sources)
#define D3(op) op " %0, %0, %8, %9\n" op " %1, %1, %8, %9\n" op " %2, %2, %8, %9\n" op " %3, %3, %8, %9\n" op " %4, %4, %8, %9\n" op " %5, %5, %8, %9\n" op " %6, %6, %8, %9\n" op " %7, %7, %8, %9\n"
// dst = op(dst, x, CONST)     (two VGPR sources + an inline constant / literal)
#define D3C(op, k) op " %0, %0, %8, " k "\n" op " %1, %1, %8, " k "\n" op " %2, %2, %8, " k "\n" op " %3, %3, %8, " k "\n" op " %4, %4, %8, " k "\n" op " %5, %5, %8, " k "\n" op " %6, %6, %8, " k "\n" op " %7, %7, %8, " k "\n"
// dst = op(dst, CONST)
#define D2C(op, k) op " %0, %0, " k "\n" op " %1, %1, " k "\n" op " %2, %2, " k "\n" op " %3, %3, " k "\n" op " %4, %4, " k "\n" op " %5, %5, " k "\n" op " %6, %6, " k "\n" op " %7, %7, " k "\n"
#define D2CR(op, k) op " %0, " k ", %0\n" op " %1, " k ", %1\n" op " %2, " k ", %2\n" op " %3, " k ", %3\n" op " %4, " k ", %4\n" op " %5, " k ", %5\n" op " %6, " k ", %6\n" op " %7, " k ", %7\n"
// dst = op(dst)
#define D1(op) op " %0, %0\n" op " %1, %1\n" op " %2, %2\n" op " %3, %3\n" op " %4, %4\n" op " %5, %5\n" op " %6, %6\n" op " %7, %7\n"
// suffix form (dpp etc.)
#define D2S(op, suf) op " %0, %0, %8 " suf "\n" op " %1, %1, %8 " suf "\n" op " %2, %2, %8 " suf "\n" op " %3, %3, %8 " suf "\n" op " %4, %4, %8 " suf "\n" op " %5, %5, %8 " suf "\n" op " %6, %6, %8 " suf "\n" op " %7, %7, %8 " suf "\n"
#define D1S(op, suf) op " %0, %0 " suf "\n" op " %1, %1 " suf "\n" op " %2, %2 " suf "\n" op " %3, %3 " suf "\n" op " %4, %4 " suf "\n" op " %5, %5 " suf "\n" op " %6, %6 " suf "\n" op " %7, %7 " suf "\n"

#define KINDS(X) \
  X(0,  "v_add_u32 (VOP2 e32)",                 D2("v_add_u32_e32")) \
  X(1,  "v_add_f32 (VOP2 e32)",                 D2("v_add_f32_e32")) \
  X(2,  "v_mul_f32 (VOP2 e32)",                 D2("v_mul_f32_e32")) \
  X(3,  "v_fmac_f32 (VOP2 e32: d += a*b)",      "v_fmac_f32_e32 %0, %8, %9\n v_fmac_f32_e32 %1, %8, %9\n v_fmac_f32_e32 %2, %8, %9\n v_fmac_f32_e32 %3, %8, %9\n v_fmac_f32_e32 %4, %8, %9\n v_fmac_f32_e32 %5, %8, %9\n v_fmac_f32_e32 %6, %8, %9\n v_fmac_f32_e32 %7, %8, %9\n") \
  X(4,  "v_fma_f32 d,d,v,v (VOP3, 3 VGPR)",     D3("v_fma_f32")) \
  X(5,  "v_fma_f32 d,d,v,1.0 (VOP3, 2 VGPR)",   D3C("v_fma_f32", "1.0")) \
  X(6,  "v_add_f32 e64 (VOP3 encoding)",        D2("v_add_f32_e64")) \
  X(7,  "v_and_b32 (VOP2 e32)",                 D2("v_and_b32_e32")) \
  X(8,  "v_xor_b32 (VOP2 e32)",                 D2("v_xor_b32_e32")) \
  X(9,  "v_lshlrev_b32 1,d (VOP2 e32)",         D2CR("v_lshlrev_b32_e32", "1")) \
  X(10, "v_max_f32 (VOP2 e32)",                 D2("v_max_f32_e32")) \
  X(11, "v_mov_b32 d,d (VOP1)",                 D1("v_mov_b32_e32")) \
  X(12, "v_cvt_f32_u32 (VOP1)",                 D1("v_cvt_f32_u32_e32")) \
  X(13, "v_bfe_u32 d,d,v,8 (VOP3)",             D3C("v_bfe_u32", "8")) \
  X(14, "v_lshl_add_u32 d,d,1,v (VOP3)",        "v_lshl_add_u32 %0, %0, 1, %8\n v_lshl_add_u32 %1, %1, 1, %8\n v_lshl_add_u32 %2, %2, 1, %8\n v_lshl_add_u32 %3, %3, 1, %8\n v_lshl_add_u32 %4, %4, 1, %8\n v_lshl_add_u32 %5, %5, 1, %8\n v_lshl_add_u32 %6, %6, 1, %8\n v_lshl_add_u32 %7, %7, 1, %8\n") \
  X(15, "v_mad_u32_u24 (VOP3)",                 D3("v_mad_u32_u24")) \
  X(16, "v_mul_u32_u24 (VOP2 e32)",             D2("v_mul_u32_u24_e32")) \
  X(17, "v_mul_lo_u32 (VOP3)",                  D2("v_mul_lo_u32")) \
  X(18, "v_cmp_gt_u32 vcc (VOPC e32)",          "v_cmp_gt_u32_e32 vcc, %0, %8\n v_cmp_gt_u32_e32 vcc, %1, %8\n v_cmp_gt_u32_e32 vcc, %2, %8\n v_cmp_gt_u32_e32 vcc, %3, %8\n v_cmp_gt_u32_e32 vcc, %4, %8\n v_cmp_gt_u32_e32 vcc, %5, %8\n v_cmp_gt_u32_e32 vcc, %6, %8\n v_cmp_gt_u32_e32 vcc, %7, %8\n") \
  X(19, "v_cmp_gt_u32 s[..] (VOP3 e64)",        "v_cmp_gt_u32_e64 s[40:41], %0, %8\n v_cmp_gt_u32_e64 s[42:43], %1, %8\n v_cmp_gt_u32_e64 s[44:45], %2, %8\n v_cmp_gt_u32_e64 s[46:47], %3, %8\n v_cmp_gt_u32_e64 s[40:41], %4, %8\n v_cmp_gt_u32_e64 s[42:43], %5, %8\n v_cmp_gt_u32_e64 s[44:45], %6, %8\n v_cmp_gt_u32_e64 s[46:47], %7, %8\n") \
  X(20, "v_cmp e32 + v_addc_co_u32 e32 (count in lanes; 2 instr)", "v_cmp_gt_u32_e32 vcc, %0, %8\n v_addc_co_u32_e32 %4, vcc, 0, %4, vcc\n v_cmp_gt_u32_e32 vcc, %1, %8\n v_addc_co_u32_e32 %5, vcc, 0, %5, vcc\n v_cmp_gt_u32_e32 vcc, %2, %8\n v_addc_co_u32_e32 %6, vcc, 0, %6, vcc\n v_cmp_gt_u32_e32 vcc, %3, %8\n v_addc_co_u32_e32 %7, vcc, 0, %7, vcc\n") \
  X(21, "v_cndmask_b32 d,d,v,vcc (VOP2 e32; vcc constant)", "v_cndmask_b32_e32 %0, %0, %8, vcc\n v_cndmask_b32_e32 %1, %1, %8, vcc\n v_cndmask_b32_e32 %2, %2, %8, vcc\n v_cndmask_b32_e32 %3, %3, %8, vcc\n v_cndmask_b32_e32 %4, %4, %8, vcc\n v_cndmask_b32_e32 %5, %5, %8, vcc\n v_cndmask_b32_e32 %6, %6, %8, vcc\n v_cndmask_b32_e32 %7, %7, %8, vcc\n") \
  X(22, "v_cndmask_b32 d,d,v,s[40:41] (VOP3 e64)", "v_cndmask_b32_e64 %0, %0, %8, s[40:41]\n v_cndmask_b32_e64 %1, %1, %8, s[40:41]\n v_cndmask_b32_e64 %2, %2, %8, s[40:41]\n v_cndmask_b32_e64 %3, %3, %8, s[40:41]\n v_cndmask_b32_e64 %4, %4, %8, s[40:41]\n v_cndmask_b32_e64 %5, %5, %8, s[40:41]\n v_cndmask_b32_e64 %6, %6, %8, s[40:41]\n v_cndmask_b32_e64 %7, %7, %8, s[40:41]\n") \
  X(23, "v_cmp e32 + v_cndmask e32 (2 instr)",  "v_cmp_gt_u32_e32 vcc, %0, %8\n v_cndmask_b32_e32 %4, %4, %8, vcc\n v_cmp_gt_u32_e32 vcc, %1, %8\n v_cndmask_b32_e32 %5, %5, %8, vcc\n v_cmp_gt_u32_e32 vcc, %2, %8\n v_cndmask_b32_e32 %6, %6, %8, vcc\n v_cmp_gt_u32_e32 vcc, %3, %8\n v_cndmask_b32_e32 %7, %7, %8, vcc\n") \
  X(24, "v_mov_b32 dpp row_shr:1",              D1S("v_mov_b32_dpp", "row_shr:1 row_mask:0xf bank_mask:0xf")) \
  X(25, "v_add_u32 dpp row_shr:1",              D2S("v_add_u32_dpp", "row_shr:1 row_mask:0xf bank_mask:0xf")) \
  X(26, "v_add_f32 dpp row_shr:1",              D2S("v_add_f32_dpp", "row_shr:1 row_mask:0xf bank_mask:0xf")) \
  X(27, "v_add_u32 dpp row_bcast:31",           D2S("v_add_u32_dpp", "row_bcast:31 row_mask:0xf bank_mask:0xf")) \
  X(28, "v_add_u32 sdwa (BYTE_0)",              D2S("v_add_u32_sdwa", "dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_0 src1_sel:DWORD")) \
  X(29, "v_readlane_b32 s,v,5",                 "v_readlane_b32 s40, %0, 5\n v_readlane_b32 s41, %1, 5\n v_readlane_b32 s42, %2, 5\n v_readlane_b32 s43, %3, 5\n v_readlane_b32 s44, %4, 5\n v_readlane_b32 s45, %5, 5\n v_readlane_b32 s46, %6, 5\n v_readlane_b32 s47, %7, 5\n") \
  X(30, "v_writelane_b32 v,s,5",                "v_writelane_b32 %0, s8, 5\n v_writelane_b32 %1, s8, 5\n v_writelane_b32 %2, s8, 5\n v_writelane_b32 %3, s8, 5\n v_writelane_b32 %4, s8, 5\n v_writelane_b32 %5, s8, 5\n v_writelane_b32 %6, s8, 5\n v_writelane_b32 %7, s8, 5\n") \
  X(31, "v_readfirstlane_b32",                  "v_readfirstlane_b32 s40, %0\n v_readfirstlane_b32 s41, %1\n v_readfirstlane_b32 s42, %2\n v_readfirstlane_b32 s43, %3\n v_readfirstlane_b32 s44, %4\n v_readfirstlane_b32 s45, %5\n v_readfirstlane_b32 s46, %6\n v_readfirstlane_b32 s47, %7\n") \
  X(32, "v_log_f32 (trans)",                    D1("v_log_f32_e32")) \
  X(33, "v_exp_f32 (trans)",                    D1("v_exp_f32_e32")) \
  X(34, "v_rcp_f32 (trans)",                    D1("v_rcp_f32_e32")) \
  X(35, "v_sqrt_f32 (trans)",                   D1("v_sqrt_f32_e32")) \
  X(36, "v_mbcnt_lo_u32_b32 d,s,d (VOP3)",      "v_mbcnt_lo_u32_b32 %0, s8, %0\n v_mbcnt_lo_u32_b32 %1, s8, %1\n v_mbcnt_lo_u32_b32 %2, s8, %2\n v_mbcnt_lo_u32_b32 %3, s8, %3\n v_mbcnt_lo_u32_b32 %4, s8, %4\n v_mbcnt_lo_u32_b32 %5, s8, %5\n v_mbcnt_lo_u32_b32 %6, s8, %6\n v_mbcnt_lo_u32_b32 %7, s8, %7\n") \
  X(37, "v_perm_b32 (VOP3)",                    D3("v_perm_b32")) \
  X(38, "v_alignbit_b32 d,d,v,8 (VOP3)",        D3C("v_alignbit_b32", "8")) \
  X(39, "v_pk_mul_f32 (VOP3P, 64-bit regs)",    "PK") \
  X(40, "v_pk_add_f32 (VOP3P)",                 "PK") \
  X(41, "v_add_f64",                            "F64") \
  X(42, "v_mul_f64",                            "F64") \
  X(43, "v_cvt_f64_f32 (VOP1)",                 "CVT64") \
  X(44, "v_add_co_u32 e32 (vcc out)",           "v_add_co_u32_e32 %0, vcc, %0, %8\n v_add_co_u32_e32 %1, vcc, %1, %8\n v_add_co_u32_e32 %2, vcc, %2, %8\n v_add_co_u32_e32 %3, vcc, %3, %8\n v_add_co_u32_e32 %4, vcc, %4, %8\n v_add_co_u32_e32 %5, vcc, %5, %8\n v_add_co_u32_e32 %6, vcc, %6, %8\n v_add_co_u32_e32 %7, vcc, %7, %8\n") \
  X(45, "v_and_or_b32 (VOP3)",                  D3("v_and_or_b32")) \
  X(46, "v_min3_u32 / v_max3 (VOP3)",           D3("v_max3_u32")) \
  X(47, "v_med3_f32 (VOP3)",                    D3("v_med3_f32")) \
  X(48, "v_sub_f32 e32",                        D2("v_sub_f32_e32")) \
  X(49, "v_cmp_class / v_cmp_lt_f32 e32 vcc",   "v_cmp_lt_f32_e32 vcc, %0, %8\n v_cmp_lt_f32_e32 vcc, %1, %8\n v_cmp_lt_f32_e32 vcc, %2, %8\n v_cmp_lt_f32_e32 vcc, %3, %8\n v_cmp_lt_f32_e32 vcc, %4, %8\n v_cmp_lt_f32_e32 vcc, %5, %8\n v_cmp_lt_f32_e32 vcc, %6, %8\n v_cmp_lt_f32_e32 vcc, %7, %8\n")

template <int KIND>
__global__ void k_cost(Rec *rec, float *sink, int iters) {
    extern __shared__ float lds[];
    unsigned u0 = threadIdx.x, u1 = u0 * 3, u2 = u0 * 5, u3 = u0 * 7, u4 = u0 + 9, u5 = u0 + 11, u6 = u0 + 13, u7 = u0 + 15;
    unsigned x = threadIdx.x * 2654435761u, y = threadIdx.x + 77;
    double f0 = threadIdx.x, f1 = f0 + 1, f2 = f0 + 2, f3 = f0 + 3, f4 = f0 + 4, f5 = f0 + 5, f6 = f0 + 6, f7 = f0 + 7;
    const double dc = 0.999;
    float2 p0 = {1.f, 2.f}, p1 = p0, p2 = p0, p3 = p0, p4 = p0, p5 = p0, p6 = p0, p7 = p0;
    const float2 pc = {0.999f, 0.998f};
    __builtin_amdgcn_s_barrier();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int r = 0; r < 8; r++) {
#define X(id, name, body) if (KIND == id) { \
            if (id == 39) asm volatile("v_pk_mul_f32 %0, %0, %8\n v_pk_mul_f32 %1, %1, %8\n v_pk_mul_f32 %2, %2, %8\n v_pk_mul_f32 %3, %3, %8\n v_pk_mul_f32 %4, %4, %8\n v_pk_mul_f32 %5, %5, %8\n v_pk_mul_f32 %6, %6, %8\n v_pk_mul_f32 %7, %7, %8\n" : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3), "+v"(p4), "+v"(p5), "+v"(p6), "+v"(p7) : "v"(pc)); \
            else if (id == 40) asm volatile("v_pk_add_f32 %0, %0, %8\n v_pk_add_f32 %1, %1, %8\n v_pk_add_f32 %2, %2, %8\n v_pk_add_f32 %3, %3, %8\n v_pk_add_f32 %4, %4, %8\n v_pk_add_f32 %5, %5, %8\n v_pk_add_f32 %6, %6, %8\n v_pk_add_f32 %7, %7, %8\n" : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3), "+v"(p4), "+v"(p5), "+v"(p6), "+v"(p7) : "v"(pc)); \
            else if (id == 41) asm volatile("v_add_f64 %0, %0, %8\n v_add_f64 %1, %1, %8\n v_add_f64 %2, %2, %8\n v_add_f64 %3, %3, %8\n v_add_f64 %4, %4, %8\n v_add_f64 %5, %5, %8\n v_add_f64 %6, %6, %8\n v_add_f64 %7, %7, %8\n" : "+v"(f0), "+v"(f1), "+v"(f2), "+v"(f3), "+v"(f4), "+v"(f5), "+v"(f6), "+v"(f7) : "v"(dc)); \
            else if (id == 42) asm volatile("v_mul_f64 %0, %0, %8\n v_mul_f64 %1, %1, %8\n v_mul_f64 %2, %2, %8\n v_mul_f64 %3, %3, %8\n v_mul_f64 %4, %4, %8\n v_mul_f64 %5, %5, %8\n v_mul_f64 %6, %6, %8\n v_mul_f64 %7, %7, %8\n" : "+v"(f0), "+v"(f1), "+v"(f2), "+v"(f3), "+v"(f4), "+v"(f5), "+v"(f6), "+v"(f7) : "v"(dc)); \
            else if (id == 43) asm volatile("v_cvt_f64_f32_e32 %0, %8\n v_cvt_f64_f32_e32 %1, %8\n v_cvt_f64_f32_e32 %2, %8\n v_cvt_f64_f32_e32 %3, %8\n v_cvt_f64_f32_e32 %4, %8\n v_cvt_f64_f32_e32 %5, %8\n v_cvt_f64_f32_e32 %6, %8\n v_cvt_f64_f32_e32 %7, %8\n" : "+v"(f0), "+v"(f1), "+v"(f2), "+v"(f3), "+v"(f4), "+v"(f5), "+v"(f6), "+v"(f7) : "v"(x)); \
            else asm volatile(body : "+v"(u0), "+v"(u1), "+v"(u2), "+v"(u3), "+v"(u4), "+v"(u5), "+v"(u6), "+v"(u7) : "v"(x), "v"(y) : "vcc", "s8", "s40", "s41", "s42", "s43", "s44", "s45", "s46", "s47"); }
            KINDS(X)
#undef X
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float out = (float)(u0 ^ u1 ^ u2 ^ u3 ^ u4 ^ u5 ^ u6 ^ u7) + (float)(f0 + f1 + f2 + f3 + f4 + f5 + f6 + f7) + p0.x + p1.x + p2.x + p3.x + p4.y + p5.y + p6.y + p7.y;
    if (out == 123.456f) sink[0] = out;
    if ((threadIdx.x & 63) == 0) {
        unsigned hw, xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        Rec r; r.t0 = t0; r.t1 = t1; r.hw = hw; r.xcc = xcc;
        rec[blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = r;
    }
}
typedef void (*kfn)(Rec *, float *, int);
struct Kind { int id; const char *name; kfn f; };
#define X(id, name, body) {id, name, k_cost<id>},
static Kind kinds[] = { KINDS(X) };
#undef X

int main(int argc, char **argv) {
    hipDeviceProp_t prop; CHECK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    Rec *rec; float *sink; const int maxw = cus * 8 * 4;
    CHECK(hipMalloc(&rec, sizeof(Rec) * maxw)); CHECK(hipMalloc(&sink, 64));
    std::vector<Rec> h(maxw);
    const int iters = argc > 1 ? atoi(argv[1]) : 2000;
    printf("# cycles of a SIMD per wave64 instruction = 1 / (instructions per cycle per SIMD, median over the SIMDs); 8 independent registers per wave\n");
    printf("| instruction | 1 wave/SIMD | 2 waves/SIMD | 4 waves/SIMD | 8 waves/SIMD |\n|---|---|---|---|---|\n");
    for (auto &kd : kinds) {
        printf("| %s |", kd.name);
        for (int w : {1, 2, 4, 8}) {
            size_t lds = (160 * 1024) / w - (w == 1 ? 1024 : 512);
            CHECK(hipFuncSetAttribute((const void *)kd.f, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            const int grid = cus * w, nw = grid * 4;
            for (int rep = 0; rep < 2; rep++) { hipLaunchKernelGGL(kd.f, dim3(grid), dim3(256), lds, 0, rec, sink, iters); CHECK(hipDeviceSynchronize()); }
            CHECK(hipMemcpy(h.data(), rec, sizeof(Rec) * nw, hipMemcpyDeviceToHost));
            std::map<unsigned, std::vector<Rec>> by_simd;
            for (int i = 0; i < nw; i++) by_simd[((h[i].xcc & 15) << 16) | (h[i].hw & 0xff30)].push_back(h[i]);
            std::vector<double> r; bool even = true;
            for (auto &kv : by_simd) {
                unsigned long long a = ~0ull, b = 0;
                for (auto &x : kv.second) { a = std::min(a, x.t0); b = std::max(b, x.t1); }
                r.push_back((double)(b - a) / ((double)iters * 64 * kv.second.size()));
                even = even && (int)kv.second.size() == w;
            }
            std::sort(r.begin(), r.end());
            printf(" %.2f%s |", r[r.size() / 2], even ? "" : " (uneven placement)");
        }
        printf("\n"); fflush(stdout);
    }
    return 0;
}
