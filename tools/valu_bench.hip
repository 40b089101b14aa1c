// valu_bench.hip — measures gfx950 per-instruction VALU issue cost (cycles per wave64 instruction per SIMD)
// with the SIMDs saturated (8 waves/SIMD).  Diagnostic tool; not part of the product library.
//   hipcc --offload-arch=gfx950 -O2 tools/valu_bench.hip -o tools/valu_bench && tools/valu_bench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <string>

#define REP8(x) x x x x x x x x
#define REP64(x) REP8(REP8(x))

#define KERNEL(name, body)                                                          \
  __global__ __launch_bounds__(256) void name(float* out, int iters, float seed) { \
    float a0 = seed + threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7; \
    float b = seed * 0.5f + 1.0f, c = seed * 0.25f + 0.5f;                         \
    for (int i = 0; i < iters; ++i) {                                               \
      asm volatile(REP8(body)                                                       \
                   : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) \
                   : "v"(b), "v"(c) : "vcc", "scc", "s20", "s21", "s22", "s23", "s24", "s25", "s26", "s27", "v100", "v101", "v102", "v103", "v104", "v105", "v106", "v107", "v108", "v109", "v110", "v111", "v112", "v113", "v114", "v115"); \
    }                                                                               \
    out[blockIdx.x * 256 + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;  \
  }

// each body = 8 independent instructions (one per accumulator); REP8 -> 64 instructions per loop trip
KERNEL(k_fma, "v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n v_fma_f32 %3, %3, %8, %9\n v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %5, %5, %8, %9\n v_fma_f32 %6, %6, %8, %9\n v_fma_f32 %7, %7, %8, %9\n")
KERNEL(k_mul, "v_mul_f32 %0, %0, %8\n v_mul_f32 %1, %1, %8\n v_mul_f32 %2, %2, %8\n v_mul_f32 %3, %3, %8\n v_mul_f32 %4, %4, %8\n v_mul_f32 %5, %5, %8\n v_mul_f32 %6, %6, %8\n v_mul_f32 %7, %7, %8\n")
KERNEL(k_add, "v_add_f32 %0, %0, %8\n v_add_f32 %1, %1, %8\n v_add_f32 %2, %2, %8\n v_add_f32 %3, %3, %8\n v_add_f32 %4, %4, %8\n v_add_f32 %5, %5, %8\n v_add_f32 %6, %6, %8\n v_add_f32 %7, %7, %8\n")
KERNEL(k_pkfma, "v_pk_fma_f32 v[100:101], v[100:101], v[102:103], v[104:105]\n v_pk_fma_f32 v[106:107], v[106:107], v[102:103], v[104:105]\n v_pk_fma_f32 v[108:109], v[108:109], v[102:103], v[104:105]\n v_pk_fma_f32 v[110:111], v[110:111], v[102:103], v[104:105]\n v_pk_fma_f32 v[112:113], v[112:113], v[102:103], v[104:105]\n v_pk_fma_f32 v[114:115], v[114:115], v[102:103], v[104:105]\n v_pk_fma_f32 v[100:101], v[100:101], v[102:103], v[104:105]\n v_pk_fma_f32 v[106:107], v[106:107], v[102:103], v[104:105]\n")
KERNEL(k_pkmul, "v_pk_mul_f32 v[100:101], v[100:101], v[102:103]\n v_pk_mul_f32 v[106:107], v[106:107], v[102:103]\n v_pk_mul_f32 v[108:109], v[108:109], v[102:103]\n v_pk_mul_f32 v[110:111], v[110:111], v[102:103]\n v_pk_mul_f32 v[112:113], v[112:113], v[102:103]\n v_pk_mul_f32 v[114:115], v[114:115], v[102:103]\n v_pk_mul_f32 v[100:101], v[100:101], v[102:103]\n v_pk_mul_f32 v[106:107], v[106:107], v[102:103]\n")
KERNEL(k_rcp, "v_rcp_f32 %0, %0\n v_rcp_f32 %1, %1\n v_rcp_f32 %2, %2\n v_rcp_f32 %3, %3\n v_rcp_f32 %4, %4\n v_rcp_f32 %5, %5\n v_rcp_f32 %6, %6\n v_rcp_f32 %7, %7\n")
KERNEL(k_sqrt, "v_sqrt_f32 %0, %0\n v_sqrt_f32 %1, %1\n v_sqrt_f32 %2, %2\n v_sqrt_f32 %3, %3\n v_sqrt_f32 %4, %4\n v_sqrt_f32 %5, %5\n v_sqrt_f32 %6, %6\n v_sqrt_f32 %7, %7\n")
KERNEL(k_readlane, "v_readlane_b32 s20, %0, 3\n v_readlane_b32 s21, %1, 5\n v_readlane_b32 s22, %2, 7\n v_readlane_b32 s23, %3, 9\n v_readlane_b32 s24, %4, 11\n v_readlane_b32 s25, %5, 13\n v_readlane_b32 s26, %6, 15\n v_readlane_b32 s27, %7, 17\n")
KERNEL(k_readlane_use, "v_readlane_b32 s20, %0, 3\n v_mul_f32 %1, s20, %1\n v_readlane_b32 s22, %2, 7\n v_mul_f32 %3, s22, %3\n v_readlane_b32 s24, %4, 11\n v_mul_f32 %5, s24, %5\n v_readlane_b32 s26, %6, 15\n v_mul_f32 %7, s26, %7\n")
KERNEL(k_cmp, "v_cmp_lt_f32 vcc, %0, %8\n v_cmp_lt_f32 vcc, %1, %8\n v_cmp_lt_f32 vcc, %2, %8\n v_cmp_lt_f32 vcc, %3, %8\n v_cmp_lt_f32 vcc, %4, %8\n v_cmp_lt_f32 vcc, %5, %8\n v_cmp_lt_f32 vcc, %6, %8\n v_cmp_lt_f32 vcc, %7, %8\n")
KERNEL(k_cmp_sgpr, "v_cmp_lt_f32 s[20:21], %0, %8\n v_cmp_lt_f32 s[22:23], %1, %8\n v_cmp_lt_f32 s[24:25], %2, %8\n v_cmp_lt_f32 s[26:27], %3, %8\n v_cmp_lt_f32 s[20:21], %4, %8\n v_cmp_lt_f32 s[22:23], %5, %8\n v_cmp_lt_f32 s[24:25], %6, %8\n v_cmp_lt_f32 s[26:27], %7, %8\n")
KERNEL(k_cndmask, "v_cndmask_b32 %0, %0, %8, vcc\n v_cndmask_b32 %1, %1, %8, vcc\n v_cndmask_b32 %2, %2, %8, vcc\n v_cndmask_b32 %3, %3, %8, vcc\n v_cndmask_b32 %4, %4, %8, vcc\n v_cndmask_b32 %5, %5, %8, vcc\n v_cndmask_b32 %6, %6, %8, vcc\n v_cndmask_b32 %7, %7, %8, vcc\n")
KERNEL(k_divscale, "v_div_scale_f32 %0, vcc, %0, %8, %0\n v_div_scale_f32 %1, vcc, %1, %8, %1\n v_div_scale_f32 %2, vcc, %2, %8, %2\n v_div_scale_f32 %3, vcc, %3, %8, %3\n v_div_scale_f32 %4, vcc, %4, %8, %4\n v_div_scale_f32 %5, vcc, %5, %8, %5\n v_div_scale_f32 %6, vcc, %6, %8, %6\n v_div_scale_f32 %7, vcc, %7, %8, %7\n")
KERNEL(k_divfmas, "v_div_fmas_f32 %0, %0, %8, %9\n v_div_fmas_f32 %1, %1, %8, %9\n v_div_fmas_f32 %2, %2, %8, %9\n v_div_fmas_f32 %3, %3, %8, %9\n v_div_fmas_f32 %4, %4, %8, %9\n v_div_fmas_f32 %5, %5, %8, %9\n v_div_fmas_f32 %6, %6, %8, %9\n v_div_fmas_f32 %7, %7, %8, %9\n")
KERNEL(k_divfixup, "v_div_fixup_f32 %0, %0, %8, %9\n v_div_fixup_f32 %1, %1, %8, %9\n v_div_fixup_f32 %2, %2, %8, %9\n v_div_fixup_f32 %3, %3, %8, %9\n v_div_fixup_f32 %4, %4, %8, %9\n v_div_fixup_f32 %5, %5, %8, %9\n v_div_fixup_f32 %6, %6, %8, %9\n v_div_fixup_f32 %7, %7, %8, %9\n")
KERNEL(k_xor, "v_xor_b32 %0, %0, %8\n v_xor_b32 %1, %1, %8\n v_xor_b32 %2, %2, %8\n v_xor_b32 %3, %3, %8\n v_xor_b32 %4, %4, %8\n v_xor_b32 %5, %5, %8\n v_xor_b32 %6, %6, %8\n v_xor_b32 %7, %7, %8\n")
KERNEL(k_cvt, "v_cvt_f32_u32 %0, %0\n v_cvt_f32_u32 %1, %1\n v_cvt_f32_u32 %2, %2\n v_cvt_f32_u32 %3, %3\n v_cvt_f32_u32 %4, %4\n v_cvt_f32_u32 %5, %5\n v_cvt_f32_u32 %6, %6\n v_cvt_f32_u32 %7, %7\n")
KERNEL(k_fma_salu, "v_fma_f32 %0, %0, %8, %9\n s_add_u32 s20, s20, 1\n v_fma_f32 %1, %1, %8, %9\n s_add_u32 s21, s21, 1\n v_fma_f32 %2, %2, %8, %9\n s_add_u32 s22, s22, 1\n v_fma_f32 %3, %3, %8, %9\n s_add_u32 s23, s23, 1\n")
KERNEL(k_salu, "s_add_u32 s20, s20, 1\n s_add_u32 s21, s21, 1\n s_add_u32 s22, s22, 1\n s_add_u32 s23, s23, 1\n s_add_u32 s24, s24, 1\n s_add_u32 s25, s25, 1\n s_add_u32 s26, s26, 1\n s_add_u32 s27, s27, 1\n")
KERNEL(k_fma_dep, "v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %0, %0, %8, %9\n")
KERNEL(k_fma_sgpr, "v_fma_f32 %0, %0, s20, %9\n v_fma_f32 %1, %1, s21, %9\n v_fma_f32 %2, %2, s22, %9\n v_fma_f32 %3, %3, s23, %9\n v_fma_f32 %4, %4, s20, %9\n v_fma_f32 %5, %5, s21, %9\n v_fma_f32 %6, %6, s22, %9\n v_fma_f32 %7, %7, s23, %9\n")


KERNEL(k_cndmask_s, "v_cndmask_b32 %0, %0, %8, s[20:21]\n v_cndmask_b32 %1, %1, %8, s[20:21]\n v_cndmask_b32 %2, %2, %8, s[22:23]\n v_cndmask_b32 %3, %3, %8, s[22:23]\n v_cndmask_b32 %4, %4, %8, s[20:21]\n v_cndmask_b32 %5, %5, %8, s[20:21]\n v_cndmask_b32 %6, %6, %8, s[22:23]\n v_cndmask_b32 %7, %7, %8, s[22:23]\n")
KERNEL(k_cmp_cnd, "v_cmp_lt_f32 vcc, %0, %8\n v_cndmask_b32 %1, %1, %8, vcc\n v_cmp_lt_f32 vcc, %2, %8\n v_cndmask_b32 %3, %3, %8, vcc\n v_cmp_lt_f32 vcc, %4, %8\n v_cndmask_b32 %5, %5, %8, vcc\n v_cmp_lt_f32 vcc, %6, %8\n v_cndmask_b32 %7, %7, %8, vcc\n")
KERNEL(k_max, "v_max_f32 %0, %0, %8\n v_max_f32 %1, %1, %8\n v_max_f32 %2, %2, %8\n v_max_f32 %3, %3, %8\n v_max_f32 %4, %4, %8\n v_max_f32 %5, %5, %8\n v_max_f32 %6, %6, %8\n v_max_f32 %7, %7, %8\n")
KERNEL(k_bfi, "v_bfi_b32 %0, %0, %8, %9\n v_bfi_b32 %1, %1, %8, %9\n v_bfi_b32 %2, %2, %8, %9\n v_bfi_b32 %3, %3, %8, %9\n v_bfi_b32 %4, %4, %8, %9\n v_bfi_b32 %5, %5, %8, %9\n v_bfi_b32 %6, %6, %8, %9\n v_bfi_b32 %7, %7, %8, %9\n")
KERNEL(k_mov, "v_mov_b32 %0, %8\n v_mov_b32 %1, %8\n v_mov_b32 %2, %8\n v_mov_b32 %3, %8\n v_mov_b32 %4, %8\n v_mov_b32 %5, %8\n v_mov_b32 %6, %8\n v_mov_b32 %7, %8\n")
KERNEL(k_dpp, "v_mov_b32_dpp %0, %1 row_shr:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %1, %2 row_shr:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %2, %3 row_shr:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %3, %4 row_shr:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %4, %5 row_shr:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %5, %6 row_shr:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %6, %7 row_shr:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %7, %0 row_shr:1 row_mask:0xf bank_mask:0xf\n")
KERNEL(k_mul_lit, "v_mul_f32 %0, 0x3f9d70a4, %0\n v_mul_f32 %1, 0x3f9d70a4, %1\n v_mul_f32 %2, 0x3f9d70a4, %2\n v_mul_f32 %3, 0x3f9d70a4, %3\n v_mul_f32 %4, 0x3f9d70a4, %4\n v_mul_f32 %5, 0x3f9d70a4, %5\n v_mul_f32 %6, 0x3f9d70a4, %6\n v_mul_f32 %7, 0x3f9d70a4, %7\n")
KERNEL(k_mul_inl, "v_mul_f32 %0, 2.0, %0\n v_mul_f32 %1, 2.0, %1\n v_mul_f32 %2, 2.0, %2\n v_mul_f32 %3, 2.0, %3\n v_mul_f32 %4, 0.5, %4\n v_mul_f32 %5, 0.5, %5\n v_mul_f32 %6, 0.5, %6\n v_mul_f32 %7, 0.5, %7\n")
KERNEL(k_fma_neg, "v_fma_f32 %0, -%0, %8, %9\n v_fma_f32 %1, -%1, %8, %9\n v_fma_f32 %2, -%2, %8, %9\n v_fma_f32 %3, -%3, %8, %9\n v_fma_f32 %4, -%4, %8, %9\n v_fma_f32 %5, -%5, %8, %9\n v_fma_f32 %6, -%6, %8, %9\n v_fma_f32 %7, -%7, %8, %9\n")
KERNEL(k_sub_e64, "v_sub_f32_e64 %0, %0, |%8|\n v_sub_f32_e64 %1, %1, |%8|\n v_sub_f32_e64 %2, %2, |%8|\n v_sub_f32_e64 %3, %3, |%8|\n v_sub_f32_e64 %4, %4, |%8|\n v_sub_f32_e64 %5, %5, |%8|\n v_sub_f32_e64 %6, %6, |%8|\n v_sub_f32_e64 %7, %7, |%8|\n")
KERNEL(k_readfirst, "v_readfirstlane_b32 s20, %0\n v_readfirstlane_b32 s21, %1\n v_readfirstlane_b32 s22, %2\n v_readfirstlane_b32 s23, %3\n v_readfirstlane_b32 s24, %4\n v_readfirstlane_b32 s25, %5\n v_readfirstlane_b32 s26, %6\n v_readfirstlane_b32 s27, %7\n")
KERNEL(k_cmp_u32, "v_cmp_lt_u32 vcc, %0, %8\n v_cmp_lt_u32 vcc, %1, %8\n v_cmp_lt_u32 vcc, %2, %8\n v_cmp_lt_u32 vcc, %3, %8\n v_cmp_lt_u32 vcc, %4, %8\n v_cmp_lt_u32 vcc, %5, %8\n v_cmp_lt_u32 vcc, %6, %8\n v_cmp_lt_u32 vcc, %7, %8\n")
KERNEL(k_cmpx, "v_cmpx_lt_f32 exec, %0, %8\n s_mov_b64 exec, -1\n v_cmpx_lt_f32 exec, %1, %8\n s_mov_b64 exec, -1\n v_cmpx_lt_f32 exec, %2, %8\n s_mov_b64 exec, -1\n v_cmpx_lt_f32 exec, %3, %8\n s_mov_b64 exec, -1\n")

// LDS broadcast reads: every lane reads the same 16 bytes
__global__ __launch_bounds__(256) void k_lds_b128(float* out, int iters, float seed) {
  __shared__ float4 buf[256];
  buf[threadIdx.x] = make_float4(seed, seed + 1, seed + 2, seed + 3);
  __syncthreads();
  float4 acc = make_float4(0, 0, 0, 0);
  int idx = (int)seed & 255;
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int u = 0; u < 16; ++u) {
      float4 v;
      asm volatile("ds_read_b128 %0, %1 offset:%2\n" : "=v"(v) : "v"(idx * 16), "i"(u * 16));
      asm volatile("s_waitcnt lgkmcnt(8)");
      acc.x += v.x;
    }
  }
  asm volatile("s_waitcnt lgkmcnt(0)");
  out[blockIdx.x * 256 + threadIdx.x] = acc.x + acc.y;
}
// same, with 8 plain FMAs per read (does the LDS read steal VALU issue?)
__global__ __launch_bounds__(256) void k_lds_b128_fma(float* out, int iters, float seed) {
  __shared__ float4 buf[256];
  buf[threadIdx.x] = make_float4(seed, seed + 1, seed + 2, seed + 3);
  __syncthreads();
  float a0 = seed, a1 = seed + 1, a2 = seed + 2, a3 = seed + 3;
  float b = seed * 0.5f, c = seed * 0.25f;
  int idx = (int)seed & 255;
  float4 v = make_float4(0, 0, 0, 0);
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      asm volatile("ds_read_b128 %0, %5 offset:%6\n v_fma_f32 %1, %1, %7, %8\n v_fma_f32 %2, %2, %7, %8\n v_fma_f32 %3, %3, %7, %8\n v_fma_f32 %4, %4, %7, %8\n v_fma_f32 %1, %1, %7, %8\n v_fma_f32 %2, %2, %7, %8\n v_fma_f32 %3, %3, %7, %8\n v_fma_f32 %4, %4, %7, %8\n"
                   : "=v"(v), "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(idx * 16), "i"(u * 16), "v"(b), "v"(c));
    }
    asm volatile("s_waitcnt lgkmcnt(0)");
  }
  out[blockIdx.x * 256 + threadIdx.x] = a0 + a1 + a2 + a3 + v.x;
}

typedef void (*kern_t)(float*, int, float);

int main(int argc, char** argv) {
  int waves_per_simd = argc > 1 ? atoi(argv[1]) : 8;
  hipDeviceProp_t prop;
  hipGetDeviceProperties(&prop, 0);
  const int cus = prop.multiProcessorCount;
  const double clk_khz = prop.clockRate;
  printf("device %s, %d CUs, clock %.0f MHz, waves/SIMD %d\n", prop.name, cus, clk_khz / 1e3, waves_per_simd);
  const int blocks = cus * waves_per_simd;   // 256 threads = 4 waves = 1 per SIMD per block
  float* out;
  hipMalloc(&out, (size_t)blocks * 256 * sizeof(float));
  struct T { const char* name; kern_t k; int per_trip; };
  std::vector<T> tests = {
      {"v_fma_f32", k_fma, 64}, {"v_mul_f32", k_mul, 64}, {"v_add_f32", k_add, 64}, {"v_pk_fma_f32", k_pkfma, 64},
      {"v_pk_mul_f32", k_pkmul, 64}, {"v_rcp_f32", k_rcp, 64}, {"v_sqrt_f32", k_sqrt, 64}, {"v_readlane_b32", k_readlane, 64},
      {"readlane+mul(sgpr) pairs", k_readlane_use, 64}, {"v_cmp_lt_f32 vcc", k_cmp, 64}, {"v_cmp_lt_f32 sgpr", k_cmp_sgpr, 64},
      {"v_cndmask_b32", k_cndmask, 64}, {"v_div_scale_f32", k_divscale, 64}, {"v_div_fmas_f32", k_divfmas, 64},
      {"v_div_fixup_f32", k_divfixup, 64}, {"v_xor_b32", k_xor, 64}, {"v_cvt_f32_u32", k_cvt, 64},
      {"fma + s_add interleaved (count fma)", k_fma_salu, 32}, {"s_add_u32 only", k_salu, 64},
      {"v_fma_f32 dependent chain", k_fma_dep, 64}, {"v_fma_f32 sgpr operand", k_fma_sgpr, 64},
      {"v_cndmask_b32 sgpr-pair mask", k_cndmask_s, 64}, {"v_cmp + v_cndmask pairs (per inst)", k_cmp_cnd, 64},
      {"v_max_f32", k_max, 64}, {"v_bfi_b32", k_bfi, 64}, {"v_mov_b32", k_mov, 64}, {"v_mov_b32 dpp row_shr", k_dpp, 64},
      {"v_mul_f32 literal", k_mul_lit, 64}, {"v_mul_f32 inline const", k_mul_inl, 64}, {"v_fma_f32 neg modifier", k_fma_neg, 64},
      {"v_sub_f32_e64 abs modifier", k_sub_e64, 64}, {"v_readfirstlane_b32", k_readfirst, 64}, {"v_cmp_lt_u32", k_cmp_u32, 64},
      {"v_cmpx + s_mov exec (count cmpx)", k_cmpx, 32},
      {"ds_read_b128 broadcast", k_lds_b128, 16}, {"ds_read_b128 + 8 fma (count groups of 9)", k_lds_b128_fma, 8}};
  const int iters = 4000; setvbuf(stdout, NULL, _IONBF, 0);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  for (auto& t : tests) {
    hipLaunchKernelGGL(t.k, dim3(blocks), dim3(256), 0, 0, out, 100, 1.0f);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(t.k, dim3(blocks), dim3(256), 0, 0, out, iters, 1.0f);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    // instructions per SIMD = waves_per_simd * iters * per_trip ; cycles = ms * clk
    const double insts = (double)waves_per_simd * iters * t.per_trip;
    const double cyc = ms * 1e-3 * clk_khz * 1e3;
    printf("%-40s %8.3f ms  %6.2f cycles/inst/SIMD (at nominal clock)\n", t.name, ms, cyc / insts);
  }
  return 0;
}
