// Measures the chip's sustained VALU issue rate (wave-instructions per second) per instruction
// kind, to price the k_sw_band kernels against.  tools/, not product.
//   hipcc --offload-arch=gfx950 -O3 -Wno-unused-value tools/valu_peak.hip -o /tmp/valu_peak && /tmp/valu_peak
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

#define REP16(X) X X X X X X X X X X X X X X X X
// OP(d, s) expands to one instruction writing accumulator d, reading accumulator s and the
// loop-invariant registers %8 %9 (and the SGPR pair %10 where a mask is needed).
#define DEFK(NAME, OP)                                                                                  \
  __global__ void __launch_bounds__(256) NAME(uint32_t *out, int iters) {                               \
    uint32_t a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5,         \
             a6 = a0 + 6, a7 = a0 + 7;                                                                  \
    uint32_t b = blockIdx.x | 1, c = 12345;                                                             \
    for (int i = 0; i < iters; ++i)                                                                     \
      asm volatile(REP16(OP("%0", "%1") OP("%1", "%2") OP("%2", "%3") OP("%3", "%4") OP("%4", "%5")     \
                             OP("%5", "%6") OP("%6", "%7") OP("%7", "%0"))                              \
                   : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)     \
                   : "v"(b), "v"(c) : "vcc", "s20", "s21", "s22");                                                           \
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7;                 \
  }

#define I2(M)      M " " 
#define OP_ADD(d, s)     "v_add_u32 " d ", " s ", %8\n"
#define OP_SUB(d, s)     "v_sub_u32 " d ", " s ", %8\n"
#define OP_AND(d, s)     "v_and_b32 " d ", " s ", %8\n"
#define OP_OR(d, s)      "v_or_b32 " d ", " s ", %8\n"
#define OP_XOR(d, s)     "v_xor_b32 " d ", " s ", %8\n"
#define OP_MOV(d, s)     "v_mov_b32 " d ", " s "\n"
#define OP_LSHL(d, s)    "v_lshlrev_b32 " d ", 3, " s "\n"
#define OP_LSHR(d, s)    "v_lshrrev_b32 " d ", 3, " s "\n"
#define OP_ASHR(d, s)    "v_ashrrev_i32 " d ", 3, " s "\n"
#define OP_MAXI(d, s)    "v_max_i32 " d ", " s ", %8\n"
#define OP_MAXU(d, s)    "v_max_u32 " d ", " s ", %8\n"
#define OP_MINI(d, s)    "v_min_i32 " d ", " s ", %8\n"
#define OP_MAXF(d, s)    "v_max_f32 " d ", " s ", %8\n"
#define OP_ADDF(d, s)    "v_add_f32 " d ", " s ", %8\n"
#define OP_FMA(d, s)     "v_fma_f32 " d ", " s ", %8, %9\n"
#define OP_MAX3(d, s)    "v_max3_i32 " d ", " s ", %8, %9\n"
#define OP_MAX3F(d, s)   "v_max3_f32 " d ", " s ", %8, %9\n"
#define OP_MED3(d, s)    "v_med3_i32 " d ", " s ", %8, %9\n"
#define OP_ADD3(d, s)    "v_add3_u32 " d ", " s ", %8, %9\n"
#define OP_LSHLADD(d, s) "v_lshl_add_u32 " d ", " s ", 2, %9\n"
#define OP_ANDOR(d, s)   "v_and_or_b32 " d ", " s ", %8, %9\n"
#define OP_BFE(d, s)     "v_bfe_u32 " d ", " s ", 3, 5\n"
#define OP_PERM(d, s)    "v_perm_b32 " d ", " s ", %8, %9\n"
#define OP_CMP(d, s)     "v_cmp_gt_i32 vcc, " s ", %8\n"
#define OP_CNDMASK(d, s) "v_cndmask_b32 " d ", " s ", %8, vcc\n"
#define OP_MAXI16(d, s)  "v_max_i16 " d ", " s ", %8\n"
#define OP_MAXU16(d, s)  "v_max_u16 " d ", " s ", %8\n"
#define OP_ADDU16(d, s)  "v_add_u16 " d ", " s ", %8\n"
#define OP_PKMAXI16(d, s) "v_pk_max_i16 " d ", " s ", %8\n"
#define OP_PKADDU16(d, s) "v_pk_add_u16 " d ", " s ", %8\n"
#define OP_PKMAXF16(d, s) "v_pk_max_f16 " d ", " s ", %8\n"
#define OP_MAXF16(d, s)  "v_max_f16 " d ", " s ", %8\n"
#define OP_MOVDPP(d, s)  "v_mov_b32_dpp " d ", " s " row_shr:1 row_mask:0xf bank_mask:0xf\n"
#define OP_ADDDPP(d, s)  "v_add_u32_dpp " d ", " s ", %8 row_shr:1 row_mask:0xf bank_mask:0xf\n"
#define OP_MAXDPP(d, s)  "v_max_i32_dpp " d ", " s ", %8 row_shr:1 row_mask:0xf bank_mask:0xf\n"
#define OP_ADDSDWA(d, s) "v_add_u32_sdwa " d ", " s ", %8 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_0 src1_sel:DWORD\n"
#define OP_SAD(d, s)     "v_sad_u8 " d ", " s ", %8, %9\n"
#define OP_MAD24(d, s)   "v_mad_u32_u24 " d ", " s ", %8, %9\n"
#define OP_MULLO(d, s)   "v_mul_lo_u32 " d ", " s ", %8\n"
#define OP_ADDCO(d, s)   "v_add_co_u32 " d ", vcc, " s ", %8\n"
#define OP_XAD(d, s)     "v_xad_u32 " d ", " s ", %8, %9\n"
#define OP_BFI(d, s)     "v_bfi_b32 " d ", " s ", %8, %9\n"
#define OP_ALIGNBIT(d, s) "v_alignbit_b32 " d ", " s ", %8, 8\n"
// compare + two selects, the running-best update of the SW sweep: through VCC (32-bit encodings)
// and through an SGPR pair (64-bit encodings); 3 instructions per OP
#define OP_CMPCND32(d, s) "v_cmp_gt_i32 vcc, " s ", %8\n v_cndmask_b32 " d ", " d ", " s ", vcc\n v_cndmask_b32 %9, %9, " s ", vcc\n"
#define OP_CMPCND64(d, s) "v_cmp_gt_i32 s[20:21], " s ", %8\n v_cndmask_b32 " d ", " d ", " s ", s[20:21]\n v_cndmask_b32 %9, %9, " s ", s[20:21]\n"
#define OP_CND32(d, s)    "v_cndmask_b32 " d ", " d ", " s ", vcc\n"
#define OP_CND64(d, s)    "v_cndmask_b32 " d ", " d ", " s ", s[20:21]\n"
#define OP_SUBREV(d, s)   "v_subrev_u32 " d ", %8, " s "\n"
#define OP_ADDS(d, s)     "v_add_u32 " d ", s22, " s "\n"
#define OP_ADDLIT(d, s)   "v_add_u32 " d ", 0x3ff, " s "\n"
#define OP_ORLIT(d, s)    "v_or_b32 " d ", 0x3ffff, " s "\n"
#define OP_MAXLIT(d, s)   "v_max_i32 " d ", 0x3ffff, " s "\n"

#define KINDS(X) X(ADD) X(SUB) X(AND) X(OR) X(XOR) X(MOV) X(LSHL) X(LSHR) X(ASHR) X(MAXI) X(MAXU) X(MINI) X(MAXF) X(ADDF) X(FMA) \
  X(MAX3) X(MAX3F) X(MED3) X(ADD3) X(LSHLADD) X(ANDOR) X(BFE) X(PERM) X(CMP) X(CNDMASK) X(MAXI16) X(MAXU16) X(ADDU16) X(PKMAXI16) \
  X(PKADDU16) X(PKMAXF16) X(MAXF16) X(MOVDPP) X(ADDDPP) X(MAXDPP) X(ADDSDWA) X(SAD) X(MAD24) X(MULLO) X(ADDCO) X(XAD) X(BFI) X(ALIGNBIT) X(CMPCND32) X(CMPCND64) X(CND32) X(CND64) X(SUBREV) X(ADDS) X(ADDLIT) X(ORLIT) X(MAXLIT)
#define MK(N) DEFK(k_##N, OP_##N)
KINDS(MK)

// 64-bit maximum through the f64 pipe (bit patterns of non-negative, non-NaN doubles order like integers)
__global__ void __launch_bounds__(256) k_MAXF64(uint32_t *out, int iters) {
  double a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, b = blockIdx.x + 0.5;
  for (int i = 0; i < iters; ++i)
    asm volatile(REP16("v_max_f64 %0, %1, %4\n v_max_f64 %1, %2, %4\n v_max_f64 %2, %3, %4\n v_max_f64 %3, %0, %4\n"
                       "v_max_f64 %0, %1, %4\n v_max_f64 %1, %2, %4\n v_max_f64 %2, %3, %4\n v_max_f64 %3, %0, %4\n")
                 : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b));
  out[blockIdx.x * blockDim.x + threadIdx.x] = (uint32_t)(a0 + a1 + a2 + a3);
}
__global__ void __launch_bounds__(256) k_CMPU64(uint32_t *out, int iters) {
  double a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, b = blockIdx.x + 0.5;
  for (int i = 0; i < iters; ++i)
    asm volatile(REP16("v_cmp_gt_u64 s[20:21], %1, %4\n v_cmp_gt_u64 s[20:21], %2, %4\n v_cmp_gt_u64 s[20:21], %3, %4\n v_cmp_gt_u64 s[20:21], %0, %4\n"
                       "v_cmp_gt_u64 s[20:21], %1, %4\n v_cmp_gt_u64 s[20:21], %2, %4\n v_cmp_gt_u64 s[20:21], %3, %4\n v_cmp_gt_u64 s[20:21], %0, %4\n")
                 : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b) : "s20", "s21");
  out[blockIdx.x * blockDim.x + threadIdx.x] = (uint32_t)(a0 + a1 + a2 + a3);
}

typedef void (*kern_t)(uint32_t *, int);
static void run(const char *name, kern_t k, int waves_per_simd, uint32_t *out) {
  const int iters = 1000, blocks = 256 * waves_per_simd;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(k, blocks, 256, 0, 0, out, iters);
  hipEventRecord(e0);
  for (int r = 0; r < 3; ++r) hipLaunchKernelGGL(k, blocks, 256, 0, 0, out, iters);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 3;
  double instr = double(blocks) * 4 /*waves*/ * iters * 128.0;
  printf("%-10s %d waves/SIMD  %8.1f G wave-instr/s\n", name, waves_per_simd, instr / ms / 1e6);
}

int main() {
  uint32_t *out; hipMalloc(&out, 256 * 8 * 256 * 4);
  for (int w : {4, 1}) {
#define RUN(N) run(#N, k_##N, w, out);
    KINDS(RUN)
    RUN(MAXF64) RUN(CMPU64)
  }
  return 0;
}
