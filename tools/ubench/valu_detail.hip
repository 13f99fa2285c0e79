// Micro-benchmark: why does the pair kernel's VALU stream issue ~45 % slower than valu_rates.hip's independent
// streams?  Suspects measured here, each as a loop of 64 wave-instructions on PHYSICAL registers chosen by hand
// (gfx950 VGPR banks = register index mod 4):
//   fma3 with its three sources in three banks / two in one bank / all in one bank, mul2 likewise, v_fmac (dst = third
//   source), literal-carrying VOP2 (8-byte encodings), v_mov_b64, SDWA, v_rsq with a dependent / independent successor,
//   v_cmp -> SGPR pair + v_cndmask, and 8-byte vs 4-byte encodings of the same operation (instruction fetch).
//   hipcc --offload-arch=gfx950 -O3 valu_detail.hip -o valu_detail && ./valu_detail
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
constexpr int LOOPS = 2000;

#define CLOB "v10", "v11", "v12", "v13", "v14", "v15", "v16", "v17", "v18", "v19", "v20", "v21", "v22", "v23", "v24", "v25", \
             "v26", "v27", "v28", "v29", "v30", "v31", "v32", "v33", "v34", "v35", "v36", "v37", "v38", "v39", "v40", "v41", "vcc", \
             "s20", "s21", "s22", "s23", "s24", "s25", "s26", "s27"
#define INIT "v_mov_b32 v10, 1.0\n v_mov_b32 v11, 1.0\n v_mov_b32 v12, 1.0\n v_mov_b32 v13, 1.0\n v_mov_b32 v14, 1.0\n v_mov_b32 v15, 1.0\n" \
             "v_mov_b32 v16, 1.0\n v_mov_b32 v17, 1.0\n v_mov_b32 v18, 0.5\n v_mov_b32 v19, 0.5\n v_mov_b32 v20, 0.5\n v_mov_b32 v21, 0.5\n" \
             "v_mov_b32 v22, 0.5\n v_mov_b32 v23, 0.5\n v_mov_b32 v24, 0.5\n v_mov_b32 v25, 0.5\n v_mov_b32 v26, 2.0\n v_mov_b32 v27, 2.0\n" \
             "v_mov_b32 v28, 2.0\n v_mov_b32 v29, 2.0\n v_mov_b32 v30, 2.0\n v_mov_b32 v31, 2.0\n v_mov_b32 v32, 2.0\n v_mov_b32 v33, 2.0\n" \
             "v_mov_b32 v34, 1.0\n v_mov_b32 v35, 1.0\n v_mov_b32 v36, 1.0\n v_mov_b32 v37, 1.0\n v_mov_b32 v38, 1.0\n v_mov_b32 v39, 1.0\n v_mov_b32 v40, 1.0\n v_mov_b32 v41, 1.0\n"

// BODY = 8 instructions; repeated 8 times per loop iteration = 64 wave-instructions
#define KERNEL(NAME, BODY)                                                    \
  __global__ void NAME(float *out) {                                          \
    asm volatile(INIT ::: CLOB);                                              \
    for (int i = 0; i < LOOPS; ++i) {                                         \
      asm volatile(BODY BODY BODY BODY BODY BODY BODY BODY ::: CLOB);         \
    }                                                                         \
    float r;                                                                  \
    asm volatile("v_add_f32 %0, v10, v11" : "=v"(r)::CLOB);                  \
    out[blockIdx.x * blockDim.x + threadIdx.x] = r;                           \
  }

// dst rotates over v10..v17 (banks 2,3,0,1,...); sources fixed
// three sources in three different banks: v18 (2), v19 (3), v20 (0)
KERNEL(k_fma_3banks, "v_fma_f32 v10, v18, v19, v20\n v_fma_f32 v11, v18, v19, v20\n v_fma_f32 v12, v18, v19, v20\n v_fma_f32 v13, v18, v19, v20\n"
                     "v_fma_f32 v14, v18, v19, v20\n v_fma_f32 v15, v18, v19, v20\n v_fma_f32 v16, v18, v19, v20\n v_fma_f32 v17, v18, v19, v20\n")
// two sources in one bank: v18 (2), v22 (2), v19 (3)
KERNEL(k_fma_2same, "v_fma_f32 v10, v18, v22, v19\n v_fma_f32 v11, v18, v22, v19\n v_fma_f32 v12, v18, v22, v19\n v_fma_f32 v13, v18, v22, v19\n"
                    "v_fma_f32 v14, v18, v22, v19\n v_fma_f32 v15, v18, v22, v19\n v_fma_f32 v16, v18, v22, v19\n v_fma_f32 v17, v18, v22, v19\n")
// all three in one bank: v18, v22, v26 (bank 2)
KERNEL(k_fma_3same, "v_fma_f32 v10, v18, v22, v26\n v_fma_f32 v11, v18, v22, v26\n v_fma_f32 v12, v18, v22, v26\n v_fma_f32 v13, v18, v22, v26\n"
                    "v_fma_f32 v14, v18, v22, v26\n v_fma_f32 v15, v18, v22, v26\n v_fma_f32 v16, v18, v22, v26\n v_fma_f32 v17, v18, v22, v26\n")
// varying sources (as real code has): each instruction reads three different registers, different from its neighbours'
KERNEL(k_fma_vary, "v_fma_f32 v10, v18, v27, v36\n v_fma_f32 v11, v19, v28, v37\n v_fma_f32 v12, v20, v29, v38\n v_fma_f32 v13, v21, v30, v39\n"
                   "v_fma_f32 v14, v22, v31, v40\n v_fma_f32 v15, v23, v32, v41\n v_fma_f32 v16, v24, v33, v34\n v_fma_f32 v17, v25, v26, v35\n")
KERNEL(k_fma_vary_same, "v_fma_f32 v10, v18, v26, v34\n v_fma_f32 v11, v19, v27, v35\n v_fma_f32 v12, v20, v28, v36\n v_fma_f32 v13, v21, v29, v37\n"
                        "v_fma_f32 v14, v22, v30, v38\n v_fma_f32 v15, v23, v31, v39\n v_fma_f32 v16, v24, v32, v40\n v_fma_f32 v17, v25, v33, v41\n")
// fma with one SGPR source (as the minimum image has)
KERNEL(k_fma_sgpr, "v_fma_f32 v10, s20, v19, v20\n v_fma_f32 v11, s20, v19, v20\n v_fma_f32 v12, s20, v19, v20\n v_fma_f32 v13, s20, v19, v20\n"
                   "v_fma_f32 v14, s20, v19, v20\n v_fma_f32 v15, s20, v19, v20\n v_fma_f32 v16, s20, v19, v20\n v_fma_f32 v17, s20, v19, v20\n")
KERNEL(k_mul_2banks, "v_mul_f32 v10, v18, v19\n v_mul_f32 v11, v18, v19\n v_mul_f32 v12, v18, v19\n v_mul_f32 v13, v18, v19\n"
                     "v_mul_f32 v14, v18, v19\n v_mul_f32 v15, v18, v19\n v_mul_f32 v16, v18, v19\n v_mul_f32 v17, v18, v19\n")
KERNEL(k_mul_same, "v_mul_f32 v10, v18, v22\n v_mul_f32 v11, v18, v22\n v_mul_f32 v12, v18, v22\n v_mul_f32 v13, v18, v22\n"
                   "v_mul_f32 v14, v18, v22\n v_mul_f32 v15, v18, v22\n v_mul_f32 v16, v18, v22\n v_mul_f32 v17, v18, v22\n")
// v_fmac: dst is the third source (2-byte shorter encoding than v_fma)
KERNEL(k_fmac, "v_fmac_f32 v10, v18, v19\n v_fmac_f32 v11, v18, v19\n v_fmac_f32 v12, v18, v19\n v_fmac_f32 v13, v18, v19\n"
               "v_fmac_f32 v14, v18, v19\n v_fmac_f32 v15, v18, v19\n v_fmac_f32 v16, v18, v19\n v_fmac_f32 v17, v18, v19\n")
// 8-byte VOP2 with a 32-bit literal
KERNEL(k_add_literal, "v_add_f32 v10, 0xcb400000, v18\n v_add_f32 v11, 0xcb400000, v19\n v_add_f32 v12, 0xcb400000, v20\n v_add_f32 v13, 0xcb400000, v21\n"
                      "v_add_f32 v14, 0xcb400000, v22\n v_add_f32 v15, 0xcb400000, v23\n v_add_f32 v16, 0xcb400000, v24\n v_add_f32 v17, 0xcb400000, v25\n")
KERNEL(k_add_inline, "v_add_f32 v10, 1.0, v18\n v_add_f32 v11, 1.0, v19\n v_add_f32 v12, 1.0, v20\n v_add_f32 v13, 1.0, v21\n"
                     "v_add_f32 v14, 1.0, v22\n v_add_f32 v15, 1.0, v23\n v_add_f32 v16, 1.0, v24\n v_add_f32 v17, 1.0, v25\n")
KERNEL(k_mov_b64, "v_mov_b64 v[10:11], v[18:19]\n v_mov_b64 v[12:13], v[20:21]\n v_mov_b64 v[14:15], v[22:23]\n v_mov_b64 v[16:17], v[24:25]\n"
                  "v_mov_b64 v[10:11], v[26:27]\n v_mov_b64 v[12:13], v[28:29]\n v_mov_b64 v[14:15], v[30:31]\n v_mov_b64 v[16:17], v[32:33]\n")
KERNEL(k_sdwa, "v_add_u32_sdwa v10, v18, v19 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_3\n v_add_u32_sdwa v11, v18, v19 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_3\n"
               "v_add_u32_sdwa v12, v18, v19 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_3\n v_add_u32_sdwa v13, v18, v19 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_3\n"
               "v_add_u32_sdwa v14, v18, v19 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_3\n v_add_u32_sdwa v15, v18, v19 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_3\n"
               "v_add_u32_sdwa v16, v18, v19 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_3\n v_add_u32_sdwa v17, v18, v19 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_3\n")
// rsq followed by an instruction that needs its result / by independent work (1 rsq + 7 plain per 8)
KERNEL(k_rsq_dep, "v_rsq_f32 v10, v18\n v_mul_f32 v11, v10, v10\n v_mul_f32 v12, v11, v11\n v_mul_f32 v13, v12, v11\n"
                  "v_mul_f32 v14, v13, v19\n v_mul_f32 v15, v14, v19\n v_mul_f32 v16, v15, v19\n v_mul_f32 v17, v16, v19\n")
KERNEL(k_rsq_indep, "v_rsq_f32 v10, v18\n v_mul_f32 v11, v20, v20\n v_mul_f32 v12, v21, v21\n v_mul_f32 v13, v22, v21\n"
                    "v_mul_f32 v14, v23, v19\n v_mul_f32 v15, v24, v19\n v_mul_f32 v16, v25, v19\n v_mul_f32 v17, v26, v19\n")
KERNEL(k_mul_dep, "v_mul_f32 v10, v18, v18\n v_mul_f32 v11, v10, v10\n v_mul_f32 v12, v11, v11\n v_mul_f32 v13, v12, v11\n"
                  "v_mul_f32 v14, v13, v19\n v_mul_f32 v15, v14, v19\n v_mul_f32 v16, v15, v19\n v_mul_f32 v17, v16, v19\n")
// 4 rsq back to back then their consumers (the pair kernel's shape: 4 entries per group)
KERNEL(k_rsq4, "v_rsq_f32 v10, v18\n v_rsq_f32 v11, v19\n v_rsq_f32 v12, v20\n v_rsq_f32 v13, v21\n"
               "v_mul_f32 v14, v10, v10\n v_mul_f32 v15, v11, v11\n v_mul_f32 v16, v12, v12\n v_mul_f32 v17, v13, v13\n")
KERNEL(k_cmp_cnd_sgpr, "v_cmp_ge_f32_e64 s[20:21], s24, v18\n v_cndmask_b32_e64 v10, 0, v19, s[20:21]\n v_cmp_ge_f32_e64 s[22:23], s24, v20\n v_cndmask_b32_e64 v11, 0, v21, s[22:23]\n"
                       "v_cmp_ge_f32_e64 s[20:21], s24, v22\n v_cndmask_b32_e64 v12, 0, v23, s[20:21]\n v_cmp_ge_f32_e64 s[22:23], s24, v24\n v_cndmask_b32_e64 v13, 0, v25, s[22:23]\n")
KERNEL(k_fma_clamp_mul, "v_fma_f32 v10, v18, v19, s24 clamp\n v_mul_f32 v11, v10, v20\n v_fma_f32 v12, v21, v19, s24 clamp\n v_mul_f32 v13, v12, v22\n"
                        "v_fma_f32 v14, v23, v19, s24 clamp\n v_mul_f32 v15, v14, v24\n v_fma_f32 v16, v25, v19, s24 clamp\n v_mul_f32 v17, v16, v26\n")
KERNEL(k_sub, "v_sub_f32 v10, v18, v19\n v_sub_f32 v11, v18, v20\n v_sub_f32 v12, v18, v21\n v_sub_f32 v13, v18, v22\n"
              "v_sub_f32 v14, v18, v23\n v_sub_f32 v15, v18, v24\n v_sub_f32 v16, v18, v25\n v_sub_f32 v17, v18, v26\n")
// same operation in the 8-byte VOP3 encoding (instruction fetch: 64 x 8 B per iteration instead of 64 x 4 B)
KERNEL(k_mul_e64, "v_mul_f32_e64 v10, v18, v19\n v_mul_f32_e64 v11, v18, v19\n v_mul_f32_e64 v12, v18, v19\n v_mul_f32_e64 v13, v18, v19\n"
                  "v_mul_f32_e64 v14, v18, v19\n v_mul_f32_e64 v15, v18, v19\n v_mul_f32_e64 v16, v18, v19\n v_mul_f32_e64 v17, v18, v19\n")


// ---- second batch: what exactly makes an SGPR operand / a transcendental expensive -----------------------
KERNEL(k_mul_sgpr_e32, "v_mul_f32 v10, s20, v18\n v_mul_f32 v11, s20, v19\n v_mul_f32 v12, s20, v20\n v_mul_f32 v13, s20, v21\n"
                       "v_mul_f32 v14, s20, v22\n v_mul_f32 v15, s20, v23\n v_mul_f32 v16, s20, v24\n v_mul_f32 v17, s20, v25\n")
KERNEL(k_sub_sgpr_e32, "v_sub_f32 v10, s20, v18\n v_sub_f32 v11, s20, v19\n v_sub_f32 v12, s20, v20\n v_sub_f32 v13, s20, v21\n"
                       "v_sub_f32 v14, s20, v22\n v_sub_f32 v15, s20, v23\n v_sub_f32 v16, s20, v24\n v_sub_f32 v17, s20, v25\n")
KERNEL(k_fma_inline, "v_fma_f32 v10, 2.0, v19, v20\n v_fma_f32 v11, 2.0, v19, v20\n v_fma_f32 v12, 2.0, v19, v20\n v_fma_f32 v13, 2.0, v19, v20\n"
                     "v_fma_f32 v14, 2.0, v19, v20\n v_fma_f32 v15, 2.0, v19, v20\n v_fma_f32 v16, 2.0, v19, v20\n v_fma_f32 v17, 2.0, v19, v20\n")
KERNEL(k_fma_sgpr_last, "v_fma_f32 v10, v18, v19, s20\n v_fma_f32 v11, v18, v19, s20\n v_fma_f32 v12, v18, v19, s20\n v_fma_f32 v13, v18, v19, s20\n"
                        "v_fma_f32 v14, v18, v19, s20\n v_fma_f32 v15, v18, v19, s20\n v_fma_f32 v16, v18, v19, s20\n v_fma_f32 v17, v18, v19, s20\n")
KERNEL(k_fma_2vgpr_same, "v_fma_f32 v10, v18, v18, v20\n v_fma_f32 v11, v19, v19, v20\n v_fma_f32 v12, v21, v21, v20\n v_fma_f32 v13, v22, v22, v20\n"
                         "v_fma_f32 v14, v23, v23, v20\n v_fma_f32 v15, v24, v24, v20\n v_fma_f32 v16, v25, v25, v20\n v_fma_f32 v17, v26, v26, v20\n")
KERNEL(k_fmac_sgpr, "v_fmac_f32 v10, s20, v19\n v_fmac_f32 v11, s20, v19\n v_fmac_f32 v12, s20, v19\n v_fmac_f32 v13, s20, v19\n"
                    "v_fmac_f32 v14, s20, v19\n v_fmac_f32 v15, s20, v19\n v_fmac_f32 v16, s20, v19\n v_fmac_f32 v17, s20, v19\n")
KERNEL(k_fmamk, "v_fmamk_f32 v10, v18, 0x3dcccccd, v19\n v_fmamk_f32 v11, v18, 0x3dcccccd, v19\n v_fmamk_f32 v12, v18, 0x3dcccccd, v19\n v_fmamk_f32 v13, v18, 0x3dcccccd, v19\n"
                "v_fmamk_f32 v14, v18, 0x3dcccccd, v19\n v_fmamk_f32 v15, v18, 0x3dcccccd, v19\n v_fmamk_f32 v16, v18, 0x3dcccccd, v19\n v_fmamk_f32 v17, v18, 0x3dcccccd, v19\n")
KERNEL(k_cmp_e32_vcc, "v_cmp_ge_f32 vcc, v19, v18\n v_cndmask_b32 v10, v20, v21, vcc\n v_cmp_ge_f32 vcc, v22, v18\n v_cndmask_b32 v11, v20, v21, vcc\n"
                      "v_cmp_ge_f32 vcc, v23, v18\n v_cndmask_b32 v12, v20, v21, vcc\n v_cmp_ge_f32 vcc, v24, v18\n v_cndmask_b32 v13, v20, v21, vcc\n")
KERNEL(k_cmp_only_e64, "v_cmp_ge_f32_e64 s[20:21], v19, v18\n v_cmp_ge_f32_e64 s[22:23], v20, v18\n v_cmp_ge_f32_e64 s[20:21], v21, v18\n v_cmp_ge_f32_e64 s[22:23], v22, v18\n"
                       "v_cmp_ge_f32_e64 s[20:21], v23, v18\n v_cmp_ge_f32_e64 s[22:23], v24, v18\n v_cmp_ge_f32_e64 s[20:21], v25, v18\n v_cmp_ge_f32_e64 s[22:23], v26, v18\n")
KERNEL(k_cnd_only_e64, "v_cndmask_b32_e64 v10, 0, v19, s[20:21]\n v_cndmask_b32_e64 v11, 0, v19, s[20:21]\n v_cndmask_b32_e64 v12, 0, v19, s[20:21]\n v_cndmask_b32_e64 v13, 0, v19, s[20:21]\n"
                       "v_cndmask_b32_e64 v14, 0, v19, s[20:21]\n v_cndmask_b32_e64 v15, 0, v19, s[20:21]\n v_cndmask_b32_e64 v16, 0, v19, s[20:21]\n v_cndmask_b32_e64 v17, 0, v19, s[20:21]\n")
KERNEL(k_perm, "v_perm_b32 v10, v18, v19, v20\n v_perm_b32 v11, v18, v19, v20\n v_perm_b32 v12, v18, v19, v20\n v_perm_b32 v13, v18, v19, v20\n"
               "v_perm_b32 v14, v18, v19, v20\n v_perm_b32 v15, v18, v19, v20\n v_perm_b32 v16, v18, v19, v20\n v_perm_b32 v17, v18, v19, v20\n")
KERNEL(k_bfe, "v_bfe_u32 v10, v18, 24, 8\n v_bfe_u32 v11, v19, 24, 8\n v_bfe_u32 v12, v20, 24, 8\n v_bfe_u32 v13, v21, 24, 8\n"
              "v_bfe_u32 v14, v22, 24, 8\n v_bfe_u32 v15, v23, 24, 8\n v_bfe_u32 v16, v24, 24, 8\n v_bfe_u32 v17, v25, 24, 8\n")
KERNEL(k_alignbit, "v_alignbit_b32 v10, v18, v19, 24\n v_alignbit_b32 v11, v18, v19, 24\n v_alignbit_b32 v12, v18, v19, 24\n v_alignbit_b32 v13, v18, v19, 24\n"
                   "v_alignbit_b32 v14, v18, v19, 24\n v_alignbit_b32 v15, v18, v19, 24\n v_alignbit_b32 v16, v18, v19, 24\n v_alignbit_b32 v17, v18, v19, 24\n")
KERNEL(k_and_or, "v_and_or_b32 v10, v18, v19, v20\n v_and_or_b32 v11, v18, v19, v20\n v_and_or_b32 v12, v18, v19, v20\n v_and_or_b32 v13, v18, v19, v20\n"
                 "v_and_or_b32 v14, v18, v19, v20\n v_and_or_b32 v15, v18, v19, v20\n v_and_or_b32 v16, v18, v19, v20\n v_and_or_b32 v17, v18, v19, v20\n")
KERNEL(k_lshl_or, "v_lshl_or_b32 v10, v18, 3, v20\n v_lshl_or_b32 v11, v18, 3, v20\n v_lshl_or_b32 v12, v18, 3, v20\n v_lshl_or_b32 v13, v18, 3, v20\n"
                  "v_lshl_or_b32 v14, v18, 3, v20\n v_lshl_or_b32 v15, v18, 3, v20\n v_lshl_or_b32 v16, v18, 3, v20\n v_lshl_or_b32 v17, v18, 3, v20\n")
KERNEL(k_and_literal, "v_and_b32 v10, 0x7fffff0, v18\n v_and_b32 v11, 0x7fffff0, v19\n v_and_b32 v12, 0x7fffff0, v20\n v_and_b32 v13, 0x7fffff0, v21\n"
                      "v_and_b32 v14, 0x7fffff0, v22\n v_and_b32 v15, 0x7fffff0, v23\n v_and_b32 v16, 0x7fffff0, v24\n v_and_b32 v17, 0x7fffff0, v25\n")
// transcendental placement: alternating with plain ops / in pairs / one per 8 / one per 16 (two bodies differ)
KERNEL(k_rsq_alt, "v_rsq_f32 v10, v18\n v_mul_f32 v11, v20, v20\n v_rsq_f32 v12, v19\n v_mul_f32 v13, v22, v21\n"
                  "v_rsq_f32 v14, v23\n v_mul_f32 v15, v24, v19\n v_rsq_f32 v16, v25\n v_mul_f32 v17, v26, v19\n")
KERNEL(k_rsq2_6, "v_rsq_f32 v10, v18\n v_rsq_f32 v11, v19\n v_mul_f32 v12, v21, v21\n v_mul_f32 v13, v22, v21\n"
                 "v_mul_f32 v14, v23, v19\n v_mul_f32 v15, v24, v19\n v_mul_f32 v16, v25, v19\n v_mul_f32 v17, v26, v19\n")
KERNEL(k_rsq4_fma4, "v_rsq_f32 v10, v18\n v_rsq_f32 v11, v19\n v_rsq_f32 v12, v20\n v_rsq_f32 v13, v21\n"
                    "v_fma_f32 v14, v22, v23, v24\n v_fma_f32 v15, v22, v23, v24\n v_fma_f32 v16, v22, v23, v24\n v_fma_f32 v17, v22, v23, v24\n")
KERNEL(k_rsq_only, "v_rsq_f32 v10, v18\n v_rsq_f32 v11, v19\n v_rsq_f32 v12, v20\n v_rsq_f32 v13, v21\n"
                   "v_rsq_f32 v14, v22\n v_rsq_f32 v15, v23\n v_rsq_f32 v16, v24\n v_rsq_f32 v17, v25\n")

struct Case { const char *name; void (*fn)(float *); };

int main() {
  hipDeviceProp_t prop;
  CHECK(hipGetDeviceProperties(&prop, 0));
  const int cus = prop.multiProcessorCount;
  float *out;
  CHECK(hipMalloc(&out, sizeof(float) * cus * 8 * 256));
  std::vector<Case> cases = {
      {"fma: 3 sources, 3 banks", k_fma_3banks}, {"fma: two sources in one bank", k_fma_2same}, {"fma: three sources in one bank", k_fma_3same},
      {"fma: varying regs, mixed banks", k_fma_vary}, {"fma: varying regs, all one bank", k_fma_vary_same}, {"fma: SGPR + 2 VGPR", k_fma_sgpr},
      {"mul: 2 banks", k_mul_2banks}, {"mul: same bank", k_mul_same}, {"mul e64 (8-byte encoding)", k_mul_e64}, {"fmac", k_fmac},
      {"add with 32-bit literal", k_add_literal}, {"add with inline constant", k_add_inline}, {"sub", k_sub}, {"v_mov_b64", k_mov_b64},
      {"v_add_u32_sdwa", k_sdwa}, {"rsq + 7 dependent mul", k_rsq_dep}, {"rsq + 7 independent mul", k_rsq_indep},
      {"8 dependent mul (no rsq)", k_mul_dep}, {"4 rsq + 4 consumers", k_rsq4}, {"cmp->sgpr + cndmask", k_cmp_cnd_sgpr},
      {"fma clamp + mul", k_fma_clamp_mul},
      {"mul e32, SGPR src0", k_mul_sgpr_e32}, {"sub e32, SGPR src0", k_sub_sgpr_e32}, {"fmac, SGPR src0", k_fmac_sgpr},
      {"fma, inline constant src0", k_fma_inline}, {"fma, SGPR as src2", k_fma_sgpr_last}, {"fma v,a,a,c (2 distinct VGPR)", k_fma_2vgpr_same},
      {"v_fmamk (literal)", k_fmamk}, {"cmp e32 -> vcc + cndmask e32", k_cmp_e32_vcc}, {"cmp e64 only (-> SGPR pair)", k_cmp_only_e64},
      {"cndmask e64 only (SGPR mask)", k_cnd_only_e64}, {"v_perm_b32", k_perm}, {"v_bfe_u32", k_bfe}, {"v_alignbit_b32", k_alignbit},
      {"v_and_or_b32", k_and_or}, {"v_lshl_or_b32", k_lshl_or}, {"v_and_b32 literal", k_and_literal},
      {"rsq, mul alternating (4+4)", k_rsq_alt}, {"2 rsq + 6 mul", k_rsq2_6}, {"4 rsq + 4 fma", k_rsq4_fma4}, {"8 rsq", k_rsq_only},
  };
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0));
  CHECK(hipEventCreate(&e1));
  for (int wps : {2, 6}) {
    printf("--- %d wave(s) per SIMD ---\n", wps);
    for (auto &c : cases) {
      const int blocks = cus * wps;
      hipLaunchKernelGGL(c.fn, dim3(blocks), dim3(256), 0, 0, out);
      CHECK(hipDeviceSynchronize());
      float best = 1e30f;
      for (int rep = 0; rep < 5; ++rep) {
        CHECK(hipEventRecord(e0, 0));
        hipLaunchKernelGGL(c.fn, dim3(blocks), dim3(256), 0, 0, out);
        CHECK(hipEventRecord(e1, 0));
        CHECK(hipEventSynchronize(e1));
        float ms;
        CHECK(hipEventElapsedTime(&ms, e0, e1));
        best = ms < best ? ms : best;
      }
      const double instr_per_simd = (double)LOOPS * 64 * wps;
      const double ns = best * 1e6 / instr_per_simd;
      printf("%-34s %8.3f ms  %6.3f ns / wave-instr / SIMD = %5.2f cycles @2.4GHz\n", c.name, best, ns, ns * 2.4);
    }
  }
  return 0;
}
