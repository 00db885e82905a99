// vgpr_banks.hip — does a lone wave's fp64 instruction cost depend on WHICH registers its operands sit in?
// 64 v_fma_f64 per pass (loop aligned to 8 bytes: every instruction is an 8-byte VOP3), destination and the three sources in
// hard-coded registers. Variants: all sources in registers = 0 mod 4, spread over the four residues, dst = a source, ...
// Build: hipcc -O3 --offload-arch=gfx950 -o vgpr_banks vgpr_banks.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#define U4(x) x x x x
#define U16(x) U4(x) U4(x) U4(x) U4(x)
#define U64(x) U16(x) U16(x) U16(x) U16(x)

#define KERNEL(NAME, BODY)                                                                                   \
    __global__ void NAME(double* out, unsigned long long* cyc, int rep) {                                    \
        asm volatile("v_mov_b64 v[20:21], 1.0\n v_mov_b64 v[22:23], 1.0\n v_mov_b64 v[24:25], 0.5\n v_mov_b64 v[26:27], 0.5\n" \
                     "v_mov_b64 v[28:29], 1.0\n v_mov_b64 v[30:31], 0.5\n v_mov_b64 v[32:33], 1.0\n v_mov_b64 v[34:35], 0.5\n" \
                     "v_mov_b64 v[36:37], 1.0\n v_mov_b64 v[38:39], 0.5\n v_mov_b64 v[40:41], 0.5\n v_mov_b64 v[42:43], 0.5\n" \
                     "v_mov_b64 v[44:45], 0.5\n v_mov_b64 v[46:47], 0.5\n" ::: "v20","v21","v22","v23","v24","v25","v26","v27","v28","v29","v30","v31","v32","v33","v34","v35","v36","v37","v38","v39","v40","v41","v42","v43","v44","v45","v46","v47"); \
        __shared__ double lds_buf[1024];                                                                    \
        lds_buf[threadIdx.x] = 1.0;                                                                          \
        asm volatile("v_mov_b32 v36, 0\n v_lshlrev_b32 v38, 2, %0\n v_mov_b32 v37, 0\n s_mov_b64 s[24:25], %1\n s_mov_b64 s[22:23], 0" : : "v"(threadIdx.x), "s"(out + 64) : "v36", "v37", "v38", "s24", "s25", "s22", "s23"); \
        const unsigned long long t0 = __builtin_amdgcn_s_memtime();                                          \
        for (int r = 0; r < rep; ++r) asm volatile(".p2align 3\n" U64(BODY) ::: "v20","v21","v22","v23","v24","v25","v26","v27","v28","v29","v30","v31","v32","v33","v34","v35","v36","v37","v38","v39","v40","v41","v42","v43","v44","v45","v46","v47","s20","s21","s22","s23","vcc","memory"); \
        const unsigned long long t1 = __builtin_amdgcn_s_memtime();                                          \
        double v; asm volatile("v_mov_b64 %0, v[20:21]" : "=v"(v));                                          \
        out[threadIdx.x] = v;                                                                                \
        if (threadIdx.x == 0) cyc[0] = t1 - t0;                                                              \
    }

// sources all at 0 mod 4 (pairs 24, 28, 32), dst 20 (0 mod 4); independent of dst (no dependence chain)
KERNEL(k_same_ind, "v_fma_f64 v[20:21], v[24:25], v[28:29], v[32:33]\n")
// sources at 0, 2, 0 mod 4 (24, 26, 28)
KERNEL(k_mix_ind, "v_fma_f64 v[20:21], v[24:25], v[26:27], v[28:29]\n")
// sources at 2, 2, 2 mod 4 (26, 30, 34)
KERNEL(k_same2_ind, "v_fma_f64 v[20:21], v[26:27], v[30:31], v[34:35]\n")
// dependent chain through src0, other sources same residue as dst
KERNEL(k_dep_same, "v_fma_f64 v[20:21], v[20:21], v[24:25], v[28:29]\n")
// dependent chain, other sources on the other residue
KERNEL(k_dep_mix, "v_fma_f64 v[20:21], v[20:21], v[26:27], v[30:31]\n")
// dependent chain, sources split
KERNEL(k_dep_split, "v_fma_f64 v[20:21], v[20:21], v[24:25], v[30:31]\n")
// v_fmac (VOP2-style accumulate), operands same / mixed
KERNEL(k_fmac_same, "v_fmac_f64_e64 v[20:21], v[24:25], v[28:29]\n")
KERNEL(k_fmac_mix, "v_fmac_f64_e64 v[20:21], v[26:27], v[30:31]\n")
// odd-aligned?? (64-bit operands must be even-aligned on gfx950: not tested)
// two interleaved chains
KERNEL(k_two_chains, "v_fma_f64 v[20:21], v[20:21], v[24:25], v[28:29]\n v_fma_f64 v[22:23], v[22:23], v[26:27], v[30:31]\n")
// 32-bit: three sources same bank / different banks
KERNEL(k32_same, "v_fma_f32 v20, v24, v28, v32\n")
KERNEL(k32_mix, "v_fma_f32 v20, v25, v30, v35\n")

// 32-bit forms: which operand positions collide?
KERNEL(k32_ab, "v_fma_f32 v20, v24, v28, v33\n")          // src0, src1 same residue
KERNEL(k32_ac, "v_fma_f32 v20, v24, v29, v32\n")          // src0, src2
KERNEL(k32_bc, "v_fma_f32 v20, v25, v28, v32\n")          // src1, src2
KERNEL(k32_dst, "v_fma_f32 v24, v25, v30, v35\n")         // sources spread, dst on src... (dst residue 0, sources 1,2,3)
KERNEL(k32_dep_same, "v_fma_f32 v20, v20, v24, v28\n")    // dependent, all residue 0
KERNEL(k32_dep_mix, "v_fma_f32 v20, v20, v25, v30\n")     // dependent, spread
KERNEL(kadd_same, "v_add_u32_e64 v20, v24, v28\n")        // two sources, same residue
KERNEL(kadd_mix, "v_add_u32_e64 v20, v24, v29\n")
KERNEL(kadd_e32_same, "v_add_u32_e32 v20, v24, v28\n v_add_u32_e32 v21, v24, v28\n")   // 4-byte encodings (pairs keep 8-byte alignment)
KERNEL(kadd_e32_mix, "v_add_u32_e32 v20, v24, v29\n v_add_u32_e32 v21, v24, v29\n")
KERNEL(kbfi_same, "v_bfi_b32 v20, v24, v28, v32\n")
KERNEL(kbfi_mix, "v_bfi_b32 v20, v24, v29, v34\n")
KERNEL(kcnd_same, "v_cndmask_b32_e64 v20, v24, v28, vcc\n")
KERNEL(kcnd_mix, "v_cndmask_b32_e64 v20, v24, v29, vcc\n")
KERNEL(kdpp, "v_mov_b32_dpp v20, v24 row_ror:8 row_mask:0xf bank_mask:0xf bound_ctrl:1\n")
KERNEL(kcvt, "v_cvt_f64_i32_e32 v[20:21], v24\n v_cvt_f64_i32_e32 v[22:23], v25\n")
KERNEL(kmovdpp64, "v_mov_b64_dpp v[20:21], v[24:25] row_newbcast:3 row_mask:0xf bank_mask:0xf\n")
KERNEL(kfmacdpp_same, "v_fmac_f64_dpp v[20:21], v[24:25], v[28:29] row_newbcast:3 row_mask:0xf bank_mask:0xf\n v_fmac_f64_dpp v[22:23], v[24:25], v[32:33] row_newbcast:3 row_mask:0xf bank_mask:0xf\n")
KERNEL(kfmacdpp_mix, "v_fmac_f64_dpp v[20:21], v[24:25], v[30:31] row_newbcast:3 row_mask:0xf bank_mask:0xf\n v_fmac_f64_dpp v[22:23], v[26:27], v[32:33] row_newbcast:3 row_mask:0xf bank_mask:0xf\n")

// ---- every instruction kind of the front-end's symbol loop, two per pattern (loop overhead 36 / 128 = 0.28 per instruction) ----
KERNEL(km_mul, "v_mul_f64 v[20:21], v[24:25], v[28:29]\n v_mul_f64 v[22:23], v[26:27], v[30:31]\n")
KERNEL(km_add, "v_add_f64 v[20:21], v[24:25], v[28:29]\n v_add_f64 v[22:23], v[26:27], v[30:31]\n")
KERNEL(km_addabs, "v_add_f64 v[20:21], |v[24:25]|, -|v[28:29]|\n v_add_f64 v[22:23], |v[26:27]|, |v[30:31]|\n")
KERNEL(km_max, "v_max_f64 v[20:21], v[24:25], v[28:29]\n v_min_f64 v[22:23], v[26:27], v[30:31]\n")
KERNEL(km_maxdep, "v_max_f64 v[20:21], v[20:21], v[28:29]\n v_min_f64 v[20:21], v[20:21], v[30:31]\n")
KERNEL(km_rcp, "v_rcp_f64_e32 v[20:21], v[24:25]\n v_rcp_f64_e32 v[22:23], v[26:27]\n")
KERNEL(km_rcp1, "v_rcp_f64_e32 v[20:21], v[24:25]\n v_add_f64 v[22:23], v[26:27], v[30:31]\n")
KERNEL(km_fract, "v_fract_f64_e32 v[20:21], v[24:25]\n v_fract_f64_e32 v[22:23], v[26:27]\n")
KERNEL(km_cvti, "v_cvt_i32_f64_e32 v20, v[24:25]\n v_cvt_i32_f64_e32 v22, v[26:27]\n")
KERNEL(km_cmp, "v_cmp_gt_f64_e64 s[20:21], 0, v[24:25]\n v_cmp_eq_f64_e64 s[22:23], 0, v[26:27]\n")
KERNEL(km_cmpcnd, "v_cmp_gt_f64_e32 vcc, 0, v[24:25]\n v_cndmask_b32_e32 v20, v26, v27, vcc\n")
KERNEL(km_readlane, "v_readlane_b32 s20, v24, 53\n v_readlane_b32 s21, v25, 53\n")
KERNEL(km_readlane_use, "v_readlane_b32 s20, v24, 53\n v_readlane_b32 s21, v25, 53\n v_mul_f64 v[20:21], s[20:21], v[28:29]\n v_add_f64 v[22:23], v[26:27], v[30:31]\n")
KERNEL(km_swap32, "v_permlane32_swap_b32_e32 v20, v22\n v_permlane32_swap_b32_e32 v21, v23\n")
KERNEL(km_swap16, "v_permlane16_swap_b32_e32 v20, v22\n v_permlane16_swap_b32_e32 v21, v23\n")
KERNEL(km_swapadd, "v_mov_b64 v[22:23], v[20:21]\n v_add_f64 v[40:41], v[24:25], v[28:29]\n v_add_f64 v[42:43], v[24:25], v[28:29]\n v_permlane32_swap_b32_e32 v20, v22\n v_permlane32_swap_b32_e32 v21, v23\n v_add_f64 v[20:21], v[20:21], v[22:23]\n")
KERNEL(km_sdwa, "v_sub_u32_sdwa v20, sext(v24), v28 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_0 src1_sel:DWORD\n v_sub_u32_sdwa v22, sext(v25), v29 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1 src1_sel:DWORD\n")
KERNEL(km_bfe, "v_bfe_i32 v20, v24, 0, 16\n v_ashrrev_i32_e64 v22, 16, v25\n")
KERNEL(km_movb64, "v_mov_b64 v[20:21], v[24:25]\n v_mov_b64 v[22:23], 0\n")
KERNEL(km_snop, "s_nop 0\n s_nop 0\n")
KERNEL(km_snop1, "s_nop 1\n s_nop 1\n")
KERNEL(km_salu, "s_add_u32 s20, s20, 1\n s_cmp_eq_u64 s[22:23], 0\n")
KERNEL(km_dppdep, "v_add_f64 v[20:21], v[24:25], v[28:29]\n s_nop 1\n v_mov_b32_dpp v22, v20 row_ror:8 row_mask:0xf bank_mask:0xf bound_ctrl:1\n v_mov_b32_dpp v23, v21 row_ror:8 row_mask:0xf bank_mask:0xf bound_ctrl:1\n")
KERNEL(km_bcastdep, "v_add_f64 v[20:21], v[24:25], v[28:29]\n s_nop 1\n v_mov_b64_dpp v[22:23], v[20:21] row_newbcast:3 row_mask:0xf bank_mask:0xf\n v_mov_b64_dpp v[40:41], v[20:21] row_newbcast:4 row_mask:0xf bank_mask:0xf\n")
KERNEL(km_fmacdep, "v_fmac_f64_dpp v[20:21], v[24:25], v[28:29] row_newbcast:3 row_mask:0xf bank_mask:0xf\n v_fmac_f64_dpp v[20:21], v[26:27], v[30:31] row_newbcast:3 row_mask:0xf bank_mask:0xf\n")

// ---- LDS and store issue / wait as the symbol loop uses them: reads, N independent fp64 adds, then the wait ----
#define ADD4 "v_add_f64 v[40:41], v[24:25], v[28:29]\n v_add_f64 v[42:43], v[26:27], v[30:31]\n v_add_f64 v[44:45], v[24:25], v[28:29]\n v_add_f64 v[46:47], v[26:27], v[30:31]\n"
KERNEL(kl_128x2_16, "ds_read_b128 v[20:23], v36\n ds_read_b128 v[32:35], v36 offset:16\n" ADD4 ADD4 ADD4 ADD4 "s_waitcnt lgkmcnt(0)\n")
KERNEL(kl_128x2_24, "ds_read_b128 v[20:23], v36\n ds_read_b128 v[32:35], v36 offset:16\n" ADD4 ADD4 ADD4 ADD4 ADD4 ADD4 "s_waitcnt lgkmcnt(0)\n")
KERNEL(kl_128x2_32, "ds_read_b128 v[20:23], v36\n ds_read_b128 v[32:35], v36 offset:16\n" ADD4 ADD4 ADD4 ADD4 ADD4 ADD4 ADD4 ADD4 "s_waitcnt lgkmcnt(0)\n")
KERNEL(kl_r2_16, "ds_read2_b32 v[20:21], v38 offset1:1\n" ADD4 ADD4 ADD4 ADD4 "s_waitcnt lgkmcnt(0)\n")
KERNEL(kl_r2_24, "ds_read2_b32 v[20:21], v38 offset1:1\n" ADD4 ADD4 ADD4 ADD4 ADD4 ADD4 "s_waitcnt lgkmcnt(0)\n")
KERNEL(kl_none_16, ADD4 ADD4 ADD4 ADD4 "s_waitcnt lgkmcnt(0)\n")
KERNEL(kl_store_16, "global_store_dwordx2 v37, v[24:25], s[24:25]\n" ADD4 ADD4 ADD4 ADD4)

// ---- scalar branches between vector work ----
KERNEL(kb_none, ADD4 ADD4)
KERNEL(kb_nt, ADD4 "s_cmp_eq_u64 s[22:23], 0\n s_cbranch_scc0 9f\n" ADD4 "9:\n")          // s[22:23] = 0: never taken
KERNEL(kb_vcc, ADD4 "v_cmp_eq_f64_e32 vcc, 0, v[24:25]\n s_cbranch_vccnz 9f\n" ADD4 "9:\n")   // v[24:25] = 0.5: never taken, compare next to its branch
KERNEL(kb_vcc_far, "v_cmp_eq_f64_e64 s[22:23], 0, v[24:25]\n" ADD4 ADD4 "s_cmp_eq_u64 s[22:23], 0\n s_cbranch_scc0 9f\n 9:\n")
KERNEL(kb_addr, "v_cvt_i32_f64_e32 v40, v[24:25]\n v_add_lshl_u32 v40, v38, v40, 2\n v_and_b32_e32 v40, 0x3ffc, v40\n ds_read2_b32 v[20:21], v40 offset1:1\n" ADD4 ADD4 ADD4 ADD4 "s_waitcnt lgkmcnt(0)\n v_add_f64 v[42:43], v[20:21], v[20:21]\n")

typedef void (*kern_t)(double*, unsigned long long*, int);
static void run(const char* name, kern_t k, double* d, unsigned long long* c, int per) {
    for (int w = 0; w < 3; ++w) k<<<1, 64>>>(d, c, 2000);
    hipDeviceSynchronize();
    unsigned long long cy; hipMemcpy(&cy, c, 8, hipMemcpyDeviceToHost);
    printf("%-44s %.2f cycles per instruction\n", name, (double)cy / 2000 / (64 * per));
}

int main() {
    double* d; unsigned long long* c;
    hipMalloc(&d, 4096 * 8); hipMalloc(&c, 8);
    run("fma_f64 independent, sources 0,0,0 mod 4", k_same_ind, d, c, 1);
    run("fma_f64 independent, sources 0,2,0 mod 4", k_mix_ind, d, c, 1);
    run("fma_f64 independent, sources 2,2,2 mod 4", k_same2_ind, d, c, 1);
    run("fma_f64 dependent, others 0,0 (dst 0)", k_dep_same, d, c, 1);
    run("fma_f64 dependent, others 2,2 (dst 0)", k_dep_mix, d, c, 1);
    run("fma_f64 dependent, others 0,2 (dst 0)", k_dep_split, d, c, 1);
    run("fmac_f64 sources 0,0 (dst 0)", k_fmac_same, d, c, 1);
    run("fmac_f64 sources 2,2 (dst 0)", k_fmac_mix, d, c, 1);
    run("fma_f64 two chains interleaved", k_two_chains, d, c, 2);
    run("fma_f32 sources 0,0,0 mod 4", k32_same, d, c, 1);
    run("fma_f32 sources 1,2,3 mod 4", k32_mix, d, c, 1);
    run("fma_f32 src0,src1 same residue", k32_ab, d, c, 1);
    run("fma_f32 src0,src2 same residue", k32_ac, d, c, 1);
    run("fma_f32 src1,src2 same residue", k32_bc, d, c, 1);
    run("fma_f32 sources spread, dst residue 0", k32_dst, d, c, 1);
    run("fma_f32 dependent, all residue 0", k32_dep_same, d, c, 1);
    run("fma_f32 dependent, spread", k32_dep_mix, d, c, 1);
    run("add_u32 e64, sources same residue", kadd_same, d, c, 1);
    run("add_u32 e64, sources differ", kadd_mix, d, c, 1);
    run("add_u32 e32 pairs, sources same residue", kadd_e32_same, d, c, 2);
    run("add_u32 e32 pairs, sources differ", kadd_e32_mix, d, c, 2);
    run("bfi_b32 sources same residue", kbfi_same, d, c, 1);
    run("bfi_b32 sources spread", kbfi_mix, d, c, 1);
    run("cndmask e64 sources same residue", kcnd_same, d, c, 1);
    run("cndmask e64 sources differ", kcnd_mix, d, c, 1);
    run("mov_b32_dpp", kdpp, d, c, 1);
    run("cvt_f64_i32 e32 pairs", kcvt, d, c, 2);
    run("mov_b64_dpp row_newbcast", kmovdpp64, d, c, 1);
    run("fmac_f64_dpp x2, src1 residues 0,0", kfmacdpp_same, d, c, 2);
    run("fmac_f64_dpp x2, src1 residues 2,0 / src0 0,2", kfmacdpp_mix, d, c, 2);
    printf("---- instruction kinds of the symbol loop (per instruction; 0.28 of it is the measuring loop)\n");
    run("v_mul_f64 x2", km_mul, d, c, 2);
    run("v_add_f64 x2", km_add, d, c, 2);
    run("v_add_f64 with |.| modifiers x2", km_addabs, d, c, 2);
    run("v_max_f64 + v_min_f64 independent", km_max, d, c, 2);
    run("v_max_f64 -> v_min_f64 dependent", km_maxdep, d, c, 2);
    run("v_rcp_f64 x2", km_rcp, d, c, 2);
    run("v_rcp_f64 + v_add_f64", km_rcp1, d, c, 2);
    run("v_fract_f64 x2", km_fract, d, c, 2);
    run("v_cvt_i32_f64 x2", km_cvti, d, c, 2);
    run("v_cmp_f64 -> SGPR pair x2", km_cmp, d, c, 2);
    run("v_cmp_f64 vcc + v_cndmask", km_cmpcnd, d, c, 2);
    run("v_readlane x2", km_readlane, d, c, 2);
    run("v_readlane x2 + use as scalar operand + add", km_readlane_use, d, c, 4);
    run("v_permlane32_swap x2", km_swap32, d, c, 2);
    run("v_permlane16_swap x2", km_swap16, d, c, 2);
    run("mov, 2 fillers, swap32 x2, add (one stage)", km_swapadd, d, c, 6);
    run("v_sub_u32_sdwa x2", km_sdwa, d, c, 2);
    run("v_bfe_i32 + v_ashrrev_i32", km_bfe, d, c, 2);
    run("v_mov_b64 x2", km_movb64, d, c, 2);
    run("s_nop 0 x2", km_snop, d, c, 2);
    run("s_nop 1 x2", km_snop1, d, c, 2);
    run("s_add_u32 + s_cmp_eq_u64", km_salu, d, c, 2);
    run("add, s_nop 1, mov_b32_dpp x2 (dependent)", km_dppdep, d, c, 4);
    run("add, s_nop 1, mov_b64_dpp x2 (dependent)", km_bcastdep, d, c, 4);
    run("v_fmac_f64_dpp x2 on ONE accumulator", km_fmacdep, d, c, 2);
    printf("---- scalar branches between 8 v_add_f64 (cycles per PATTERN)\n");
    run("8 adds", kb_none, d, c, 1);
    run("8 adds + s_cmp + s_cbranch_scc0 not taken", kb_nt, d, c, 1);
    run("8 adds + v_cmp vcc + s_cbranch_vccnz adjacent, not taken", kb_vcc, d, c, 1);
    run("v_cmp -> SGPR, 8 adds, s_cmp + s_cbranch not taken", kb_vcc_far, d, c, 1);
    run("cvt, add_lshl, and, ds_read2, 16 adds, wait, use", kb_addr, d, c, 1);
    printf("---- LDS reads / store behind N independent v_add_f64 and a wait (cycles per PATTERN; N adds alone = 4 N)\n");
    run("16 adds + s_waitcnt (nothing outstanding)", kl_none_16, d, c, 1);
    run("2 x ds_read_b128, 16 adds, wait", kl_128x2_16, d, c, 1);
    run("2 x ds_read_b128, 24 adds, wait", kl_128x2_24, d, c, 1);
    run("2 x ds_read_b128, 32 adds, wait", kl_128x2_32, d, c, 1);
    run("ds_read2_b32, 16 adds, wait", kl_r2_16, d, c, 1);
    run("ds_read2_b32, 24 adds, wait", kl_r2_24, d, c, 1);
    run("global_store_dwordx2 (same address), 16 adds", kl_store_16, d, c, 1);
    return 0;
}
