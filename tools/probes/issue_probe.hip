// probe: what does ONE wave alone on a SIMD pay per instruction on gfx950?  physics_kernel runs one wave per SIMD (512 registers), so its time is
// the sum of these prices over its ~66 k-instruction stream.  Each case is N copies of a short pattern inside one wave; cycles / instruction
// from s_memtime around it (shader clock).  Patterns: independent / dependent v_fma, DPP adds (dependent, independent, mov + add), packed fp32,
// v_rcp, compare -> select through VCC / an SGPR pair, SALU between VALU, branches taken / not taken, LDS read -> use.
// build + run on the GPU box:  hipcc --offload-arch=gfx950 -O2 tools/probes/issue_probe.hip -o /tmp/issue_probe && /tmp/issue_probe
#include <hip/hip_runtime.h>
#include <cstdio>

#define REP 512
#define STR2(x) #x
#define STR(x) STR2(x)

// time one asm body (already repeated with .rept) : returns cycles
#define TIMED(name, nper, body, clob...)                                                                        \
  {                                                                                                            \
    long long t0, t1;                                                                                          \
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0) :: "memory"); \
    asm volatile(".rept " STR(REP) "\n\t" body "\n\t.endr" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7), "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(c0), "v"(c1), "v"(pc) : "vcc", "scc", "s20", "s21", "s22", "s23", "s24", "s25", "s26", "s27", "memory"); \
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1) :: "memory");                                \
    if (threadIdx.x == 0) { out[idx * 2] = (float)(t1 - t0) / (REP * (nper)); out[idx * 2 + 1] = (float)(nper); }   \
    idx++;                                                                                                     \
  }

typedef float f2 __attribute__((ext_vector_type(2)));

__global__ __launch_bounds__(64) void probe(float* out, float seed) {
  float a0 = seed + threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
  f2 p0{a0, a1}, p1{a2, a3}, p2{a4, a5}, p3{a6, a7}, pc{0.999f, 1.001f};
  float c0 = 0.999f, c1 = 1e-3f;
  __shared__ float sh[256];
  sh[threadIdx.x] = a0; sh[threadIdx.x + 64] = a1;
  __syncthreads();
  int idx = 0;
  // 0 empty timing overhead reference: one s_nop
  TIMED("s_nop", 1, "s_nop 0")
  // 1 independent fma x8
  TIMED("fma indep8", 8, "v_fma_f32 %0, %0, %12, %13\n\tv_fma_f32 %1, %1, %12, %13\n\tv_fma_f32 %2, %2, %12, %13\n\tv_fma_f32 %3, %3, %12, %13\n\tv_fma_f32 %4, %4, %12, %13\n\tv_fma_f32 %5, %5, %12, %13\n\tv_fma_f32 %6, %6, %12, %13\n\tv_fma_f32 %7, %7, %12, %13")
  // 2 dependent fma
  TIMED("fma dep", 1, "v_fma_f32 %0, %0, %12, %13")
  // 3 two interleaved chains
  TIMED("fma dep2", 2, "v_fma_f32 %0, %0, %12, %13\n\tv_fma_f32 %1, %1, %12, %13")
  // 4 dependent DPP add (butterfly step on the same register)
  TIMED("dpp add dep", 1, "v_add_f32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf bound_ctrl:1")
  // 5 independent DPP adds x4
  TIMED("dpp add indep4", 4, "v_add_f32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf bound_ctrl:1\n\tv_add_f32_dpp %1, %1, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf bound_ctrl:1\n\tv_add_f32_dpp %2, %2, %2 row_ror:8 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\tv_add_f32_dpp %3, %3, %3 row_ror:4 row_mask:0xf bank_mask:0xf bound_ctrl:1")
  // 6 the 4-step sixteen-lane sum of ONE value (dependent chain of 4 DPP adds) followed by a plain add
  TIMED("sum16 chain", 5, "v_add_f32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf bound_ctrl:1\n\tv_add_f32_dpp %0, %0, %0 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf bound_ctrl:1\n\tv_add_f32_dpp %0, %0, %0 row_ror:8 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\tv_add_f32_dpp %0, %0, %0 row_ror:4 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\tv_mul_f32 %0, %0, %13")
  // 7 plain VALU feeding a DPP read of its result (the hazard the compiler pads with s_nop)
  TIMED("valu->dpp", 2, "v_mul_f32 %0, %0, %12\n\ts_nop 1\n\tv_add_f32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf bound_ctrl:1")
  // 8 packed fma independent x4 / 9 dependent
  TIMED("pk_fma indep4", 4, "v_pk_fma_f32 %8, %8, %14, %14\n\tv_pk_fma_f32 %9, %9, %14, %14\n\tv_pk_fma_f32 %10, %10, %14, %14\n\tv_pk_fma_f32 %11, %11, %14, %14")
  TIMED("pk_fma dep", 1, "v_pk_fma_f32 %8, %8, %14, %14")
  // 10 rcp independent x4 / 11 dependent
  TIMED("rcp indep4", 4, "v_rcp_f32 %0, %0\n\tv_rcp_f32 %1, %1\n\tv_rcp_f32 %2, %2\n\tv_rcp_f32 %3, %3")
  TIMED("rcp dep", 1, "v_rcp_f32 %0, %0")
  // 12 rcp followed by 3 independent fmas (does the transcendental overlap?)
  TIMED("rcp + 3 fma", 4, "v_rcp_f32 %0, %0\n\tv_fma_f32 %1, %1, %12, %13\n\tv_fma_f32 %2, %2, %12, %13\n\tv_fma_f32 %3, %3, %12, %13")
  // 13 compare -> select through VCC (dependent pair)
  TIMED("cmp vcc + cndmask", 2, "v_cmp_lt_f32 vcc, %0, %12\n\tv_cndmask_b32 %0, %0, %1, vcc")
  // 14 compare into an SGPR pair, mask logic on the scalar unit, select
  TIMED("cmp sgpr + s_and + cndmask", 4, "v_cmp_lt_f32 s[20:21], %0, %12\n\tv_cmp_gt_f32 s[22:23], %1, %13\n\ts_and_b64 s[20:21], s[20:21], s[22:23]\n\tv_cndmask_b32 %0, %0, %1, s[20:21]")
  // 15 SALU between independent VALU: does a scalar instruction take an issue slot of the wave?
  TIMED("fma + s_and alternating", 2, "v_fma_f32 %0, %0, %12, %13\n\ts_and_b64 s[24:25], s[24:25], s[26:27]")
  TIMED("fma indep2 + 2 salu", 4, "v_fma_f32 %0, %0, %12, %13\n\ts_and_b64 s[24:25], s[24:25], s[26:27]\n\tv_fma_f32 %1, %1, %12, %13\n\ts_or_b64 s[20:21], s[20:21], s[22:23]")
  // 17 branch not taken / 18 taken (to the next instruction)
  TIMED("branch not taken + fma", 3, "s_cmp_eq_u32 0, 1\n\ts_cbranch_scc1 1f\n\tv_fma_f32 %0, %0, %12, %13\n1:")
  TIMED("branch taken + fma", 3, "s_cmp_eq_u32 0, 0\n\ts_cbranch_scc1 1f\n\tv_fma_f32 %0, %0, %12, %13\n1:\n\tv_fma_f32 %1, %1, %12, %13")
  // 19 vcc-based uniform branch as the compiler emits it: v_cmp -> s_cbranch_vccz
  TIMED("v_cmp + s_cbranch_vccz (not taken)", 3, "v_cmp_lt_f32 vcc, %13, %12\n\ts_cbranch_vccz 1f\n\tv_fma_f32 %0, %0, %12, %13\n1:")
  // 20 mov_dpp + add (the un-fused last butterfly step)
  TIMED("mov_dpp + add", 2, "v_mov_b32_dpp %1, %0 row_ror:4 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\tv_add_f32 %0, %0, %1")
  // 21 v_min3 / v_med3 (3-input ops) dependent
  TIMED("min3 dep", 1, "v_min3_f32 %0, %0, %1, %2")
  // 22 fma chain with 4 accumulators (ILP 4)
  TIMED("fma dep4", 4, "v_fma_f32 %0, %0, %12, %13\n\tv_fma_f32 %1, %1, %12, %13\n\tv_fma_f32 %2, %2, %12, %13\n\tv_fma_f32 %3, %3, %12, %13")
  // 23 s_and_saveexec region entry / exit around one fma
  TIMED("saveexec region", 3, "s_and_saveexec_b64 s[20:21], vcc\n\tv_fma_f32 %0, %0, %12, %13\n\ts_or_b64 exec, exec, s[20:21]")
  // 24 v_readfirstlane + s_cmp + scalar branch
  TIMED("pk_mul clamp + pk_fma pair", 2, "v_pk_mul_f32 %8, %9, %14 clamp\n\tv_pk_fma_f32 %10, %8, %14, %10")
  // 25 v_pk_fma with op_sel broadcast (as the line search emits)
  TIMED("pk_fma op_sel", 2, "v_pk_fma_f32 %8, %9, %14, %10 op_sel_hi:[0,1,1]\n\tv_pk_fma_f32 %11, %10, %14, %9 op_sel_hi:[0,1,1]")

  // ---- encoding size: the same arithmetic as 4-byte VOP2 (v_fmac / v_add e32) and as 8-byte VOP3 (v_fma, e64 forms)
  TIMED("fmac_e32 indep8", 8, "v_fmac_f32_e32 %0, %12, %13\n\tv_fmac_f32_e32 %1, %12, %13\n\tv_fmac_f32_e32 %2, %12, %13\n\tv_fmac_f32_e32 %3, %12, %13\n\tv_fmac_f32_e32 %4, %12, %13\n\tv_fmac_f32_e32 %5, %12, %13\n\tv_fmac_f32_e32 %6, %12, %13\n\tv_fmac_f32_e32 %7, %12, %13")
  TIMED("fmac_e32 dep", 1, "v_fmac_f32_e32 %0, %12, %13")
  TIMED("add_e32 dep", 1, "v_add_f32_e32 %0, %0, %12")
  TIMED("add_e64 dep", 1, "v_add_f32_e64 %0, %0, %12")
  TIMED("mul_e32 indep4", 4, "v_mul_f32_e32 %0, %0, %12\n\tv_mul_f32_e32 %1, %1, %12\n\tv_mul_f32_e32 %2, %2, %12\n\tv_mul_f32_e32 %3, %3, %12")
  TIMED("mul_e64 indep4", 4, "v_mul_f32_e64 %0, %0, %12\n\tv_mul_f32_e64 %1, %1, %12\n\tv_mul_f32_e64 %2, %2, %12\n\tv_mul_f32_e64 %3, %3, %12")
  TIMED("mul literal (8B) indep4", 4, "v_mul_f32_e32 %0, 0x3f7fbe77, %0\n\tv_mul_f32_e32 %1, 0x3f7fbe77, %1\n\tv_mul_f32_e32 %2, 0x3f7fbe77, %2\n\tv_mul_f32_e32 %3, 0x3f7fbe77, %3")
  TIMED("cndmask e32 vcc indep4", 4, "v_cndmask_b32_e32 %0, %0, %1, vcc\n\tv_cndmask_b32_e32 %2, %2, %3, vcc\n\tv_cndmask_b32_e32 %4, %4, %5, vcc\n\tv_cndmask_b32_e32 %6, %6, %7, vcc")
  TIMED("cndmask e64 sgpr indep4", 4, "v_cndmask_b32_e64 %0, %0, %1, s[20:21]\n\tv_cndmask_b32_e64 %2, %2, %3, s[20:21]\n\tv_cndmask_b32_e64 %4, %4, %5, s[20:21]\n\tv_cndmask_b32_e64 %6, %6, %7, s[20:21]")
  // the same 8-fma pattern executed from a LOOP whose body (64 instructions, 512 B) stays resident: is the straight-line price a fetch price?
  {
    long long t0, t1; int cnt = REP / 8;
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0) :: "memory");
    asm volatile("s_mov_b32 s20, %14\n2:\n\t.rept 8\n\tv_fma_f32 %0, %0, %12, %13\n\tv_fma_f32 %1, %1, %12, %13\n\tv_fma_f32 %2, %2, %12, %13\n\tv_fma_f32 %3, %3, %12, %13\n\tv_fma_f32 %4, %4, %12, %13\n\tv_fma_f32 %5, %5, %12, %13\n\tv_fma_f32 %6, %6, %12, %13\n\tv_fma_f32 %7, %7, %12, %13\n\t.endr\n\ts_sub_u32 s20, s20, 1\n\ts_cmp_lg_u32 s20, 0\n\ts_cbranch_scc1 2b"
                 : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7), "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(c0), "v"(c1), "s"(cnt) : "vcc", "scc", "s20", "memory");
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1) :: "memory");
    if (threadIdx.x == 0) { out[idx * 2] = (float)(t1 - t0) / (REP * 8); out[idx * 2 + 1] = 8.f; }
    idx++;
  }
  {
    long long t0, t1; int cnt = REP / 8;
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0) :: "memory");
    asm volatile("s_mov_b32 s20, %14\n2:\n\t.rept 8\n\tv_fmac_f32_e32 %0, %12, %13\n\tv_fmac_f32_e32 %1, %12, %13\n\tv_fmac_f32_e32 %2, %12, %13\n\tv_fmac_f32_e32 %3, %12, %13\n\tv_fmac_f32_e32 %4, %12, %13\n\tv_fmac_f32_e32 %5, %12, %13\n\tv_fmac_f32_e32 %6, %12, %13\n\tv_fmac_f32_e32 %7, %12, %13\n\t.endr\n\ts_sub_u32 s20, s20, 1\n\ts_cmp_lg_u32 s20, 0\n\ts_cbranch_scc1 2b"
                 : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7), "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(c0), "v"(c1), "s"(cnt) : "vcc", "scc", "s20", "memory");
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1) :: "memory");
    if (threadIdx.x == 0) { out[idx * 2] = (float)(t1 - t0) / (REP * 8); out[idx * 2 + 1] = 8.f; }
    idx++;
  }
  out[120] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + p0.x + p1.y + p2.x + p3.y;
}

int main() {
  float* d; hipMalloc(&d, 1024);
  float h[256];
  const char* names[] = {"s_nop 0", "v_fma independent x8", "v_fma dependent chain", "v_fma two chains", "v_add_dpp dependent", "v_add_dpp independent x4",
                         "16-lane sum chain (4 dpp + mul)", "v_mul; s_nop 1; v_add_dpp (dependent)", "v_pk_fma independent x4", "v_pk_fma dependent", "v_rcp independent x4",
                         "v_rcp dependent", "v_rcp + 3 fma", "v_cmp vcc + v_cndmask (dep)", "2 v_cmp sgpr + s_and + v_cndmask", "v_fma + s_and alternating",
                         "2 fma + 2 salu", "s_cmp + branch not taken + fma", "s_cmp + branch taken + 2 fma (3 counted)", "v_cmp + s_cbranch_vccz not taken + fma", "v_mov_dpp + v_add",
                         "v_min3 dependent", "v_fma four chains", "saveexec + fma + restore", "pk_mul clamp + pk_fma", "pk_fma op_sel x2",
                         "v_fmac_e32 (4 B) independent x8", "v_fmac_e32 dependent", "v_add_e32 dependent", "v_add_e64 (8 B) dependent", "v_mul_e32 independent x4", "v_mul_e64 (8 B) independent x4", "v_mul_e32 + literal (8 B) independent x4", "v_cndmask_e32 vcc x4", "v_cndmask_e64 sgpr (8 B) x4", "LOOP of 64 v_fma (8 B)", "LOOP of 64 v_fmac_e32 (4 B)"};
  for (int blocks : {1, 1, 1024}) {
    hipMemset(d, 0, 1024);
    hipLaunchKernelGGL(probe, dim3(blocks), dim3(64), 0, 0, d, 1.0f);
    hipDeviceSynchronize();
    hipMemcpy(h, d, 1024, hipMemcpyDeviceToHost);
    printf("== %d block(s) of one wave (block 0 reports; 1024 = one wave per SIMD, 2048 = two)\n", blocks);
    for (int i = 0; i < 37; i++) printf("  %-44s %6.2f cycles / instruction (%d per pattern -> %6.1f cycles / pattern)\n", names[i], h[2 * i], (int)h[2 * i + 1], h[2 * i] * h[2 * i + 1]);
  }
  return 0;
}
