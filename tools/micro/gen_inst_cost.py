#!/usr/bin/env python3
"""Generates inst_cost.hip: per-instruction issue cost microbenchmark for gfx950 (see the header it writes)."""
K = []  # (name, decl, asm template with {a} accumulator and operands, constraint builder, use-expr)
def k(name, ctype, init, asm, ins, sreg=False):
    K.append((name, ctype, init, asm, ins, sreg))
k("v_fma_f32", "float", "threadIdx.x", "v_fma_f32 %0, %1, %2, %0", [("v", "float", "1.0001f"), ("v", "float", "0.5f")])
k("v_fma_f32_sgpr", "float", "threadIdx.x", "v_fma_f32 %0, %1, %2, %0", [("s", "float", "1.0001f"), ("v", "float", "0.5f")])
k("v_add_u32", "unsigned", "threadIdx.x", "v_add_u32 %0, %1, %0", [("v", "unsigned", "3u")])
k("v_cndmask_b32", "unsigned", "threadIdx.x", "v_cndmask_b32 %0, %1, %0, vcc", [("v", "unsigned", "3u")])
k("v_cmp_lt_u32", "unsigned", "threadIdx.x", "v_cmp_lt_u32 vcc, %0, %1", [("v", "unsigned", "3u")])
k("v_cmp_lt_u64", "unsigned long long", "threadIdx.x", "v_cmp_lt_u64 vcc, %0, %1", [("v", "unsigned long long", "3ull")])
k("v_cvt_f32_u32", "float", "threadIdx.x", "v_cvt_f32_u32 %0, %1", [("v", "unsigned", "3u")])
k("v_lshl_add_u64", "unsigned long long", "threadIdx.x", "v_lshl_add_u64 %0, %0, 0, %1", [("v", "unsigned long long", "3ull")])
k("v_fma_f64", "double", "threadIdx.x", "v_fma_f64 %0, %1, %2, %0", [("v", "double", "1.0001"), ("v", "double", "0.5")])
k("v_fma_f64_sgpr", "double", "threadIdx.x", "v_fma_f64 %0, %1, %2, %0", [("s", "double", "1.0001"), ("v", "double", "0.5")])
k("v_mul_f64", "double", "threadIdx.x", "v_mul_f64 %0, %1, %0", [("v", "double", "1.0001")])
k("v_add_f64", "double", "threadIdx.x", "v_add_f64 %0, %1, %0", [("v", "double", "1.0001")])
k("v_cvt_f64_f32", "double", "threadIdx.x", "v_cvt_f64_f32 %0, %1", [("v", "float", "1.5f")])
k("v_cvt_f32_f64", "float", "threadIdx.x", "v_cvt_f32_f64 %0, %1", [("v", "double", "1.5")])
k("v_cvt_u32_f64", "unsigned", "threadIdx.x", "v_cvt_u32_f64 %0, %1", [("v", "double", "1.5")])
k("v_cvt_f64_u32", "double", "threadIdx.x", "v_cvt_f64_u32 %0, %1", [("v", "unsigned", "7u")])
k("v_ldexp_f64", "double", "threadIdx.x", "v_ldexp_f64 %0, %0, %1", [("v", "int", "1")])
k("v_exp_f32", "float", "threadIdx.x", "v_exp_f32 %0, %1", [("v", "float", "0.5f")])
k("v_rcp_f32", "float", "threadIdx.x", "v_rcp_f32 %0, %1", [("v", "float", "1.5f")])
k("v_add_f32_dpp", "float", "threadIdx.x", "v_add_f32_dpp %0, %1, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf", [("v", "float", "1.5f")])
k("v_mov_b32", "float", "threadIdx.x", "v_mov_b32 %0, %1", [("v", "float", "1.5f")])
k("v_mov_b64", "double", "threadIdx.x", "v_mov_b64 %0, %1", [("v", "double", "1.5")])
k("v_readfirstlane", "unsigned", "blockIdx.x", "v_readfirstlane_b32 %0, %1", [("v", "unsigned", "threadIdx.x")], True)
k("s_add_u32", "unsigned", "blockIdx.x", "s_add_u32 %0, %0, 1", [], True)
k("s_cmp_lt_i32", "unsigned", "blockIdx.x", "s_cmp_lt_i32 %0, 6", [], True)
k("branch_not_taken", "unsigned", "blockIdx.x", "s_cmp_lt_i32 %0, 0\\n s_cbranch_scc1 1f\\n s_add_u32 %0, %0, 1\\n1:", [], True)
k("branch_taken", "unsigned", "blockIdx.x", "s_cmp_gt_i32 %0, -1\\n s_cbranch_scc1 1f\\n s_add_u32 %0, %0, 1\\n1:", [], True)
k("exec_branch", "unsigned", "threadIdx.x", "v_cmp_gt_u32 vcc, 1000, %0\\n s_and_saveexec_b64 s[20:21], vcc\\n s_cbranch_execz 1f\\n v_add_u32 %0, 1, %0\\n1: s_or_b64 exec, exec, s[20:21]", [])
k("v_cndmask_e64_sgprmask", "unsigned", "threadIdx.x", "v_cndmask_b32_e64 %0, %1, %0, s[22:23]", [("v", "unsigned", "3u")])
k("v_cmp_then_cndmask", "unsigned", "threadIdx.x", "v_cmp_lt_u32 vcc, %0, %1\\n v_cndmask_b32 %0, %1, %0, vcc", [("v", "unsigned", "3u")])
k("v_cmp_e64_then_cndmask_e64", "unsigned", "threadIdx.x", "v_cmp_lt_u32_e64 s[22:23], %0, %1\\n v_cndmask_b32_e64 %0, %1, %0, s[22:23]", [("v", "unsigned", "3u")])
k("v_mul_f32", "float", "threadIdx.x", "v_mul_f32 %0, %1, %0", [("v", "float", "1.0001f")])
k("v_add_f32", "float", "threadIdx.x", "v_add_f32 %0, %1, %0", [("v", "float", "1.0001f")])
k("v_mul_f32_sgpr", "float", "threadIdx.x", "v_mul_f32 %0, %1, %0", [("s", "float", "1.0001f")])
k("v_mul_f32_inline", "float", "threadIdx.x", "v_mul_f32 %0, 2.0, %0", [])
k("v_fmaak_f32_literal", "float", "threadIdx.x", "v_fmaak_f32 %0, %1, %0, 0x3f800001", [("v", "float", "1.0001f")])
k("v_fma_f32_inline", "float", "threadIdx.x", "v_fma_f32 %0, %1, %0, 1.0", [("v", "float", "1.0001f")])
k("v_fma_f32_neg_abs", "float", "threadIdx.x", "v_fma_f32 %0, -%1, |%0|, 1.0", [("v", "float", "1.0001f")])
k("v_max_f32", "float", "threadIdx.x", "v_max_f32 %0, %1, %0", [("v", "float", "1.0001f")])
k("v_and_b32", "unsigned", "threadIdx.x", "v_and_b32 %0, %1, %0", [("v", "unsigned", "0xffffu")])
k("v_xor_b32", "unsigned", "threadIdx.x", "v_xor_b32 %0, %1, %0", [("v", "unsigned", "0xffffu")])
k("v_lshlrev_b32", "unsigned", "threadIdx.x", "v_lshlrev_b32 %0, 1, %0", [])
k("v_sub_u32", "unsigned", "threadIdx.x", "v_sub_u32 %0, %0, %1", [("v", "unsigned", "3u")])
k("v_add_co_u32", "unsigned", "threadIdx.x", "v_add_co_u32 %0, vcc, %1, %0", [("v", "unsigned", "3u")])
k("v_addc_co_u32", "unsigned", "threadIdx.x", "v_addc_co_u32 %0, vcc, %1, %0, vcc", [("v", "unsigned", "3u")])
k("v_mul_lo_u32", "unsigned", "threadIdx.x", "v_mul_lo_u32 %0, %1, %0", [("v", "unsigned", "3u")])
k("v_cvt_f32_i32", "float", "threadIdx.x", "v_cvt_f32_i32 %0, %1", [("v", "int", "3")])
k("v_cvt_i32_f32", "int", "threadIdx.x", "v_cvt_i32_f32 %0, %1", [("v", "float", "3.5f")])
k("v_floor_f32", "float", "threadIdx.x", "v_floor_f32 %0, %1", [("v", "float", "3.5f")])
k("v_med3_f32", "float", "threadIdx.x", "v_med3_f32 %0, %0, %1, 1.0", [("v", "float", "0.5f")])
k("v_bfe_u32", "unsigned", "threadIdx.x", "v_bfe_u32 %0, %0, 3, 5", [])
k("v_trunc_f64", "double", "threadIdx.x", "v_trunc_f64 %0, %1", [("v", "double", "3.5")])
k("v_floor_f64", "double", "threadIdx.x", "v_floor_f64 %0, %1", [("v", "double", "3.5")])
k("v_cvt_f64_i32", "double", "threadIdx.x", "v_cvt_f64_i32 %0, %1", [("v", "int", "3")])
k("v_pk_fma_f32", "double", "threadIdx.x", "v_pk_fma_f32 %0, %1, %1, %0", [("v", "double", "1.5")])
k("v_pk_mul_f32", "double", "threadIdx.x", "v_pk_mul_f32 %0, %1, %0", [("v", "double", "1.5")])
k("v_pk_add_f32", "double", "threadIdx.x", "v_pk_add_f32 %0, %1, %0", [("v", "double", "1.5")])
k("s_mov_b32", "unsigned", "blockIdx.x", "s_mov_b32 %0, 5", [], True)
k("s_and_b64_exec", "unsigned", "blockIdx.x", "s_and_b64 s[22:23], exec, s[22:23]", [], True)
k("s_nop", "unsigned", "blockIdx.x", "s_nop 0", [], True)
k("fma_then_s_add (pair)", "float", "threadIdx.x", "v_fma_f32 %0, %1, %2, %0\\n s_add_u32 s20, s20, 1", [("v", "float", "1.0001f"), ("v", "float", "0.5f")])
k("fma64_then_s_add (pair)", "double", "threadIdx.x", "v_fma_f64 %0, %1, %2, %0\\n s_add_u32 s20, s20, 1", [("v", "double", "1.0001"), ("v", "double", "0.5")])
k("fma_then_2x_s_add (triple)", "float", "threadIdx.x", "v_fma_f32 %0, %1, %2, %0\\n s_add_u32 s20, s20, 1\\n s_add_u32 s21, s21, 1", [("v", "float", "1.0001f"), ("v", "float", "0.5f")])
k("fma_then_taken_branch (pair+)", "float", "threadIdx.x", "v_fma_f32 %0, %1, %2, %0\\n s_cmp_gt_i32 s20, -1\\n s_cbranch_scc1 1f\\n s_nop 0\\n1:", [("v", "float", "1.0001f"), ("v", "float", "0.5f")])
out = ['''// GENERATED by gen_inst_cost.py.  Per-instruction issue cost on gfx950 (MI355X): SIMD-time per wave64
// instruction, from runs of 32 inline-asm instructions over 4 rotating accumulators, at 1 / 4 / 8 waves per SIMD.
//   hipcc --offload-arch=gfx950 -O3 -o inst_cost inst_cost.hip && ./inst_cost
#include <hip/hip_runtime.h>
#include <cstdio>
''']
for name, ctype, init, asm, ins, sreg in K:
    c = "s" if sreg else "v"
    decl = "".join(f"  {ctype} a{i} = ({ctype})({init}) + {i};\n" for i in range(4))
    decl += "".join(f"  {t} b{j} = {v};\n" for j, (_, t, v) in enumerate(ins))
    # eight instructions per asm statement (round 4): the compiler pads EVERY inline-asm statement with an `s_nop 0` it cannot
    # prove unnecessary, and with one instruction per statement round 3's figures had a scalar instruction between every two
    import re as _re
    body = ""
    cons_in = ", ".join(f'"{cc}"(b{j})' for j, (cc, _, _) in enumerate(ins))
    for g in range(4):
        lines = [_re.sub(r"%(\d)", lambda m, r=r: f"%{r % 4}" if m.group(1) == "0" else f"%{3 + int(m.group(1))}", asm) for r in range(8 * g, 8 * g + 8)]
        body += '      asm volatile("' + "\\n ".join(lines) + '" : ' + ", ".join(f'"+{c}"(a{i})' for i in range(4)) + f' : {cons_in} : "vcc", "scc", "s20", "s21", "s22", "s23");\n'
    ident = lambda n: "".join(ch if ch.isalnum() else "_" for ch in n)
    out.append(f'''__global__ __launch_bounds__(64) void k_{ident(name)}(float* out, int iters) {{
{decl}  for (int it = 0; it < iters; ++it) {{
{body}    }}
  out[blockIdx.x * 64 + threadIdx.x] = (float)a0 + (float)a1 + (float)a2 + (float)a3;
}}
''')
out.append('''template <class Kern> double run(Kern kern, float* out, int waves, int iters) {
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  const int blocks = 256 * 4 * waves;
  hipLaunchKernelGGL(kern, dim3(blocks), dim3(64), 0, 0, out, 10); (void)hipDeviceSynchronize();
  (void)hipEventRecord(e0); hipLaunchKernelGGL(kern, dim3(blocks), dim3(64), 0, 0, out, iters); (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
  float ms; (void)hipEventElapsedTime(&ms, e0, e1);
  return ms * 1e-3 * 1024 / ((double)blocks * iters * 32.0); // SIMD-seconds per asm statement
}
int main() {
  float* out; (void)hipMalloc(&out, 1 << 24);
  const int iters = 20000;
  for (int waves : {1, 4, 8}) {
    const double ref = run(k_v_fma_f32, out, waves, iters);
    printf("--- %d waves / SIMD; v_fma_f32 = %.3f ns per wave-instruction per SIMD\\n", waves, ref * 1e9);
''')
for name, *_ in K:
    out.append(f'    {{ const double t = run(k_{"".join(ch if ch.isalnum() else "_" for ch in name)}, out, waves, iters); printf("%-20s %.3f ns  = %.2f x v_fma_f32\\n", "{name}", t * 1e9, t / ref); }}\n')
out.append("  }\n  return 0;\n}\n")
open("inst_cost.hip", "w").write("".join(out))
