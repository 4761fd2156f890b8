#!/usr/bin/env python3
"""Generates mix_bound.hip: the VALU issue bound of the million-voice window, MEASURED.

docs/VALU_COSTS.md prices instruction classes one at a time (a loop of one instruction); summed over the window's measured
class mix those prices exceed the SIMD time the step really had (1.00 - 1.10 of it), so they are not a
bound — different classes overlap in ways a single-instruction loop cannot show.  This kernel issues the MIX itself: per 100
VALU instructions the shares `profiles/r03_welsh-1m-window_summary.json` measured with SQ_INSTS_VALU_* (f64 20.8 %,
conversions 9.3 %, transcendental 2.1 %, 64-bit integer 4.3 %, 32-bit integer 2.6 %, f32 add / mul / fma 44.6 % — 19 % of
them with an SGPR operand, the static share in the class-specialised bodies' text —, everything else 16.4 %: moves,
selects, compares), interleaved evenly, as INDEPENDENT chains (eight accumulators per class, so no instruction waits for
the one before it), at the occupancies the real kernels run at (5 / 5 / 4 / 4 waves per SIMD through round 5; all 5 since `profiles/r05_budgets_ab.log`) and at 8.  What it prints —
ns of SIMD time per wave-instruction of this mix — times the step's measured instruction count is the time below which no
schedule of this instruction stream can finish: bench.py reports step time against it as valu.frac_of_measured_bound.

    python3 tools/micro/gen_mix_bound.py && hipcc --offload-arch=gfx950 -O3 -o /tmp/mix_bound tools/micro/mix_bound.hip && /tmp/mix_bound
"""
import os
import re

# (class, asm, accumulator type, [(constraint, type, value)], count per 100 in round 4, ... in round 5, ... in round 6 before the LFO look-ahead (MIX_ROUND=60), ... at the end of round 6)
# Round 5 (profiles/r05_welsh-1m-window_summary.json: 20 (19 since the last build) of the 32 patches run the filter in fp32, docs/DSP_SPEC.md section 11):
# 3.69e8 -> 3.37e8 wave-instructions per step, of which f64 20.8 -> 12.4 %, conversions 9.3 -> 6.3 %, fp32 add / mul / fma 44.6 -> 54.6 %,
# everything else 16.4 -> 16.9 %, transcendental 2.1 -> 2.3 %, 64-bit integer 4.3 -> 4.7 %, 32-bit integer 2.6 -> 2.8 % (the fp32 step in
# three-operand assembly and the polynomial envelopes took instructions out; the shares of what stayed grew).
_MIX_BOTH = [
    ("f64_add", "v_add_f64 %0, %1, %0", "double", [("v", "double", "1.0001")], 6, 3, 3, 2),
    ("f64_mul", "v_mul_f64 %0, %1, %0", "double", [("v", "double", "1.0001")], 4, 3, 4, 3),
    ("f64_fma", "v_fma_f64 %0, %1, %2, %0", "double", [("v", "double", "1.0001"), ("v", "double", "0.5")], 11, 6, 8, 6),
    ("cvt_f64_f32", "v_cvt_f64_f32 %0, %1", "double", [("v", "float", "1.5f")], 3, 2, 3, 3),
    ("cvt_f32_f64", "v_cvt_f32_f64 %0, %1", "float", [("v", "double", "1.5")], 3, 2, 2, 2),
    ("cvt_f32_u32", "v_cvt_f32_u32 %0, %1", "float", [("v", "unsigned", "3u")], 3, 2, 2, 3),
    ("trans_exp", "v_exp_f32 %0, %1", "float", [("v", "float", "0.5f")], 1, 1, 0, 0),
    ("trans_rcp", "v_rcp_f32 %0, %1", "float", [("v", "float", "1.5f")], 1, 1, 1, 0),
    ("int64", "v_lshl_add_u64 %0, %0, 0, %1", "unsigned long long", [("v", "unsigned long long", "3ull")], 4, 5, 7, 8),
    ("int32", "v_add_u32 %0, %1, %0", "unsigned", [("v", "unsigned", "3u")], 3, 3, 4, 6),
    ("f32_add", "v_add_f32 %0, %1, %0", "float", [("v", "float", "1.0001f")], 10, 12, 10, 10),
    ("f32_mul", "v_mul_f32 %0, %1, %0", "float", [("v", "float", "1.0001f")], 14, 15, 11, 12),
    ("f32_fma", "v_fma_f32 %0, %1, %2, %0", "float", [("v", "float", "1.0001f"), ("v", "float", "0.5f")], 12, 18, 15, 15),
    ("f32_add_s", "v_add_f32 %0, %1, %0", "float", [("s", "float", "1.0001f")], 2, 2, 2, 3),
    ("f32_mul_s", "v_mul_f32 %0, %1, %0", "float", [("s", "float", "1.0001f")], 4, 4, 3, 3),
    ("f32_fma_s", "v_fma_f32 %0, %1, %2, %0", "float", [("s", "float", "1.0001f"), ("v", "float", "0.5f")], 3, 4, 4, 3),
    ("mov", "v_mov_b32 %0, %1", "float", [("v", "float", "1.5f")], 8, 8, 8, 8),
    ("cndmask", "v_cndmask_b32 %0, %1, %0, vcc", "unsigned", [("v", "unsigned", "3u")], 4, 4, 6, 6),
    ("cmp", "v_cmp_lt_u32 vcc, %0, %1", "unsigned", [("v", "unsigned", "3u")], 4, 5, 7, 7),
]
# Round 6 (profiles/r06_welsh-1m-window_summary.json: the coefficient look-ahead took the retune out of most frames — 3.39e8 -> 2.55e8
# wave-instructions per step): transcendental 2.3 -> 0.4 %, f64 12.4 -> 15.0 %, conversions 6.3 -> 7.1 %, 64-bit integer 4.7 -> 7.3 %,
# 32-bit integer 2.8 -> 3.8 %, fp32 add / mul / fma 54.6 -> 45.3 %, everything else 16.9 -> 21.1 % (the shares of what stayed grew).
# End of round 6 (the same file, retaken: the LFO look-ahead and the table frames' own loop — 2.34e8 -> 2.05e8 wave-instructions per step): f64 16.3 -> 11.2 %,
# conversions 7.8 -> 8.1 %, transcendental 0.4 -> 0.1 %, 64-bit integer 8.0 -> 7.6 %, 32-bit integer 4.1 -> 6.0 %, fp32 add / mul / fma 44.7 -> 45.5 %, everything else 18.7 -> 21.5 %.
ROUND = int(os.environ.get("MIX_ROUND", "6"))   # MIX_ROUND=4 / 5 / 60 regenerate the earlier kernels (profiles/r04_mix_bound.json, r05_, round 6 before the LFO look-ahead)
MIX = [m[:4] + (m[4] if ROUND == 4 else m[5] if ROUND == 5 else m[6] if ROUND == 60 else m[7],) for m in _MIX_BOTH]
# MIX_NO_SGPR=1: the same mix with every SGPR operand of an fp32 instruction replaced by a VGPR (what the bound would be if the kernels
# kept their wave-uniform constants in vector registers; written to mix_bound_nosgpr.hip)
NO_SGPR = os.environ.get("MIX_NO_SGPR", "0") == "1"
if NO_SGPR:
    MIX = [m[:3] + ([("v",) + i[1:] for i in m[3]],) + m[4:] for m in MIX]
assert sum(m[4] for m in MIX) == 100
ACC = 2  # independent chains per class (eight need 183 VGPRs: two waves per SIMD whatever the grid — measured, round 4 — and the figure then describes that occupancy, not the one asked for)


def schedule():
    """The 100 instructions in an even interleave: instruction j of a class with n instructions sits at (j + 0.5) / n."""
    slots = []
    for ci, (_, _, _, _, n) in enumerate(MIX):
        slots += [((j + 0.5) / n, ci, j) for j in range(n)]
    return [(ci, j) for _, ci, j in sorted(slots)]


def kernel(name, chains):
    out = [f"__global__ __launch_bounds__(64) void {name}(float* out, int iters) {{"]
    for ci, (cls, _, ctype, ins, _) in enumerate(MIX):
        for a in range(chains):
            out.append(f"  {ctype} a{ci}_{a} = ({ctype})(threadIdx.x) + {a + 1};")
        for k, (_, t, v) in enumerate(ins):
            out.append(f"  {t} b{ci}_{k} = {v};")
    out.append("  for (int it = 0; it < iters; ++it) {")
    use = [0] * len(MIX)
    seq = []
    for rep in range(2):  # two copies of the 100 per trip: the loop's own scalar instructions stay under 2 %
        for ci, _ in schedule():
            seq.append((ci, f"a{ci}_{use[ci] % chains}"))
            use[ci] += 1
    # ten instructions per asm statement: the compiler pads every inline-asm statement with an s_nop (it cannot see
    # inside), one per instruction would put a scalar instruction between every two vector ones
    for g in range(0, len(seq), 10):
        outs, ins_, text = [], [], []
        for ci, acc in seq[g:g + 10]:
            _, asm, _, ins, _ = MIX[ci]
            if acc not in outs:
                outs.append(acc)
            for k, (c, _, _) in enumerate(ins):
                if (c, f"b{ci}_{k}") not in ins_:
                    ins_.append((c, f"b{ci}_{k}"))
        for ci, acc in seq[g:g + 10]:
            _, asm, _, ins, _ = MIX[ci]
            idx = [outs.index(acc)] + [len(outs) + ins_.index((c, f"b{ci}_{k}")) for k, (c, _, _) in enumerate(ins)]
            text.append(re.sub(r"%(\d)", lambda m: f"%{idx[int(m.group(1))]}", asm))
        body = "\\n\\t".join(text)
        out.append(f'    asm volatile("{body}" : ' + ", ".join(f'"+v"({a})' for a in outs) + " : " + ", ".join(f'"{c}"({b})' for c, b in ins_) + ' : "vcc");')
    out.append("  }")
    out.append("  float r = 0.0f;")
    for ci in range(len(MIX)):
        for a in range(chains):
            out.append(f"  r += (float)a{ci}_{a};")
    out.append("  out[blockIdx.x * 64 + threadIdx.x] = r;")
    out.append("}")
    return "\n".join(out)


MIXJSON = ", ".join('\\"' + m[0] + '\\": ' + str(m[4]) for m in MIX).replace('\\\\', '\\')
SRC = f'''// GENERATED by tools/micro/gen_mix_bound.py (see its header): the measured VALU class mix of the million-voice window issued
// as independent chains, at the occupancies the render kernels run at.  Prints ns of SIMD time per wave-instruction.
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/mix_bound tools/micro/mix_bound.hip && /tmp/mix_bound
#include <hip/hip_runtime.h>
#include <cstdio>
{kernel("mix_independent", ACC)}
{kernel("mix_one_chain_per_class", 1)}
template <class Kern> double run(Kern kern, float* out, int waves, int iters) {{
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  const int blocks = 256 * 4 * waves; // one 64-lane workgroup per wave: `waves` per SIMD on all 1,024 SIMDs
  hipLaunchKernelGGL(kern, dim3(blocks), dim3(64), 0, 0, out, 10); (void)hipDeviceSynchronize();
  double best = 1e30;
  for (int rep = 0; rep < 5; ++rep) {{
    (void)hipEventRecord(e0); hipLaunchKernelGGL(kern, dim3(blocks), dim3(64), 0, 0, out, iters); (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    const double t = ms * 1e-3 * 1024 / ((double)blocks * iters * 200.0); // SIMD-seconds per wave-instruction
    if (t < best) best = t;
  }}
  return best;
}}
int main() {{
  float* out; (void)hipMalloc(&out, 1 << 24);
  const int iters = 4000;
  printf("{{\\"mix_per_100\\": {{{MIXJSON}}}, \\"independent_chains_per_class\\": {ACC}, \\"ns_per_wave_instruction\\": {{");
  bool first = true;
  for (int waves : {{1, 2, 4, 5, 8}}) {{
    const double a = run(mix_independent, out, waves, iters), b = run(mix_one_chain_per_class, out, waves, iters);
    printf("%s\\"%d\\": {{\\"independent\\": %.4f, \\"one_chain_per_class\\": %.4f}}", first ? "" : ", ", waves, a * 1e9, b * 1e9);
    first = false;
  }}
  printf("}}}}\\n");
  return 0;
}}
'''
open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "mix_bound_nosgpr.hip" if NO_SGPR else "mix_bound.hip"), "w").write(SRC)
