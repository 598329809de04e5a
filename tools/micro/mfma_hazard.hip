// MFMA dependency probe for gfx950: how many wait states does the HARDWARE need between an MFMA and a dependent / conflicting
// instruction, for the three MFMAs this library issues?  Every pattern is hand-written inline asm on fixed registers (hipcc pads
// nothing inside an asm statement), parametrised by the number N of wait states (N x `s_nop 0`) between the two instructions; a
// pattern's result is compared with the same pattern at N = 40 (far beyond any dependency).  The table it prints is what
// tools/isa_hazards.py (the listing lint run by tests/test_isa_hazards.py) checks every compiled kernel against.
//
//   build + run on the GPU box:  hipcc --offload-arch=gfx950 -O1 -o tools/micro/mfma_hazard tools/micro/mfma_hazard.hip && ./tools/micro/mfma_hazard
//
// Patterns (MF = the MFMA under test; D, A, B, C, T* = disjoint fixed register tuples):
//   raw_valu     MF D,A,B,C ; N ; v_mov R,D                 VALU read of the result                  (LLVM gfx950: passes + 4 for XDL, passes + 2 for f32)
//   waw_valu     MF D,A,B,C ; N ; v_mov D,junk              VALU write of the result registers       (same)
//   war_c        MF D,A,B,C ; N ; v_mov C,junk   (C != D)   VALU write of a SrcC the MFMA still reads (LLVM: 3 for 8-pass)
//   war_b/war_a  MF D,A,B,C ; N ; v_mov B,junk              VALU write of SrcB / SrcA                (LLVM: none)
//   chain_cd     MF T1,A,B,C ; N ; MF D,A,B,T1  (D != T1)   dependent MFMA, SrcC exactly the first's vDst, C != D  (LLVM: 0)
//   chain_inpl   MF T1,A,B,C ; N ; MF T1,A,B,T1             in-place accumulation (C == D)           (LLVM: 0, hardware interlock)
//   chain_ab     MF T1,A,B,C ; N ; MF D,T1,B,C              result as SrcA of the next MFMA          (LLVM gfx950: passes + 4 / passes + 2)
//   chain_part   MF T1,..    ; N ; MF D,A,B,{T1[2:3],X[0:1]} SrcC overlapping the first's vDst partially (LLVM gfx950: passes + 2)
//   d_on_b       MF B',A,B',C                                vDst in the registers of its own SrcB
//   d_on_a       MF A',A',B,C                                vDst in the registers of its own SrcA
//   valu_to_ab   v_mov B,val ; N ; MF D,A,B,C               VALU write -> MFMA SrcB read
//   valu_to_c    v_mov C,val ; N ; MF D,A,B,C               VALU write -> MFMA SrcC read
//   pv_chain     MF T1,A,T1,C ; MF T2,A,T2,T1 ; MF T3,A,T3,T2 ; N ; v_mov T1,junk ; v_mov T2,junk
//                (the three-link P.V chain of DESIGN.md §3e: C != D in every link, each D on its own spent B, the dead tuples rewritten N states later)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include <vector>
#include <string>

typedef unsigned u32;
struct Q { u32 x[4]; };

// register map (all clobbered): A v[100:103] B v[104:107] C v[108:111] D v[112:115] T1 v[116:119] T2 v[120:123] T3 v[124:127] R v[128:131] J v132
#define CLOB "v100","v101","v102","v103","v104","v105","v106","v107","v108","v109","v110","v111","v112","v113","v114","v115", \
             "v116","v117","v118","v119","v120","v121","v122","v123","v124","v125","v126","v127","v128","v129","v130","v131","v132","v133","v134","v135"
#define NOPS "\n .rept %c[n]\n s_nop 0\n .endr\n"
#define DRAIN "\n s_nop 15\n s_nop 15\n s_nop 15\n"

enum { I_F16 = 0, I_BF16 = 1, I_F32 = 2 };

// MF(d, a, b, c) for the three instructions: the 16x16x4 f32 MFMA takes single-register A / B (the first register of the tuple)
#define MF_F16(d, a, b, c)  " v_mfma_f32_16x16x32_f16 v[" #d ":" #d "+3], v[" #a ":" #a "+3], v[" #b ":" #b "+3], v[" #c ":" #c "+3]\n"
#define MF_BF16(d, a, b, c) " v_mfma_f32_16x16x32_bf16 v[" #d ":" #d "+3], v[" #a ":" #a "+3], v[" #b ":" #b "+3], v[" #c ":" #c "+3]\n"
#define MF_F32(d, a, b, c)  " v_mfma_f32_16x16x4_f32 v[" #d ":" #d "+3], v" #a ", v" #b ", v[" #c ":" #c "+3]\n"

#define JUNK4(r) " v_mov_b32 v[" #r "], v132\n v_mov_b32 v[" #r "+1], v132\n v_mov_b32 v[" #r "+2], v132\n v_mov_b32 v[" #r "+3], v132\n"
#define COPY4(d, s) " v_mov_b32 v[" #d "], v[" #s "]\n v_mov_b32 v[" #d "+1], v[" #s "+1]\n v_mov_b32 v[" #d "+2], v[" #s "+2]\n v_mov_b32 v[" #d "+3], v[" #s "+3]\n"

#define PROLOGUE                                                                                                          \
  " v_mov_b32 v100, %[a0]\n v_mov_b32 v101, %[a1]\n v_mov_b32 v102, %[a2]\n v_mov_b32 v103, %[a3]\n"                      \
  " v_mov_b32 v104, %[b0]\n v_mov_b32 v105, %[b1]\n v_mov_b32 v106, %[b2]\n v_mov_b32 v107, %[b3]\n"                      \
  " v_mov_b32 v108, %[c0]\n v_mov_b32 v109, %[c1]\n v_mov_b32 v110, %[c2]\n v_mov_b32 v111, %[c3]\n"                      \
  " v_mov_b32 v132, %[j]\n"                                                                                               \
  " v_mov_b32 v112, 0\n v_mov_b32 v113, 0\n v_mov_b32 v114, 0\n v_mov_b32 v115, 0\n"                                      \
  COPY4(116, 104) COPY4(120, 104) COPY4(124, 104) COPY4(128, 112) DRAIN

#define EPILOGUE(res) DRAIN " v_mov_b32 %[r0], v[" #res "]\n v_mov_b32 %[r1], v[" #res "+1]\n v_mov_b32 %[r2], v[" #res "+2]\n v_mov_b32 %[r3], v[" #res "+3]\n"

#define KERNEL(name, MF, body, res)                                                                                       \
  template <int N> __global__ __launch_bounds__(64) void name(const Q* A, const Q* B, const Q* C, Q* out, u32 junk) {                           \
    const int i = blockIdx.x * 64 + threadIdx.x;                                                                          \
    const Q a = A[i], b = B[i], c = C[i];                                                                                 \
    Q r;                                                                                                                  \
    asm volatile(PROLOGUE body EPILOGUE(res)                                                                              \
                 : [r0] "=&v"(r.x[0]), [r1] "=&v"(r.x[1]), [r2] "=&v"(r.x[2]), [r3] "=&v"(r.x[3])                         \
                 : [a0] "v"(a.x[0]), [a1] "v"(a.x[1]), [a2] "v"(a.x[2]), [a3] "v"(a.x[3]), [b0] "v"(b.x[0]), [b1] "v"(b.x[1]), \
                   [b2] "v"(b.x[2]), [b3] "v"(b.x[3]), [c0] "v"(c.x[0]), [c1] "v"(c.x[1]), [c2] "v"(c.x[2]), [c3] "v"(c.x[3]), \
                   [j] "v"(junk), [n] "i"(N)                                                                              \
                 : CLOB);                                                                                                 \
    out[i] = r;                                                                                                           \
  }

#define ALL_PATTERNS(sfx, MF)                                                                                             \
  KERNEL(raw_valu_##sfx, MF, MF(112, 100, 104, 108) NOPS COPY4(128, 112), 128)                                            \
  KERNEL(waw_valu_##sfx, MF, MF(112, 100, 104, 108) NOPS JUNK4(112), 112)                                                 \
  KERNEL(war_c_##sfx, MF, MF(112, 100, 104, 108) NOPS JUNK4(108), 112)                                                    \
  KERNEL(war_b_##sfx, MF, MF(112, 100, 104, 108) NOPS JUNK4(104), 112)                                                    \
  KERNEL(war_a_##sfx, MF, MF(112, 100, 104, 108) NOPS JUNK4(100), 112)                                                    \
  KERNEL(chain_cd_##sfx, MF, MF(116, 100, 104, 108) NOPS MF(112, 100, 104, 116), 112)                                     \
  KERNEL(chain_inpl_##sfx, MF, MF(116, 100, 104, 108) NOPS MF(116, 100, 104, 116), 116)                                   \
  KERNEL(chain_ab_##sfx, MF, MF(116, 100, 104, 108) NOPS MF(112, 116, 104, 108), 112)                                     \
  KERNEL(chain_part_##sfx, MF, MF(116, 100, 104, 108) NOPS MF(112, 100, 104, 118), 112)                                   \
  KERNEL(d_on_b_##sfx, MF, NOPS MF(116, 100, 116, 108), 116)                                                              \
  KERNEL(d_on_a_##sfx, MF, COPY4(116, 100) DRAIN NOPS MF(116, 116, 104, 108), 116)                                        \
  KERNEL(valu_to_ab_##sfx, MF, COPY4(104, 100) NOPS MF(112, 100, 104, 108), 112)                                          \
  KERNEL(valu_to_c_##sfx, MF, COPY4(108, 100) NOPS MF(112, 100, 104, 108), 112)                                           \
  KERNEL(pv_chain_##sfx, MF, MF(116, 100, 116, 108) MF(120, 100, 120, 116) MF(124, 100, 124, 120) NOPS JUNK4(116) JUNK4(120), 124)

ALL_PATTERNS(f16, MF_F16)
ALL_PATTERNS(bf16, MF_BF16)
ALL_PATTERNS(f32, MF_F32)

// mixed kinds (the f22 attention backward chains f32 and f16 MFMAs): x = v_mfma_f32_16x16x32_f16, s = v_mfma_f32_16x16x4_f32
//   cd_xs / cd_sx      first's vDst EXACTLY the second's SrcC (C != D), first x second s / first s second x
//   ab_xs / ab_sx      first's vDst as SrcA of the second
//   part_xs / part_sx  second's SrcC overlaps the first's vDst partially
//   waw_xs / waw_sx    second's vDst (SrcC = 0 registers D) overlaps the first's vDst partially, nothing reads the first: which write lands last?
//   waw_xx / waw_ss    the same with one kind
#define MIX_PATTERNS                                                                                                      \
  KERNEL(cd_xs_mix, _, MF_F16(116, 100, 104, 108) NOPS MF_F32(112, 100, 104, 116), 112)                                   \
  KERNEL(cd_sx_mix, _, MF_F32(116, 100, 104, 108) NOPS MF_F16(112, 100, 104, 116), 112)                                   \
  KERNEL(ab_xs_mix, _, MF_F16(116, 100, 104, 108) NOPS MF_F32(112, 116, 104, 108), 112)                                   \
  KERNEL(ab_sx_mix, _, MF_F32(116, 100, 104, 108) NOPS MF_F16(112, 116, 104, 108), 112)                                   \
  KERNEL(part_xs_mix, _, MF_F16(116, 100, 104, 108) NOPS MF_F32(112, 100, 104, 118), 112)                                 \
  KERNEL(part_sx_mix, _, MF_F32(116, 100, 104, 108) NOPS MF_F16(112, 100, 104, 118), 112)                                 \
  KERNEL(waw_xs_mix, _, MF_F16(116, 100, 104, 108) NOPS MF_F32(118, 100, 104, 112), 118)                                  \
  KERNEL(waw_sx_mix, _, MF_F32(116, 100, 104, 108) NOPS MF_F16(118, 100, 104, 112), 118)                                  \
  KERNEL(waw_xx_mix, _, MF_F16(116, 100, 104, 108) NOPS MF_F16(118, 100, 104, 112), 118)                                  \
  KERNEL(waw_ss_mix, _, MF_F32(116, 100, 104, 108) NOPS MF_F32(118, 100, 104, 112), 118)
MIX_PATTERNS

typedef void (*kern_t)(const Q*, const Q*, const Q*, Q*, u32);
constexpr int NMAX = 24, NREF = 40;

#define TABLE(name)                                                                                                       \
  static kern_t name##_tab(int n) {                                                                                       \
    switch (n) {                                                                                                          \
      case 0: return name<0>; case 1: return name<1>; case 2: return name<2>; case 3: return name<3>; case 4: return name<4>; \
      case 5: return name<5>; case 6: return name<6>; case 7: return name<7>; case 8: return name<8>; case 9: return name<9>; \
      case 10: return name<10>; case 11: return name<11>; case 12: return name<12>; case 13: return name<13>;             \
      case 14: return name<14>; case 15: return name<15>; case 16: return name<16>; case 17: return name<17>;             \
      case 18: return name<18>; case 19: return name<19>; case 20: return name<20>; case 21: return name<21>;             \
      case 22: return name<22>; case 23: return name<23>; case 24: return name<24>; default: return name<NREF>;           \
    }                                                                                                                     \
  }
#define TABLES(sfx)                                                                                                       \
  TABLE(raw_valu_##sfx) TABLE(waw_valu_##sfx) TABLE(war_c_##sfx) TABLE(war_b_##sfx) TABLE(war_a_##sfx) TABLE(chain_cd_##sfx) \
  TABLE(chain_inpl_##sfx) TABLE(chain_ab_##sfx) TABLE(chain_part_##sfx) TABLE(d_on_b_##sfx) TABLE(d_on_a_##sfx)           \
  TABLE(valu_to_ab_##sfx) TABLE(valu_to_c_##sfx) TABLE(pv_chain_##sfx)
TABLES(f16)
TABLES(bf16)
TABLES(f32)
TABLE(cd_xs_mix) TABLE(cd_sx_mix) TABLE(ab_xs_mix) TABLE(ab_sx_mix) TABLE(part_xs_mix) TABLE(part_sx_mix)
TABLE(waw_xs_mix) TABLE(waw_sx_mix) TABLE(waw_xx_mix) TABLE(waw_ss_mix)

struct Pat { const char* name; kern_t (*tab)(int); };

static u32 f2h2(float lo, float hi) {
  _Float16 a = (_Float16)lo, b = (_Float16)hi;
  unsigned short ua, ub; memcpy(&ua, &a, 2); memcpy(&ub, &b, 2);
  return (u32)ua | ((u32)ub << 16);
}
static u32 f2bf2(float lo, float hi) {
  u32 a, b; memcpy(&a, &lo, 4); memcpy(&b, &hi, 4);
  return (a >> 16) | (b & 0xffff0000u);
}

int main() {
  const int blocks = 256, n = blocks * 64;      // one wave per CU: nothing else on the SIMD hides a missing wait state
  std::vector<Q> hA(n), hB(n), hC(n), ref(n), got(n);
  Q *dA, *dB, *dC, *dO;
  hipMalloc(&dA, n * sizeof(Q)); hipMalloc(&dB, n * sizeof(Q)); hipMalloc(&dC, n * sizeof(Q)); hipMalloc(&dO, n * sizeof(Q));
  const char* inames[4] = {"v_mfma_f32_16x16x32_f16", "v_mfma_f32_16x16x32_bf16", "v_mfma_f32_16x16x4_f32",
                           "mixed: x = v_mfma_f32_16x16x32_f16, s = v_mfma_f32_16x16x4_f32 (f16-formatted operands; the f32 MFMA reads their first register as a float)"};
  for (int instr = 0; instr < 4; ++instr) {
    unsigned s = 12345u + instr;
    auto rnd = [&]() { s = s * 1664525u + 1013904223u; return ((s >> 8) & 0xffff) / 65536.f - 0.5f; };
    for (int i = 0; i < n; ++i)
      for (int k = 0; k < 4; ++k) {
        float a0 = rnd(), a1 = rnd(), b0 = rnd(), b1 = rnd(), c = rnd();
        if (instr == I_F16 || instr == 3) { hA[i].x[k] = f2h2(a0, a1); hB[i].x[k] = f2h2(b0, b1); }
        else if (instr == I_BF16) { hA[i].x[k] = f2bf2(a0, a1); hB[i].x[k] = f2bf2(b0, b1); }
        else { memcpy(&hA[i].x[k], &a0, 4); memcpy(&hB[i].x[k], &b0, 4); }
        memcpy(&hC[i].x[k], &c, 4);
      }
    hipMemcpy(dA, hA.data(), n * sizeof(Q), hipMemcpyHostToDevice);
    hipMemcpy(dB, hB.data(), n * sizeof(Q), hipMemcpyHostToDevice);
    hipMemcpy(dC, hC.data(), n * sizeof(Q), hipMemcpyHostToDevice);
    u32 junk; { float j = 777.25f; if (instr == I_F16 || instr == 3) junk = f2h2(j, j); else if (instr == I_BF16) junk = f2bf2(j, j); else memcpy(&junk, &j, 4); }
#define P(name, sfx) Pat{#name, name##_##sfx##_tab}
#define PATS(sfx) {P(raw_valu, sfx), P(waw_valu, sfx), P(war_c, sfx), P(war_b, sfx), P(war_a, sfx), P(chain_cd, sfx), P(chain_inpl, sfx), \
                   P(chain_ab, sfx), P(chain_part, sfx), P(d_on_b, sfx), P(d_on_a, sfx), P(valu_to_ab, sfx), P(valu_to_c, sfx), P(pv_chain, sfx)}
    std::vector<Pat> pats;
    if (instr == I_F16) pats = PATS(f16); else if (instr == I_BF16) pats = PATS(bf16); else if (instr == I_F32) pats = PATS(f32);
    else pats = {P(cd_xs, mix), P(cd_sx, mix), P(ab_xs, mix), P(ab_sx, mix), P(part_xs, mix), P(part_sx, mix), P(waw_xs, mix), P(waw_sx, mix),
                 P(waw_xx, mix), P(waw_ss, mix)};
    printf("== %s\n", inames[instr]);
    for (const Pat& p : pats) {
      hipLaunchKernelGGL(p.tab(NREF), dim3(blocks), dim3(64), 0, 0, dA, dB, dC, dO, junk);
      hipMemcpy(ref.data(), dO, n * sizeof(Q), hipMemcpyDeviceToHost);
      std::string line; int min_safe = 0; long worst = 0;
      for (int N = 0; N <= NMAX; ++N) {
        long bad = 0;
        for (int rep = 0; rep < 3; ++rep) {
          hipLaunchKernelGGL(p.tab(N), dim3(blocks), dim3(64), 0, 0, dA, dB, dC, dO, junk);
          hipMemcpy(got.data(), dO, n * sizeof(Q), hipMemcpyDeviceToHost);
          for (int i = 0; i < n; ++i) if (memcmp(&got[i], &ref[i], sizeof(Q)) != 0) ++bad;
        }
        line += bad ? 'X' : '.';
        if (bad) { min_safe = N + 1; if (bad > worst) worst = bad; }
      }
      printf("%-12s N=0..%d %s   min safe wait states %d%s\n", p.name, NMAX, line.c_str(), min_safe, min_safe > NMAX ? " (or more)" : "");
    }
  }
  if (hipDeviceSynchronize() != hipSuccess) { printf("HIP error\n"); return 1; }
  return 0;
}
