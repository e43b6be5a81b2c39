// Energy-ledger build (tools/ledger.sh): every f16 MFMA of the conv engine becomes an empty asm statement that keeps the
// register dependencies — all data movement, address arithmetic, barriers and epilogues stay, the matrix pipe idles.
// Injected with `FCP_BUILD_FLAGS="-include tools/probes/fcp_no_mfma.h"`; results are garbage by construction.
#pragma once
template <class A, class B, class C>
__attribute__((device)) __attribute__((always_inline)) inline C fcp_nop_mfma16(A a, B b, C c) {
  asm volatile("" : "+v"(c) : "v"(a), "v"(b));
  return c;
}
#define __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, x, y, z) fcp_nop_mfma16((a), (b), (c))
