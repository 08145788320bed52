// sstore_probe.hip -- does gfx950 execute SCALAR stores (s_store_dwordx2 + s_dcache_wb), and does the data SGPR pair have to
// stay untouched until the store has completed?  Background (CHANGELOG.md, round 4, "ranked for a next round"): LeakyReLU
// masks as 64-bit LANE masks -- one v_cmp per value register in the training forward, stored from the scalar unit, applied in
// the dgrad chain with v_cndmask ... s[n:n+1] -- would take the dgrad chain's mask work from 3.5 to 2.5 VALU instructions per
// value and drop its activation re-reads.  This probe answers the two hardware questions that design rests on.
//   hipcc --offload-arch=gfx950 -O3 tools/micro/sstore_probe.hip -o /tmp/sstore_probe && /tmp/sstore_probe
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

template <bool WAIT_EACH>
__global__ void __launch_bounds__(256) store_masks(const float* __restrict__ x, uint64_t* __restrict__ out, int nreg) {
  const int lane = threadIdx.x & 63;
  const long wave = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  const float* xw = x + wave * nreg * 64;
  unsigned long o = (unsigned long)(out + wave * nreg);
  o = ((unsigned long)(unsigned)__builtin_amdgcn_readfirstlane((int)(o >> 32)) << 32) | (unsigned)__builtin_amdgcn_readfirstlane((int)o);
#pragma unroll 8
  for (int i = 0; i < nreg; ++i) {
    const float v = xw[(long)i * 64 + lane];
    const uint64_t m = __builtin_amdgcn_ballot_w64(v > 0.f);          // v_cmp_gt_f32 into an SGPR pair
    const int off = i * 8;
    asm volatile("s_store_dwordx2 %0, %1, %2" ::"s"(m), "s"(o), "s"(off) : "memory");
    if (WAIT_EACH) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  }
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_dcache_wb\n\ts_waitcnt lgkmcnt(0)" ::: "memory");
}

// the reader the dgrad chain would be: masks through the scalar cache, select with an SGPR-pair condition
__global__ void __launch_bounds__(256) apply_masks(const float* __restrict__ x, const uint64_t* __restrict__ masks, float* __restrict__ y, int nreg) {
  const int lane = threadIdx.x & 63;
  const long wave = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  const float* xw = x + wave * nreg * 64;
  float* yw = y + wave * nreg * 64;
  unsigned long o = (unsigned long)(masks + wave * nreg);
  o = ((unsigned long)(unsigned)__builtin_amdgcn_readfirstlane((int)(o >> 32)) << 32) | (unsigned)__builtin_amdgcn_readfirstlane((int)o);
  typedef const uint64_t __attribute__((address_space(4))) * cmask_ptr;   // constant address space + uniform address = s_load
  cmask_ptr mw = (cmask_ptr)o;
  for (int i = 0; i < nreg; ++i) {
    const float v = xw[(long)i * 64 + lane];
    const uint64_t m = mw[i];                                          // uniform address: s_load_dwordx2
    float r;
    asm volatile("v_cndmask_b32_e64 %0, %1, %2, %3" : "=v"(r) : "v"(0.01f * v), "v"(v), "s"(m));
    yw[(long)i * 64 + lane] = r;
  }
}

int main() {
  const int nreg = 128, waves = 4 * 2048;
  const long n = (long)waves * nreg * 64;
  std::vector<float> hx(n);
  unsigned s = 12345u;
  for (long i = 0; i < n; ++i) { s = s * 1664525u + 1013904223u; hx[i] = ((int)(s >> 8) % 2001 - 1000) * 1e-3f; }
  float *dx, *dy; uint64_t* dm;
  hipMalloc(&dx, n * 4); hipMalloc(&dy, n * 4); hipMalloc(&dm, (long)waves * nreg * 8);
  hipMemcpy(dx, hx.data(), n * 4, hipMemcpyHostToDevice);
  std::vector<uint64_t> hm((long)waves * nreg);
  std::vector<float> hy(n);
  for (int variant = 0; variant < 2; ++variant) {
    hipMemset(dm, 0xff, (long)waves * nreg * 8);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    if (variant == 0) hipLaunchKernelGGL(store_masks<true>, dim3(waves / 4), dim3(256), 0, 0, dx, dm, nreg);
    else hipLaunchKernelGGL(store_masks<false>, dim3(waves / 4), dim3(256), 0, 0, dx, dm, nreg);
    hipEventRecord(e1);
    hipError_t err = hipDeviceSynchronize();
    float ms = 0; hipEventElapsedTime(&ms, e0, e1);
    hipMemcpy(hm.data(), dm, (long)waves * nreg * 8, hipMemcpyDeviceToHost);
    long bad = 0;
    for (long w = 0; w < waves; ++w)
      for (int i = 0; i < nreg; ++i) {
        uint64_t ref = 0;
        for (int l = 0; l < 64; ++l) ref |= (uint64_t)(hx[(w * nreg + i) * 64 + l] > 0.f) << l;
        bad += ref != hm[w * nreg + i];
      }
    printf("s_store_dwordx2, %s: %s, %ld of %ld masks wrong, %.3f ms (%.1f M stores/s)\n",
           variant == 0 ? "s_waitcnt lgkmcnt(0) after every store" : "no wait between stores (data SGPRs reused at once)",
           hipGetErrorString(err), bad, (long)waves * nreg, ms, (double)waves * nreg / ms * 1e-3);
    if (variant == 0 && bad == 0) {
      hipLaunchKernelGGL(apply_masks, dim3(waves / 4), dim3(256), 0, 0, dx, dm, dy, nreg);
      err = hipDeviceSynchronize();
      hipMemcpy(hy.data(), dy, n * 4, hipMemcpyDeviceToHost);
      long badv = 0;
      for (long i = 0; i < n; ++i) badv += hy[i] != (hx[i] > 0.f ? hx[i] : 0.01f * hx[i]);
      printf("s_load_dwordx2 + v_cndmask_b32 with an SGPR-pair mask: %s, %ld of %ld values wrong\n", hipGetErrorString(err), badv, n);
    }
  }
  return 0;
}
