// Probe: operand / result lane layout of v_mfma_f32_4x4x1_16B_f32 on gfx950 (diagnostic).
// Expectation: 16 blocks of 4 lanes; A: lane 4b+i holds A_b[i]; B: lane 4b+j holds B_b[j]; D: lane 4b+j, vgpr i = D_b[i][j].
#include <hip/hip_runtime.h>
#include <cstdio>
using f4 = __attribute__((ext_vector_type(4))) float;
__global__ void k(const float *a, const float *b, float *d) {
  const int l = threadIdx.x;
  f4 acc = {0, 0, 0, 0};
  acc = __builtin_amdgcn_mfma_f32_4x4x1f32(a[l], b[l], acc, 0, 0, 0);
  for (int r = 0; r < 4; ++r) d[l * 4 + r] = acc[r];
}
int main() {
  float ha[64], hb[64], hd[256], *da, *db, *dd;
  for (int l = 0; l < 64; ++l) { ha[l] = 1.0f + l; hb[l] = 100.0f + 3 * l; }
  (void)hipMalloc(&da, sizeof ha); (void)hipMalloc(&db, sizeof hb); (void)hipMalloc(&dd, sizeof hd);
  (void)hipMemcpy(da, ha, sizeof ha, hipMemcpyHostToDevice); (void)hipMemcpy(db, hb, sizeof hb, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, da, db, dd);
  (void)hipMemcpy(hd, dd, sizeof hd, hipMemcpyDeviceToHost);
  int bad = 0;
  for (int l = 0; l < 64; ++l)
    for (int r = 0; r < 4; ++r) {
      const int blk = l / 4, j = l % 4;
      const float want = ha[4 * blk + r] * hb[4 * blk + j];
      if (hd[l * 4 + r] != want) { if (bad < 5) printf("lane %d reg %d: got %g want %g\n", l, r, hd[l * 4 + r], want); ++bad; }
    }
  printf("mfma_f32_4x4x1_16B layout: %d mismatches (0 = as expected)\n", bad);
  return 0;
}
