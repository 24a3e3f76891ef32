// Probe: issue cost (cycles per wave-instruction per SIMD) of the float64 vector instructions the float64 interior runs on,
// at 1 and 2 waves per SIMD.  Each test runs R x 64 instructions on 16 independent registers (s_memtime counts at 100 MHz:
// the table is scaled by the ratio to the v_add_f32 row, whose cost is known: tools/probes/issue_probe.hip, 2.07 cycles).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
template <int T>
__global__ void __launch_bounds__(512) k(double *out, unsigned long long *cyc, int reps, double s) {
  double v[16];
  float f[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) { v[i] = threadIdx.x * 0.001 + i + 1.0; f[i] = (float)v[i]; }
  __syncthreads();
  unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
  for (int r = 0; r < reps; ++r) {
#pragma unroll
    for (int u = 0; u < 4; ++u) {
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        if (T == 0) asm volatile("v_fma_f64 %0, %0, %1, %0" : "+v"(v[i]) : "v"(s));
        if (T == 1) asm volatile("v_add_f64 %0, %0, %1" : "+v"(v[i]) : "v"(s));
        if (T == 2) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(v[i]) : "v"(s));
        if (T == 3) asm volatile("v_add_f32 %0, %0, %1" : "+v"(f[i]) : "v"((float)s));
        if (T == 4) asm volatile("v_cvt_f64_f32 %0, %1" : "=v"(v[i]) : "v"(f[i]));
        if (T == 5) asm volatile("v_cvt_f32_f64 %0, %1" : "=v"(f[i]) : "v"(v[i]));
        if (T == 6) asm volatile("v_rsq_f64 %0, %0" : "+v"(v[i]));
        if (T == 7) asm volatile("v_sqrt_f64 %0, %0" : "+v"(v[i]));
        if (T == 8) v[i] = sqrt(v[i]);       // the compiler's IEEE expansion (counted as one)
        if (T == 9) asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(f[i]) : "v"((float)s));
      }
    }
  }
  __syncthreads();
  unsigned long long t1 = __builtin_amdgcn_s_memrealtime();
  double acc = 0;
#pragma unroll
  for (int i = 0; i < 16; ++i) acc += v[i] + f[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
template <int T>
double run(const char *name, int threads, double ref) {
  double *out; unsigned long long *cyc;
  hipMalloc(&out, 256 * 512 * 8); hipMalloc(&cyc, 256 * 8);
  const int reps = 200;
  hipLaunchKernelGGL(k<T>, dim3(256), dim3(threads), 0, 0, out, cyc, reps, 1.0000001);
  hipDeviceSynchronize();
  std::vector<unsigned long long> h(256);
  hipMemcpy(h.data(), cyc, 256 * 8, hipMemcpyDeviceToHost);
  double mean = 0; for (auto c : h) mean += c; mean /= 256;
  const double per = mean / (reps * 64.0) / (threads / 256);   // 100 MHz ticks per wave-instruction per SIMD
  printf("%-34s %4d thr: %.4f ticks per wave-instr per SIMD%s\n", name, threads, per, ref > 0 ? "" : "  (reference row)");
  if (ref > 0) printf("%-34s          = %.2f cycles at 2.07 per v_add_f32\n", "", per / ref * 2.07);
  hipFree(out); hipFree(cyc);
  return per;
}
int main() {
  for (int threads : {256, 512}) {
    const double ref = run<3>("v_add_f32", threads, 0);
    run<9>("v_fma_f32", threads, ref);
    run<0>("v_fma_f64", threads, ref); run<1>("v_add_f64", threads, ref); run<2>("v_mul_f64", threads, ref);
    run<4>("v_cvt_f64_f32", threads, ref); run<5>("v_cvt_f32_f64", threads, ref);
    run<6>("v_rsq_f64", threads, ref); run<7>("v_sqrt_f64", threads, ref); run<8>("sqrt(double), IEEE expansion", threads, ref);
  }
  return 0;
}
