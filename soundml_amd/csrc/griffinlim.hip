// Elementwise steps of Stft.griffin_lim (stft.ml:941-1017); the transforms between them are the library's own
// synthesis and analysis launches (capi.cpp strings them together on the caller's stream):
//   z      = S * angles                                   spectrum handed to the synthesis
//   angles = unit (c_k - a c_{k-1}),  unit e = e / (|e| + tiny)   (tiny: the smallest positive normal number,
//            which makes the map total: an exactly vanishing bin gives 0, not NaN -- stft.ml:957-960)
#include "smx_internal.hpp"

#include <cfloat>

namespace smx {
namespace {

template <typename T> struct Vec2;
template <> struct Vec2<float> { using type = float2; };
template <> struct Vec2<double> { using type = double2; };

template <typename T>
__global__ void __launch_bounds__(256) gl_init_kernel(const T *phase, typename Vec2<T>::type *angles, int64_t total) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    typename Vec2<T>::type a;
    if (phase) {
      a.x = (T)cos((double)phase[i]);   // the reference evaluates cos / sin in float64 (stft.ml:987-988)
      a.y = (T)sin((double)phase[i]);
    } else {
      a.x = (T)1;
      a.y = (T)0;
    }
    angles[i] = a;
  }
}

template <typename T>
__global__ void __launch_bounds__(256) gl_apply_kernel(const T *mag, const typename Vec2<T>::type *angles,
                                                       typename Vec2<T>::type *z, int64_t total) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const T m = mag[i];
    typename Vec2<T>::type a = angles[i];
    a.x *= m;
    a.y *= m;
    z[i] = a;
  }
}

template <typename T>
__global__ void __launch_bounds__(256) gl_update_kernel(const typename Vec2<T>::type *rebuilt,
                                                        const typename Vec2<T>::type *previous, T beta,
                                                        typename Vec2<T>::type *angles, int64_t total) {
  const T tiny = sizeof(T) == 8 ? (T)DBL_MIN : (T)FLT_MIN;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    typename Vec2<T>::type e = rebuilt[i];
    if (previous) {
      const typename Vec2<T>::type p = previous[i];
      e.x -= beta * p.x;
      e.y -= beta * p.y;
    }
    const T m = (T)hypot((double)e.x, (double)e.y) + tiny;
    e.x /= m;
    e.y /= m;
    angles[i] = e;
  }
}

template <typename Ta, typename Tb>
__global__ void __launch_bounds__(256) gl_convert_kernel(const Ta *src, Tb *dst, int64_t total) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) dst[i] = (Tb)src[i];
}

// [lead][bins][frames] -> [lead][rows][pitch] (frame-major, rows >= frames, pitch >= bins): Griffin-Lim's own layout for the
// fft-2048 pipelines (capi.cpp); 32 x 32 tiles through LDS, both sides coalesced
template <typename T>
__global__ void __launch_bounds__(256) gl_to_frame_major_kernel(const T *src, T *dst, int64_t bins, int64_t frames, int64_t rows, int64_t pitch) {
  __shared__ T tile[32][33];
  const int64_t clip = blockIdx.z;
  const int64_t f0 = (int64_t)blockIdx.x * 32, b0 = (int64_t)blockIdx.y * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;   // 32 x 8
  const T *s = src + clip * bins * frames;
  T *d = dst + clip * rows * pitch;
  for (int r = ty; r < 32; r += 8)
    if (b0 + r < bins && f0 + tx < frames) tile[r][tx] = s[(b0 + r) * frames + f0 + tx];
  __syncthreads();
  for (int r = ty; r < 32; r += 8)
    if (f0 + r < frames && b0 + tx < bins) d[(f0 + r) * pitch + b0 + tx] = tile[tx][r];
}

unsigned grid_for(int64_t total) { return (unsigned)std::min<int64_t>((total + 255) / 256, 16384); }

}  // namespace

void launch_gl_widen(const float *src, double *dst, int64_t total, hipStream_t stream) {
  if (total <= 0) return;
  SMX_LAUNCH((gl_convert_kernel<float, double>), dim3(grid_for(total)), dim3(256), 0, stream, src, dst, total);
  SMX_HIP_CHECK(hipGetLastError());
}
void launch_gl_narrow(const double *src, float *dst, int64_t total, hipStream_t stream) {
  if (total <= 0) return;
  SMX_LAUNCH((gl_convert_kernel<double, float>), dim3(grid_for(total)), dim3(256), 0, stream, src, dst, total);
  SMX_HIP_CHECK(hipGetLastError());
}

void launch_gl_to_frame_major(const void *src, void *dst, int64_t lead, int64_t bins, int64_t frames, int64_t rows, int64_t pitch, int elem_bytes,
                              hipStream_t stream) {
  if (lead <= 0 || bins <= 0 || frames <= 0) return;
  if (lead > 65535) throw Failure("griffin_lim: too many clips for the frame-major transposition");
  const dim3 grid((unsigned)((frames + 31) / 32), (unsigned)((bins + 31) / 32), (unsigned)lead);
  if (elem_bytes == 8)
    SMX_LAUNCH(gl_to_frame_major_kernel<float2>, grid, dim3(256), 0, stream, (const float2 *)src, (float2 *)dst, bins, frames, rows, pitch);
  else
    SMX_LAUNCH(gl_to_frame_major_kernel<float>, grid, dim3(256), 0, stream, (const float *)src, (float *)dst, bins, frames, rows, pitch);
  SMX_HIP_CHECK(hipGetLastError());
}

void launch_gl_init(const void *phase, void *angles, int64_t total, int elem_bytes, hipStream_t stream) {
  if (total <= 0) return;
  if (elem_bytes == 8)
    SMX_LAUNCH(gl_init_kernel<double>, dim3(grid_for(total)), dim3(256), 0, stream, (const double *)phase, (double2 *)angles, total);
  else
    SMX_LAUNCH(gl_init_kernel<float>, dim3(grid_for(total)), dim3(256), 0, stream, (const float *)phase, (float2 *)angles, total);
  SMX_HIP_CHECK(hipGetLastError());
}

void launch_gl_apply(const void *mag, const void *angles, void *z, int64_t total, int elem_bytes, hipStream_t stream) {
  if (total <= 0) return;
  if (elem_bytes == 8)
    SMX_LAUNCH(gl_apply_kernel<double>, dim3(grid_for(total)), dim3(256), 0, stream, (const double *)mag, (const double2 *)angles, (double2 *)z, total);
  else
    SMX_LAUNCH(gl_apply_kernel<float>, dim3(grid_for(total)), dim3(256), 0, stream, (const float *)mag, (const float2 *)angles, (float2 *)z, total);
  SMX_HIP_CHECK(hipGetLastError());
}

void launch_gl_update(const void *rebuilt, const void *previous, double beta, void *angles, int64_t total,
                      int elem_bytes, hipStream_t stream) {
  if (total <= 0) return;
  if (elem_bytes == 8)
    SMX_LAUNCH(gl_update_kernel<double>, dim3(grid_for(total)), dim3(256), 0, stream, (const double2 *)rebuilt, (const double2 *)previous, beta, (double2 *)angles, total);
  else
    SMX_LAUNCH(gl_update_kernel<float>, dim3(grid_for(total)), dim3(256), 0, stream, (const float2 *)rebuilt, (const float2 *)previous, (float)beta, (float2 *)angles, total);
  SMX_HIP_CHECK(hipGetLastError());
}

}  // namespace smx
