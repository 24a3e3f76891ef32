// Do an upload and a download share the link or serialise?  (round 5: torch's two-stream copies of 0.49 GB up + 0.98 GB down
// finish at 8.6 / 25.8 ms = one after the other.)  The same two transfers between page-locked host memory and the device with
// the DMA engine (hipMemcpyAsync) and with copy KERNELS that read / write the mapped host memory from the shader engines.
//   hipcc -O3 --offload-arch=gfx950 -o tools/probes/pcie_duplex_probe tools/probes/pcie_duplex_probe.hip && tools/probes/pcie_duplex_probe
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
__global__ void __launch_bounds__(256) copy_kernel(float4 *dst, const float4 *src, size_t n) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) dst[i] = src[i];
}
static double now() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main() {
  const size_t up = (size_t)256 * 480000 * 4, down = (size_t)256 * 1025 * 938 * 4;
  float *hx, *hy, *dx, *dy;
  CK(hipHostMalloc((void **)&hx, up, hipHostMallocPortable));
  CK(hipHostMalloc((void **)&hy, down, hipHostMallocPortable));
  CK(hipMalloc((void **)&dx, up));
  CK(hipMalloc((void **)&dy, down));
  for (size_t i = 0; i < up / 4; ++i) hx[i] = (float)i;
  CK(hipMemset(dy, 1, down));
  hipStream_t s1, s2;
  CK(hipStreamCreateWithFlags(&s1, hipStreamNonBlocking));
  CK(hipStreamCreateWithFlags(&s2, hipStreamNonBlocking));
  auto run = [&](const char *name, int up_mode, int down_mode, int blocks) {   // 0: none, 1: DMA, 2: kernel
    printf("%-58s", name);
    for (int rep = 0; rep < 5; ++rep) {
      (void)hipDeviceSynchronize();
      const double t0 = now();
      if (up_mode == 1) (void)hipMemcpyAsync(dx, hx, up, hipMemcpyHostToDevice, s1);
      if (up_mode == 2) hipLaunchKernelGGL(copy_kernel, dim3(blocks), dim3(256), 0, s1, (float4 *)dx, (const float4 *)hx, up / 16);
      if (down_mode == 1) (void)hipMemcpyAsync(hy, dy, down, hipMemcpyDeviceToHost, s2);
      if (down_mode == 2) hipLaunchKernelGGL(copy_kernel, dim3(blocks), dim3(256), 0, s2, (float4 *)hy, (const float4 *)dy, down / 16);
      double tu = 0, td = 0;
      if (up_mode) { (void)hipStreamSynchronize(s1); tu = now() - t0; }
      if (down_mode) { (void)hipStreamSynchronize(s2); td = now() - t0; }
      printf("  %5.1f / %5.1f", tu, td);
    }
    printf("\n");
    return 0;
  };
  printf("0.49 GB up, 0.98 GB down; ms at which the upload / the download is complete, 5 repetitions\n");
  run("upload alone, DMA", 1, 0, 0);
  run("download alone, DMA", 0, 1, 0);
  run("both, DMA + DMA", 1, 1, 0);
  for (int blocks : {256}) {
    char name[96];
    snprintf(name, sizeof name, "upload alone, kernel (%d blocks)", blocks);
    run(name, 2, 0, blocks);
    snprintf(name, sizeof name, "download alone, kernel (%d blocks)", blocks);
    run(name, 0, 2, blocks);
    snprintf(name, sizeof name, "both, upload by kernel (%d blocks) + download by DMA", blocks);
    run(name, 2, 1, blocks);
    snprintf(name, sizeof name, "both, upload by DMA + download by kernel (%d blocks)", blocks);
    run(name, 1, 2, blocks);
  }
  // the library's shape: the transfers in pieces (hipMemcpyAsync each), at most `inflight` upload pieces enqueued at a time
  auto pieces = [&](size_t up_piece, size_t down_piece, int inflight) {
    printf("upload in %3zu MB pieces (%2d in flight), download in %4zu MB pieces:", up_piece >> 20, inflight, down_piece >> 20);
    const int nup = (int)((up + up_piece - 1) / up_piece);
    hipEvent_t ev[64];
    for (auto &e : ev) (void)hipEventCreateWithFlags(&e, hipEventDisableTiming);
    for (int rep = 0; rep < 4; ++rep) {
      (void)hipDeviceSynchronize();
      const double t0 = now();
      for (size_t off = 0; off < down; off += down_piece)
        (void)hipMemcpyAsync((char *)hy + off, (char *)dy + off, down - off < down_piece ? down - off : down_piece, hipMemcpyDeviceToHost, s2);
      for (int i = 0; i < nup; ++i) {
        if (i >= inflight) (void)hipEventSynchronize(ev[(i - inflight) % 64]);
        const size_t off = (size_t)i * up_piece;
        (void)hipMemcpyAsync((char *)dx + off, (char *)hx + off, up - off < up_piece ? up - off : up_piece, hipMemcpyHostToDevice, s1);
        (void)hipEventRecord(ev[i % 64], s1);
      }
      (void)hipStreamSynchronize(s1);
      const double tu = now() - t0;
      (void)hipStreamSynchronize(s2);
      printf("  %5.1f / %5.1f", tu, now() - t0);
    }
    printf("\n");
  };
  // ... and with the library's dependencies: a kernel on a third stream waits for every second upload piece (hipStreamWaitEvent) and
  // the download pieces of its unit wait for the kernel -- on the device (hipStreamWaitEvent) or on the host (hipEventSynchronize)
  auto chained = [&](bool host_wait) {
    printf("16 MB pieces, kernels between them, downloads wait for their kernel on the %s:", host_wait ? "HOST  " : "DEVICE");
    hipStream_t s3;
    (void)hipStreamCreateWithFlags(&s3, hipStreamNonBlocking);
    const size_t piece = (size_t)16 << 20;
    const int nup = (int)((up + piece - 1) / piece), units = (nup + 1) / 2;
    hipEvent_t ue[64], ke[64];
    for (auto &e : ue) (void)hipEventCreateWithFlags(&e, hipEventDisableTiming);
    for (auto &e : ke) (void)hipEventCreateWithFlags(&e, hipEventDisableTiming);
    for (int rep = 0; rep < 4; ++rep) {
      (void)hipDeviceSynchronize();
      const double t0 = now();
      double tu = 0;
      size_t doff = 0;
      for (int u = 0; u < units; ++u) {
        for (int i = 2 * u; i < 2 * u + 2 && i < nup; ++i) {
          if (i >= 4) (void)hipEventSynchronize(ue[(i - 4) % 64]);
          const size_t off = (size_t)i * piece;
          (void)hipMemcpyAsync((char *)dx + off, (char *)hx + off, up - off < piece ? up - off : piece, hipMemcpyHostToDevice, s1);
          (void)hipEventRecord(ue[i % 64], s1);
        }
        const int last = (2 * u + 1 < nup ? 2 * u + 1 : nup - 1);
        (void)hipStreamWaitEvent(s3, ue[last % 64], 0);
        hipLaunchKernelGGL(copy_kernel, dim3(256), dim3(256), 0, s3, (float4 *)dy, (const float4 *)dx, (size_t)1 << 20);   // 16 MB on the device: a stand-in for the unit's kernels
        (void)hipEventRecord(ke[u % 64], s3);
        if (host_wait) (void)hipEventSynchronize(ke[u % 64]);
        else (void)hipStreamWaitEvent(s2, ke[u % 64], 0);
        const size_t dend = u + 1 == units ? down : (size_t)((double)down * (u + 1) / units) & ~(size_t)255;
        for (; doff < dend; doff += piece)
          (void)hipMemcpyAsync((char *)hy + doff, (char *)dy + doff, dend - doff < piece ? dend - doff : piece, hipMemcpyDeviceToHost, s2);
        doff = dend;
      }
      (void)hipStreamSynchronize(s1);
      tu = now() - t0;
      (void)hipStreamSynchronize(s2);
      printf("  %5.1f / %5.1f", tu, now() - t0);
    }
    printf("\n");
  };
  chained(false);
  chained(true);
  pieces(up, down, 64);
  pieces((size_t)16 << 20, (size_t)16 << 20, 64);
  pieces((size_t)16 << 20, (size_t)16 << 20, 4);
  pieces((size_t)16 << 20, (size_t)16 << 20, 2);
  pieces((size_t)16 << 20, (size_t)16 << 20, 1);
  pieces((size_t)16 << 20, down, 4);
  pieces((size_t)16 << 20, (size_t)64 << 20, 4);
  pieces((size_t)64 << 20, (size_t)64 << 20, 2);
  pieces((size_t)4 << 20, (size_t)16 << 20, 8);
  return 0;
}
