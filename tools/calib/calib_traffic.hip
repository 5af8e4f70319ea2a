// calib_traffic.hip - known-byte-count access patterns in the shapes vvcx_compress_kernel uses, to calibrate rocprofv3's FETCH_SIZE / WRITE_SIZE
// (MI355X_MICROARCH.md: "calibrate on a known byte count in your own access pattern before trusting an absolute").  Diagnostic only; not part of libvvcx.so.
//   cal_w16   16 B per lane, coalesced stream                      (the guide's calibrated case)
//   cal_w2     2 B per lane, coalesced (one int16 sample per lane: 128 B per wave store, how block rows / pools are written)
//   cal_w2_blk 2 B per lane, 16 lanes of a wave active: one 4x4 int16 block (32 B) per store, blocks 256 B apart
//   cal_w2_hot 2 B per lane into a 1 KB region per wave, rewritten `rep` times (a stream's hot scratch lines: does a rewrite count again?)
//   cal_r16 / cal_r2 / cal_r2_hot  the same for reads
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>
#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

extern "C" __global__ void cal_w16(uint4 *p, size_t n) { for (size_t i = blockIdx.x * (size_t) blockDim.x + threadIdx.x; i < n; i += (size_t) gridDim.x * blockDim.x) p[i] = make_uint4((unsigned) i, 1, 2, 3); }
extern "C" __global__ void cal_w2(int16_t *p, size_t n) { for (size_t i = blockIdx.x * (size_t) blockDim.x + threadIdx.x; i < n; i += (size_t) gridDim.x * blockDim.x) p[i] = (int16_t) i; }
extern "C" __global__ void cal_w2_blk(int16_t *p, size_t nblk)
{
  const int lane = threadIdx.x & 63; const size_t wave = (blockIdx.x * (size_t) blockDim.x + threadIdx.x) >> 6, nw = ((size_t) gridDim.x * blockDim.x) >> 6;
  for (size_t b = wave; b < nblk; b += nw) if (lane < 16) p[b * 128 + lane] = (int16_t) b;
}
extern "C" __global__ void cal_w2_hot(int16_t *p, int rep)
{
  const int lane = threadIdx.x & 63; const size_t wave = (blockIdx.x * (size_t) blockDim.x + threadIdx.x) >> 6;
  volatile int16_t *q = p + wave * 512;
  for (int r = 0; r < rep; r++) for (int e = lane; e < 512; e += 64) q[e] = (int16_t) (r + e);
}
extern "C" __global__ void cal_r16(const uint4 *p, size_t n, unsigned *out) { unsigned s = 0; for (size_t i = blockIdx.x * (size_t) blockDim.x + threadIdx.x; i < n; i += (size_t) gridDim.x * blockDim.x) { const uint4 v = p[i]; s += v.x ^ v.y ^ v.z ^ v.w; } if (s == 0x12345678u) *out = s; }
extern "C" __global__ void cal_r2(const int16_t *p, size_t n, unsigned *out) { unsigned s = 0; for (size_t i = blockIdx.x * (size_t) blockDim.x + threadIdx.x; i < n; i += (size_t) gridDim.x * blockDim.x) s += (unsigned) p[i]; if (s == 0x12345678u) *out = s; }
extern "C" __global__ void cal_r2_hot(const int16_t *p, int rep, unsigned *out)
{
  const int lane = threadIdx.x & 63; const size_t wave = (blockIdx.x * (size_t) blockDim.x + threadIdx.x) >> 6;
  const volatile int16_t *q = p + wave * 512; unsigned s = 0;
  for (int r = 0; r < rep; r++) for (int e = lane; e < 512; e += 64) s += (unsigned) q[e];
  if (s == 0x12345678u) *out = s;
}

int main()
{
  const size_t BYTES = (size_t) 2 << 30;                  // 2 GiB per streaming pattern: far past L2 (4 MB / XCD) and the 256 MB Infinity Cache
  void *buf; unsigned *out;
  CHK(hipMalloc(&buf, BYTES)); CHK(hipMalloc((void **) &out, 4)); CHK(hipMemset(buf, 1, BYTES));
  const int grid = 256 * 8, block = 256, waves = grid * block / 64, rep = 2000;
  const size_t nblk = BYTES / 256;
  hipLaunchKernelGGL(cal_w16, dim3(grid), dim3(block), 0, 0, (uint4 *) buf, BYTES / 16);
  hipLaunchKernelGGL(cal_w2, dim3(grid), dim3(block), 0, 0, (int16_t *) buf, BYTES / 2);
  hipLaunchKernelGGL(cal_w2_blk, dim3(grid), dim3(block), 0, 0, (int16_t *) buf, nblk);
  hipLaunchKernelGGL(cal_w2_hot, dim3(grid), dim3(block), 0, 0, (int16_t *) buf, rep);
  hipLaunchKernelGGL(cal_r16, dim3(grid), dim3(block), 0, 0, (const uint4 *) buf, BYTES / 16, out);
  hipLaunchKernelGGL(cal_r2, dim3(grid), dim3(block), 0, 0, (const int16_t *) buf, BYTES / 2, out);
  hipLaunchKernelGGL(cal_r2_hot, dim3(grid), dim3(block), 0, 0, (const int16_t *) buf, rep, out);
  CHK(hipDeviceSynchronize());
  printf("{\"cal_w16\": %zu, \"cal_w2\": %zu, \"cal_w2_blk\": %zu, \"cal_w2_hot\": %zu, \"cal_r16\": %zu, \"cal_r2\": %zu, \"cal_r2_hot\": %zu, \"hot_region_bytes\": %zu}\n",
         BYTES, BYTES, nblk * 32, (size_t) waves * 1024 * rep, BYTES, BYTES, (size_t) waves * 1024 * rep, (size_t) waves * 1024);
  return 0;
}
