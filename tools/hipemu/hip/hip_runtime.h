// tools/hipemu/hip/hip_runtime.h — a tiny single-process HIP *emulator* for the CPU debug build.
//
// TEST INFRASTRUCTURE ONLY.  It lets the unmodified product sources (csrc/*.hip) be compiled with g++
// (-I tools/hipemu) so that the device code's control logic, indexing and barrier placement can be
// exercised, sanitised (ASan/UBSan) and compared with the oracle in this GPU-less container.  Every
// thread of a workgroup is a ucontext fiber; __syncthreads / wave barriers / shuffles are fiber
// rendezvous.  Workgroups run one after another.  It is never loaded by the product (vvcx.py loads the
// gfx950 library only) and is not a fallback path.
#pragma once
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <stdio.h>
#include <math.h>
#include <vector>
#include <functional>

#define __global__
#define __device__
#define __host__
#define __shared__ static
#define __launch_bounds__(...)
#define __noinline__ __attribute__((noinline))
#define __forceinline__ inline
#define VX_EMU 1          // the sources are being compiled for the CPU debug emulator
#define VX_NO_MFMA 1      // the matrix-core form of the 32- / 64-point first transform stage exists on the device only; the emulation runs the general loop
// packed 16-bit helpers of the transform stage (device: v_dot2c_i32_i16 / v_pk_sub_i16)
#define VX_DOT2_I16(a_, b_, c_) ((c_) + (int) (int16_t) (a_) * (int) (int16_t) (b_) + (int) (int16_t) ((a_) >> 16) * (int) (int16_t) ((b_) >> 16))
#define VX_PKSUB_I16(a_, b_) ((uint32_t) (uint16_t) ((a_) - (b_)) | ((uint32_t) (uint16_t) (((a_) >> 16) - ((b_) >> 16)) << 16))
#define VX_REG_BARRIER(x) ((void) 0)      // device builds: an empty asm that keeps a value in a register (see csrc/vvcx_depquant_dev.h)
// workgroup-shared storage is not cleared between workgroups on the GPU: poison it at kernel entry so that reads of never-written fields misbehave here too
// range assertions on derived addresses (path-node offsets from ancestor fields, template rows from packed state bits, decision slots): checked in the emulation only
#define VX_CHECK(c) do { if (!(c)) { fprintf(stderr, "hipemu: VX_CHECK failed: %s (%s:%d), thread %u\n", #c, __FILE__, __LINE__, hipemu::g_cur->tidx.x); abort(); } } while (0)
#define VX_POISON_LDS(obj) do { if (hipemu::g_cur->tidx.x == 0) memset((void *) &(obj), 0xA5, sizeof(obj)); hipemu::syncthreads(); } while (0)

struct uint2 { unsigned x, y; };
struct dim3 { unsigned x, y, z; dim3(unsigned a = 1, unsigned b = 1, unsigned c = 1) : x(a), y(b), z(c) {} };

namespace hipemu {
struct Fiber { void *sp; char *stack; dim3 tidx; bool done; };
extern "C" void hipemu_switch(void **save_sp, void *load_sp);
struct Block {
  std::vector<Fiber> fibers; unsigned n; unsigned arrived, gen;
  unsigned warrived[32], wgen[32]; uint64_t wslot[32][64];
};
extern Block *g_block; extern Fiber *g_cur; extern void *g_sched_sp; extern dim3 g_blockIdx, g_blockDim, g_gridDim;
inline void yield() { hipemu_switch(&g_cur->sp, g_sched_sp); }
inline void syncthreads()
{
  Block &b = *g_block; const unsigned gen = b.gen;
  if (++b.arrived == b.n) { b.arrived = 0; b.gen++; } else while (b.gen == gen) yield();
}
inline void wave_barrier()
{
  Block &b = *g_block; const unsigned w = g_cur->tidx.x >> 6; const unsigned gen = b.wgen[w];
  unsigned nw = b.n - w * 64; if (nw > 64) nw = 64;
  if (++b.warrived[w] == nw) { b.warrived[w] = 0; b.wgen[w]++; } else while (b.wgen[w] == gen) yield();
}
template <typename T> inline T shfl_idx(T v, int src)
{
  Block &b = *g_block; const unsigned w = g_cur->tidx.x >> 6, l = g_cur->tidx.x & 63;
  uint64_t bits = 0; memcpy(&bits, &v, sizeof(T));
  b.wslot[w][l] = bits; wave_barrier();
  const uint64_t r = b.wslot[w][src & 63]; wave_barrier();
  T out; memcpy(&out, &r, sizeof(T)); return out;
}
void launch(std::function<void()> body, dim3 grid, dim3 block);
}

#define threadIdx (hipemu::g_cur->tidx)
#define blockIdx (hipemu::g_blockIdx)
#define blockDim (hipemu::g_blockDim)
#define gridDim (hipemu::g_gridDim)

inline void __syncthreads() { hipemu::syncthreads(); }
inline void __threadfence_block() {}
inline void __threadfence() {}
#define __builtin_amdgcn_fence(order, scope) ((void) 0)
inline void __builtin_amdgcn_wave_barrier() { hipemu::wave_barrier(); }
template <typename T> inline T __shfl_xor(T v, int mask) { return hipemu::shfl_idx(v, (int) ((hipemu::g_cur->tidx.x & 63) ^ (unsigned) mask)); }
template <typename T> inline T __shfl(T v, int src) { return hipemu::shfl_idx(v, src); }
inline long long clock64() { return 0; }
inline long long wall_clock64() { return 0; }
inline int __builtin_amdgcn_readfirstlane(int v) { return v; }
inline void __builtin_amdgcn_s_setprio(int) {}
inline unsigned long long __ballot(int pred)
{
  hipemu::Block &b = *hipemu::g_block; const unsigned w = hipemu::g_cur->tidx.x >> 6, l = hipemu::g_cur->tidx.x & 63;
  unsigned nw = b.n - w * 64; if (nw > 64) nw = 64;
  b.wslot[w][l] = pred ? 1 : 0; hipemu::wave_barrier();
  unsigned long long m = 0; for (unsigned i = 0; i < nw; i++) m |= (unsigned long long) (b.wslot[w][i] & 1) << i;
  hipemu::wave_barrier();
  return m;
}
inline unsigned __builtin_amdgcn_mbcnt_lo(unsigned m, unsigned base) { const unsigned l = hipemu::g_cur->tidx.x & 63; return base + (unsigned) __builtin_popcount(m & (l >= 32 ? 0xffffffffu : ((1u << l) - 1))); }
inline unsigned __builtin_amdgcn_mbcnt_hi(unsigned m, unsigned base) { const unsigned l = hipemu::g_cur->tidx.x & 63; return base + (unsigned) __builtin_popcount(m & (l <= 32 ? 0u : ((1u << (l - 32)) - 1))); }
inline int __builtin_amdgcn_readlane(int v, int lane) { return hipemu::shfl_idx(v, lane); }
// DPP controls used by the kernel's wave reductions: quad_perm (0x00-0xFF), row_ror (0x121-0x12F), row_bcast15/31 (0x142/0x143)
inline int __builtin_amdgcn_update_dpp(int old, int src, int ctrl, int row_mask, int bank_mask, bool bound_ctrl)
{
  hipemu::Block &b = *hipemu::g_block; const unsigned w = hipemu::g_cur->tidx.x >> 6, l = hipemu::g_cur->tidx.x & 63;
  b.wslot[w][l] = (uint64_t) (uint32_t) src; hipemu::wave_barrier();
  int sl = -1;
  if (ctrl <= 0xFF) sl = (int) ((l & ~3u) | ((unsigned) (ctrl >> (2 * (l & 3))) & 3));
  else if (ctrl >= 0x121 && ctrl <= 0x12F) sl = (int) ((l & ~15u) | ((l - (unsigned) (ctrl & 15)) & 15));
  else if (ctrl == 0x140) sl = (int) ((l & ~15u) | (15 - (l & 15)));          // row_mirror
  else if (ctrl == 0x141) sl = (int) ((l & ~7u) | (7 - (l & 7)));             // row_half_mirror
  else if (ctrl == 0x142) sl = (l >> 4) >= 1 ? (int) (((l >> 4) - 1) * 16 + 15) : -1;
  else if (ctrl == 0x143) sl = (l >> 4) >= 2 ? 31 : -1;
  const bool en = ((row_mask >> (l >> 4)) & 1) && ((bank_mask >> ((l >> 2) & 3)) & 1);
  const int r = !en ? old : (sl >= 0 ? (int) (uint32_t) b.wslot[w][sl] : (bound_ctrl ? 0 : old));
  hipemu::wave_barrier();
  return r;
}
inline int __clzll(long long v) { return v == 0 ? 64 : __builtin_clzll((unsigned long long) v); }
inline int __popcll(unsigned long long v) { return __builtin_popcountll(v); }
inline int __ffsll(unsigned long long v) { return __builtin_ffsll((long long) v); }
inline int __clz(int v) { return v == 0 ? 32 : __builtin_clz((unsigned) v); }
inline unsigned long long atomicAdd(unsigned long long *p, unsigned long long v) { const unsigned long long o = *p; *p += v; return o; }
inline unsigned atomicAdd(unsigned *p, unsigned v) { const unsigned o = *p; *p += v; return o; }
inline int atomicAdd(int *p, int v) { const int o = *p; *p += v; return o; }

// ---- the few host runtime calls the C-ABI layer uses
typedef int hipError_t; typedef void *hipStream_t; typedef void *hipEvent_t;
enum { hipSuccess = 0 };
enum hipMemcpyKind { hipMemcpyHostToDevice, hipMemcpyDeviceToHost, hipMemcpyDeviceToDevice };
inline hipError_t hipSetDevice(int) { return 0; }
inline hipError_t hipGetDevice(int *d) { *d = 0; return 0; }
enum { hipDeviceAttributeMultiprocessorCount = 0 };
inline hipError_t hipDeviceGetAttribute(int *v, int, int) { *v = 1; return 0; }
inline hipError_t hipOccupancyMaxActiveBlocksPerMultiprocessor(int *v, const void *, int, size_t) { *v = 1; return 0; }
// Device allocations are NOT zeroed on the GPU and a write beyond one is a fault that only sometimes shows (whether pages happen to be mapped behind it - round 2's
// "layout-dependent" abort was such a write, 3 MB behind a per-stream scratch slot).  The emulation therefore fills every allocation with a poison pattern (a read
// before the first write differs from the oracle instead of reading zeros) and puts red zones of HIPEMU_REDZONE bytes on both sides that hipFree checks.
#ifndef HIPEMU_REDZONE
#define HIPEMU_REDZONE (4u << 20)
#endif
namespace hipemu {
struct AllocHdr { size_t n; uint64_t magic; };
inline void check_zone(const unsigned char *z, size_t n, const char *what, const void *user)
{
  for (size_t i = 0; i < n; i++) if (z[i] != 0xCB) { fprintf(stderr, "hipemu: write %s allocation %p (offset %zu of the red zone)\n", what, user, i); abort(); }
}
}
inline hipError_t hipMalloc(void **p, size_t n)
{
  if (!n) n = 1;
  unsigned char *raw = (unsigned char *) malloc(sizeof(hipemu::AllocHdr) + 2 * (size_t) HIPEMU_REDZONE + n);
  if (!raw) return 2;
  hipemu::AllocHdr *h = (hipemu::AllocHdr *) raw; h->n = n; h->magic = 0x48495045ull;
  memset(raw + sizeof(hipemu::AllocHdr), 0xCB, HIPEMU_REDZONE);
  memset(raw + sizeof(hipemu::AllocHdr) + HIPEMU_REDZONE, 0xA5, n);
  memset(raw + sizeof(hipemu::AllocHdr) + HIPEMU_REDZONE + n, 0xCB, HIPEMU_REDZONE);
  *p = raw + sizeof(hipemu::AllocHdr) + HIPEMU_REDZONE;
  return 0;
}
inline hipError_t hipFree(void *p)
{
  if (!p) return 0;
  unsigned char *raw = (unsigned char *) p - HIPEMU_REDZONE - sizeof(hipemu::AllocHdr);
  hipemu::AllocHdr *h = (hipemu::AllocHdr *) raw;
  if (h->magic != 0x48495045ull) { fprintf(stderr, "hipemu: hipFree of a pointer hipMalloc did not return (%p)\n", p); abort(); }
  hipemu::check_zone(raw + sizeof(hipemu::AllocHdr), HIPEMU_REDZONE, "in front of", p);
  hipemu::check_zone((unsigned char *) p + h->n, HIPEMU_REDZONE, "behind", p);
  h->magic = 0;
  free(raw);
  return 0;
}
inline hipError_t hipMemset(void *p, int v, size_t n) { memset(p, v, n); return 0; }
inline hipError_t hipMemsetAsync(void *p, int v, size_t n, hipStream_t) { memset(p, v, n); return 0; }
inline hipError_t hipMemcpy(void *d, const void *s, size_t n, hipMemcpyKind) { memcpy(d, s, n); return 0; }
inline hipError_t hipMemcpyAsync(void *d, const void *s, size_t n, hipMemcpyKind, hipStream_t) { memcpy(d, s, n); return 0; }
inline hipError_t hipStreamSynchronize(hipStream_t) { return 0; }
enum { hipErrorNotReady = 600 };
inline hipError_t hipStreamQuery(hipStream_t) { return 0; }
inline hipError_t hipHostMalloc(void **p, size_t n, unsigned) { *p = calloc(1, n ? n : 1); return *p ? 0 : 2; }
inline hipError_t hipHostFree(void *p) { free(p); return 0; }
inline hipError_t hipDeviceSynchronize() { return 0; }
inline hipError_t hipEventCreate(hipEvent_t *e) { *e = nullptr; return 0; }
inline hipError_t hipEventDestroy(hipEvent_t) { return 0; }
inline hipError_t hipEventRecord(hipEvent_t, hipStream_t) { return 0; }
inline hipError_t hipEventSynchronize(hipEvent_t) { return 0; }
inline hipError_t hipEventElapsedTime(float *ms, hipEvent_t, hipEvent_t) { *ms = 0.f; return 0; }
inline hipError_t hipGetLastError() { return 0; }
inline const char *hipGetErrorString(hipError_t) { return "hipemu"; }
#define hipLaunchKernelGGL(kernel, grid, block, shmem, stream, ...) hipemu::launch([&]() { kernel(__VA_ARGS__); }, (grid), (block))
