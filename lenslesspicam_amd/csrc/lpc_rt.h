// lpc_rt.h -- thin runtime layer for the deconvolution engine.
//
// Product build (hipcc, gfx950): everything maps 1:1 onto the HIP runtime.
//
// LPC_SIMT_EMU build (g++, tests only): the SAME kernel sources are compiled for
// the host and executed by a cooperative-fibre SIMT emulator that lives under
// tests/simt_emu/.  It exists because the build container has no GPU: it lets the
// CPU test-suite execute the real kernel bodies (index maths, LDS staging, barrier
// structure) against the oracle before a GPU-minute is spent.  It is NOT a product
// path: lenslesspicam_amd never builds, ships or loads it, and the product library
// refuses to run without a HIP device.
#pragma once

#include <cstddef>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <cmath>

#if defined(LPC_SIMT_EMU)

// ----------------------------------------------------------------- emulator --
#include <functional>

struct dim3 {
  unsigned x, y, z;
  dim3(unsigned x_ = 1, unsigned y_ = 1, unsigned z_ = 1) : x(x_), y(y_), z(z_) {}
};
struct float2 { float x, y; };
struct float4 { float x, y, z, w; };
struct double2 { double x, y; };
static inline double2 make_double2(double x, double y) { double2 r; r.x = x; r.y = y; return r; }
static inline float2 make_float2(float x, float y) { float2 r; r.x = x; r.y = y; return r; }
static inline float4 make_float4(float x, float y, float z, float w) { float4 r; r.x = x; r.y = y; r.z = z; r.w = w; return r; }

namespace lpc_emu {
struct ThreadCtx { dim3 tid, bid, bdim, gdim; char* smem; };
ThreadCtx& ctx();                 // current fibre's coordinates
void barrier();                   // __syncthreads()
void launch(dim3 grid, dim3 block, size_t smem_bytes, const std::function<void()>& body);
}  // namespace lpc_emu

#define __global__
#define __device__
#define __host__
#define __forceinline__ inline
#define __launch_bounds__(...)
#define LPC_RESTRICT
#define threadIdx (lpc_emu::ctx().tid)
#define blockIdx (lpc_emu::ctx().bid)
#define blockDim (lpc_emu::ctx().bdim)
#define gridDim (lpc_emu::ctx().gdim)
#define __syncthreads() lpc_emu::barrier()
#define LPC_DYN_SMEM(name) char* name = lpc_emu::ctx().smem
#define LPC_TID(nt) ((int)threadIdx.x)
static inline unsigned __umulhi(unsigned a, unsigned b) { return (unsigned)(((uint64_t)a * (uint64_t)b) >> 32); }
// range-checked row accesses (see the HIP branch): the emulator checks the range itself
struct lpc_rsrc { char* base; unsigned bytes; };
static inline lpc_rsrc lpc_make_rsrc(const void* base, unsigned bytes) { lpc_rsrc r; r.base = (char*)base; r.bytes = bytes; return r; }
static inline int lpc_opaque(int x) { return x; }
#define LPC_SCHED_FENCE() ((void)0)

typedef void* lpcStream_t;
typedef int lpcError_t;
#define lpcSuccess 0

namespace rt {
static inline const char* err_string(lpcError_t) { return "emu"; }
static inline lpcError_t dev_malloc(void** p, size_t n) { *p = std::malloc(n ? n : 1); return *p ? 0 : 1; }
static inline lpcError_t dev_free(void* p) { std::free(p); return 0; }
static inline lpcError_t dev_malloc_async(void** p, size_t n, lpcStream_t) { return dev_malloc(p, n); }
static inline lpcError_t dev_free_async(void* p, lpcStream_t) { std::free(p); return 0; }
static inline lpcError_t memset_async(void* p, int v, size_t n, lpcStream_t) { std::memset(p, v, n); return 0; }
static inline lpcError_t copy_d2d_async(void* d, const void* s, size_t n, lpcStream_t) { std::memmove(d, s, n); return 0; }
static inline lpcError_t copy_h2d_async(void* d, const void* s, size_t n, lpcStream_t) { std::memcpy(d, s, n); return 0; }
static inline lpcError_t copy_d2h_async(void* d, const void* s, size_t n, lpcStream_t) { std::memcpy(d, s, n); return 0; }
static inline lpcError_t stream_sync(lpcStream_t) { return 0; }
static inline lpcError_t last_error() { return 0; }
static inline lpcError_t device_count(int* n) { *n = 1; return 0; }
static inline lpcError_t current_device(int* d) { *d = 0; return 0; }
static inline lpcError_t set_max_dyn_smem(const void*, size_t) { return 0; }
static inline const char* backend_name() { return "simt-emu(test-only)"; }
static inline int cu_count() { return 2; }
}  // namespace rt

#define LPC_LAUNCH(kernel, grid, block, smem, stream, ...) \
  lpc_emu::launch((grid), (block), (smem), [=]() { kernel(__VA_ARGS__); })

#else

// ---------------------------------------------------------------------- HIP --
#include <hip/hip_runtime.h>

#define LPC_RESTRICT __restrict__
// all LDS is dynamic and 16-byte aligned (guide: Guideline 17)
#define LPC_DYN_SMEM(name)                                              \
  extern __shared__ __attribute__((aligned(16))) char lpc_dyn_smem_[];  \
  char* name = lpc_dyn_smem_
// threadIdx.x with its range: __launch_bounds__ alone does not tell the optimiser that tid < NT, and without it every
// `tid + k * NT < NELEM` guard that is always true stays a branch -- with the load inside it and a `s_waitcnt vmcnt(0)`
// behind the load (k_cols_mid_admm_seq: 4 of a lane's 17 tile loads went out one HBM latency after the other)
static __device__ __forceinline__ int lpc_tid_below(unsigned nt) {
  const unsigned t = threadIdx.x;
  __builtin_assume(t < nt);
  return (int)t;
}
#define LPC_TID(nt) lpc_tid_below((unsigned)(nt))

// LPC_STAMP (timing builds of a plan module only, tools/stamp_timeline.py): lane 0 of every workgroup writes the 100-MHz
// real-time counter at kernel entry, behind every barrier and -- after its stores have been acknowledged -- at exit into
// lpc_stamp_buf[kernel][workgroup][slot]; the last launch of each kernel stays readable through lpc_module_stamps().
#if defined(LPC_STAMP)
#define LPC_STAMP_KERNELS 4
#define LPC_STAMP_WGS 4096
#define LPC_STAMP_SLOTS 32
static __device__ unsigned long long lpc_stamp_buf[LPC_STAMP_KERNELS * LPC_STAMP_WGS * LPC_STAMP_SLOTS];
__shared__ unsigned lpc_stamp_lds[2];
static __device__ __forceinline__ void lpc_stamp_put() {
  if (threadIdx.x == 0) {
    const unsigned wg = blockIdx.x + gridDim.x * blockIdx.y, n = lpc_stamp_lds[1];
    if (wg < LPC_STAMP_WGS && n < LPC_STAMP_SLOTS - 1)
      lpc_stamp_buf[((size_t)lpc_stamp_lds[0] * LPC_STAMP_WGS + wg) * LPC_STAMP_SLOTS + 1 + n] = __builtin_amdgcn_s_memrealtime();
    lpc_stamp_lds[1] = n + 1;
  }
}
static __device__ __forceinline__ void lpc_stamp_begin(unsigned kid) {
  if (threadIdx.x == 0) { lpc_stamp_lds[0] = kid; lpc_stamp_lds[1] = 0; }
  lpc_stamp_put();
}
static __device__ __forceinline__ void lpc_stamp_end() {
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  lpc_stamp_put();
  if (threadIdx.x == 0) {
    const unsigned wg = blockIdx.x + gridDim.x * blockIdx.y;
    if (wg < LPC_STAMP_WGS) lpc_stamp_buf[((size_t)lpc_stamp_lds[0] * LPC_STAMP_WGS + wg) * LPC_STAMP_SLOTS] = lpc_stamp_lds[1];
  }
}
static __device__ __forceinline__ void lpc_sync_stamped() { __syncthreads(); lpc_stamp_put(); }
#define __syncthreads() lpc_sync_stamped()
#define LPC_STAMP_BEGIN(kid) lpc_stamp_begin(kid)
#define LPC_STAMP_END() lpc_stamp_end()
#endif

// Range-checked accesses to one row of an un-padded plane ("pad on load, crop on store" done by the address unit): a raw
// buffer resource over the row's bytes; a load whose byte offset (unsigned: a negative column is a huge offset) lies
// outside it returns zero, a store outside it is dropped -- no clamped addresses, no selects, no branches, and one 32-bit
// offset register per access instead of a 64-bit address.  The offset must arrive COMPLETE in the VGPR: the range check
// does not wrap, so "negative offset + positive immediate" would be out of range even when the sum is not -- lpc_opaque()
// keeps the compiler from splitting a constant off into the instruction's immediate field.
typedef __amdgpu_buffer_rsrc_t lpc_rsrc;
static __device__ __forceinline__ lpc_rsrc lpc_make_rsrc(const void* base, unsigned bytes) {
  return __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, (int)bytes, 0x00020000);   // gfx9: DATA_FORMAT = 32 bit
}
static __device__ __forceinline__ int lpc_opaque(int x) { asm volatile("" : "+v"(x)); return x; }
// nothing is scheduled across this point (keeps a batch of loads behind the arithmetic whose registers it needs)
#define LPC_SCHED_FENCE() __builtin_amdgcn_sched_barrier(0)
typedef hipStream_t lpcStream_t;
typedef hipError_t lpcError_t;
#define lpcSuccess hipSuccess

namespace rt {
static inline const char* err_string(lpcError_t e) { return hipGetErrorString(e); }
static inline lpcError_t dev_malloc(void** p, size_t n) { return hipMalloc(p, n ? n : 1); }
static inline lpcError_t dev_free(void* p) { return hipFree(p); }
// stream-ordered scratch (the device's default memory pool): no host synchronisation on either side
static inline lpcError_t dev_malloc_async(void** p, size_t n, lpcStream_t s) { return hipMallocAsync(p, n ? n : 1, s); }
static inline lpcError_t dev_free_async(void* p, lpcStream_t s) { return hipFreeAsync(p, s); }
static inline lpcError_t memset_async(void* p, int v, size_t n, lpcStream_t s) { return hipMemsetAsync(p, v, n, s); }
static inline lpcError_t copy_d2d_async(void* d, const void* s, size_t n, lpcStream_t st) { return hipMemcpyAsync(d, s, n, hipMemcpyDeviceToDevice, st); }
static inline lpcError_t copy_h2d_async(void* d, const void* s, size_t n, lpcStream_t st) { return hipMemcpyAsync(d, s, n, hipMemcpyHostToDevice, st); }
static inline lpcError_t copy_d2h_async(void* d, const void* s, size_t n, lpcStream_t st) { return hipMemcpyAsync(d, s, n, hipMemcpyDeviceToHost, st); }
static inline lpcError_t stream_sync(lpcStream_t s) { return hipStreamSynchronize(s); }
static inline lpcError_t last_error() { return hipGetLastError(); }
static inline lpcError_t device_count(int* n) { return hipGetDeviceCount(n); }
static inline lpcError_t current_device(int* d) { return hipGetDevice(d); }
static inline lpcError_t set_max_dyn_smem(const void* fn, size_t bytes) {
  return hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
}
static inline const char* backend_name() { return "hip-gfx950"; }
static inline int cu_count() {     // compute units of the current device (persistent kernels size their grids by it)
  int dev = 0, n = 0;
  if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) return 256;
  return n;
}
}  // namespace rt

#define LPC_LAUNCH(kernel, grid, block, smem, stream, ...) \
  hipLaunchKernelGGL(kernel, (grid), (block), (smem), (stream), __VA_ARGS__)

#endif

// ------------------------------------------------------------ arithmetic type --
// The engine is written once over `real`; the SAME translation unit is compiled twice:
// liblpc.so (float, the reference's default dtype) and liblpc_f64.so (-DLPC_DOUBLE, dtype="float64").
#ifdef LPC_DOUBLE
typedef double real;
typedef double2 real2;
struct real4_t { double x, y, z, w; };
#define make_real2 make_double2
#define LPC_REAL_NAME "float64"
static __host__ __device__ __forceinline__ real rmax(real a, real b) { return fmax(a, b); }
static __host__ __device__ __forceinline__ real rmin(real a, real b) { return fmin(a, b); }
static __host__ __device__ __forceinline__ real rabs(real a) { return fabs(a); }
static __host__ __device__ __forceinline__ real rsqrt_of(real a) { return sqrt(a); }
#else
typedef float real;
typedef float2 real2;
typedef float4 real4_t;     // four reals in one 16-byte access
#define make_real2 make_float2
#define LPC_REAL_NAME "float32"
static __host__ __device__ __forceinline__ real rmax(real a, real b) { return fmaxf(a, b); }
static __host__ __device__ __forceinline__ real rmin(real a, real b) { return fminf(a, b); }
static __host__ __device__ __forceinline__ real rabs(real a) { return fabsf(a); }
static __host__ __device__ __forceinline__ real rsqrt_of(real a) { return sqrtf(a); }
#endif

// two adjacent floats of a row through its buffer resource (byte offset of the first; both in range or both out)
#if defined(LPC_SIMT_EMU)
static inline real2 lpc_buf_load2(lpc_rsrc r, int off) {
  return (unsigned)off + 2 * sizeof(real) <= r.bytes && (unsigned)off < r.bytes ? *(const real2*)(r.base + off) : make_real2((real)0., (real)0.);
}
static inline void lpc_buf_store2(lpc_rsrc r, int off, real2 v) {
  if ((unsigned)off + 2 * sizeof(real) <= r.bytes && (unsigned)off < r.bytes) *(real2*)(r.base + off) = v;
}
#elif !defined(LPC_DOUBLE)
// (the loaded / stored pair is converted as a WHOLE vector: hipcc 7.2 folds `bit_cast<float>(v.x), bit_cast<float>(v.y)` on
// the builtin's result into two copies of a one-dword load)
static __device__ __forceinline__ real2 lpc_buf_load2(lpc_rsrc r, int off) {
  typedef unsigned lpc_u2 __attribute__((ext_vector_type(2)));
  typedef float lpc_f2 __attribute__((ext_vector_type(2)));
  const lpc_f2 f = __builtin_bit_cast(lpc_f2, (lpc_u2)__builtin_amdgcn_raw_buffer_load_b64(r, off, 0, 0));
  return make_real2(f.x, f.y);
}
static __device__ __forceinline__ void lpc_buf_store2(lpc_rsrc r, int off, real2 v) {
  typedef unsigned lpc_u2 __attribute__((ext_vector_type(2)));
  typedef float lpc_f2 __attribute__((ext_vector_type(2)));
  lpc_f2 f;
  f.x = v.x;
  f.y = v.y;
  __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(lpc_u2, f), r, off, 0, 0);
}
#endif

// ------------------------------------------------------------ two-wide float math --
// v2f: two independent float values per operand.  On gfx950 `+ - *` and fma2() on this type become v_pk_add_f32 /
// v_pk_mul_f32 / v_pk_fma_f32 -- two results per VALU issue slot, which matters where a kernel is bound by instruction
// issue (the image-domain half of the fused ADMM row kernel: 4 independent pixels per lane).  The emulator build
// uses a plain struct.
#if defined(LPC_SIMT_EMU)
struct v2f { float x, y; };
static inline v2f mk2(float a, float b) { v2f r; r.x = a; r.y = b; return r; }
static inline v2f operator+(v2f a, v2f b) { return mk2(a.x + b.x, a.y + b.y); }
static inline v2f operator-(v2f a, v2f b) { return mk2(a.x - b.x, a.y - b.y); }
static inline v2f operator*(v2f a, v2f b) { return mk2(a.x * b.x, a.y * b.y); }
static inline v2f operator*(float a, v2f b) { return mk2(a * b.x, a * b.y); }
static inline v2f fma2(v2f a, v2f b, v2f c) { return mk2(std::fma(a.x, b.x, c.x), std::fma(a.y, b.y, c.y)); }
#else
typedef float v2f __attribute__((ext_vector_type(2)));
static __device__ __forceinline__ v2f mk2(float a, float b) { v2f r; r.x = a; r.y = b; return r; }
static __device__ __forceinline__ v2f fma2(v2f a, v2f b, v2f c) { return __builtin_elementwise_fma(a, b, c); }
#endif

// 24-bit multiply (full-rate v_mul_u32_u24 / v_mad_u32_u24; a 32-bit v_mul_lo_u32 issues at quarter rate): tile-local
// row indices times a row pitch in bytes, both far below 2^24
#if defined(LPC_SIMT_EMU)
static inline unsigned mul24(unsigned a, unsigned b) { return a * b; }
#else
static __device__ __forceinline__ unsigned mul24(unsigned a, unsigned b) { return __umul24(a, b); }
#endif

// ------------------------------------------------------------ small helpers --
// Complex helpers.  In the float32 device build a complex number IS a two-wide operand (LPC_PK_COMPLEX): add / subtract
// are one v_pk_add_f32, a product is v_pk_mul_f32 + v_pk_fma_f32 on (re, im) and the swapped pair -- written on v2f
// so that the compiler keeps every value in its own aligned register pair instead of pairing unrelated scalars
// (which cost the 540-point middle 815 register moves out of 3979 VALU instructions, profiles/r02_notes.md).
#if defined(__HIP_DEVICE_COMPILE__) && !defined(LPC_DOUBLE) && !defined(LPC_SIMT_EMU) && !defined(LPC_NO_PK_COMPLEX)
#define LPC_PK_COMPLEX 1
static __device__ __forceinline__ v2f c2v(real2 a) { return mk2(a.x, a.y); }
static __device__ __forceinline__ real2 v2c(v2f a) { return make_real2(a.x, a.y); }
static __device__ __forceinline__ v2f swap2(v2f a) { return __builtin_shufflevector(a, a, 1, 0); }
#endif
static __host__ __device__ __forceinline__ real2 cmul(real2 a, real2 b) {
#ifdef LPC_PK_COMPLEX
  const v2f A = c2v(a);
  return v2c(fma2(swap2(A), mk2(-b.y, b.y), A * mk2(b.x, b.x)));
#else
  return make_real2(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x);
#endif
}
static __host__ __device__ __forceinline__ real2 cmul_conj(real2 a, real2 b) {  // a * conj(b)
#ifdef LPC_PK_COMPLEX
  const v2f A = c2v(a);
  return v2c(fma2(swap2(A), mk2(b.y, -b.y), A * mk2(b.x, b.x)));
#else
  return make_real2(a.x * b.x + a.y * b.y, a.y * b.x - a.x * b.y);
#endif
}
static __host__ __device__ __forceinline__ real2 cadd(real2 a, real2 b) {
#ifdef LPC_PK_COMPLEX
  return v2c(c2v(a) + c2v(b));
#else
  return make_real2(a.x + b.x, a.y + b.y);
#endif
}
static __host__ __device__ __forceinline__ real2 csub(real2 a, real2 b) {
#ifdef LPC_PK_COMPLEX
  return v2c(c2v(a) - c2v(b));
#else
  return make_real2(a.x - b.x, a.y - b.y);
#endif
}
static __host__ __device__ __forceinline__ real2 cconj(real2 a) { return make_real2(a.x, -a.y); }
static __host__ __device__ __forceinline__ real2 cscale(real2 a, real s) {
#ifdef LPC_PK_COMPLEX
  return v2c(c2v(a) * mk2(s, s));
#else
  return make_real2(a.x * s, a.y * s);
#endif
}
// a * s + b (s real)
static __host__ __device__ __forceinline__ real2 caxpy(real2 a, real s, real2 b) {
#ifdef LPC_PK_COMPLEX
  return v2c(fma2(c2v(a), mk2(s, s), c2v(b)));
#else
  return make_real2(a.x * s + b.x, a.y * s + b.y);
#endif
}
// -i a = (a.y, -a.x)   /   +i a = (-a.y, a.x)
static __host__ __device__ __forceinline__ real2 cmul_mi(real2 a) {
#ifdef LPC_PK_COMPLEX
  const v2f A = c2v(a);
  return v2c(__builtin_shufflevector(A, -A, 1, 2));
#else
  return make_real2(a.y, -a.x);
#endif
}
static __host__ __device__ __forceinline__ real2 cmul_pi(real2 a) {
#ifdef LPC_PK_COMPLEX
  const v2f A = c2v(a);
  return v2c(__builtin_shufflevector(-A, A, 1, 2));
#else
  return make_real2(-a.y, a.x);
#endif
}

// division of small non-negative ints by a plan-time constant: q = floor(n/d) for
// n*d < 2^32 (all tile-local indices here are < 2^16).
struct FastDiv {
  unsigned d, m;  // m = floor(2^32/d) + 1  (d >= 2);  d == 1 handled separately
};
static inline FastDiv make_fastdiv(unsigned d) {
  FastDiv f; f.d = d; f.m = d > 1 ? (unsigned)((0x100000000ull / d) + 1ull) : 0u; return f;
}
static __host__ __device__ __forceinline__ FastDiv make_fastdiv_dev1() {
  FastDiv f; f.d = 1; f.m = 0; return f;
}
static __device__ __forceinline__ unsigned fd_div(unsigned n, FastDiv f) {
  return f.d == 1 ? n : __umulhi(n, f.m);
}

#ifndef LPC_STAMP_BEGIN
#define LPC_STAMP_BEGIN(kid) ((void)0)
#define LPC_STAMP_END() ((void)0)
#endif
