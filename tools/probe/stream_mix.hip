// Probe: achievable HBM bandwidth of an 8-read / 7-write streaming mix (the traffic shape of the fused
// ADMM prox/update kernel) for three access shapes.  Build: hipcc --offload-arch=gfx950 -O3 stream_mix.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
struct Ptrs { const float* in[8]; float* out[7]; };

// A: scalar lanes, 16 x 64 tiles (what k_admm_spatial does today)
__global__ __launch_bounds__(256) void k_tile_scalar(Ptrs p, int H, int W, int tiles_x) {
  const int ty = blockIdx.x / tiles_x, tx = blockIdx.x - ty * tiles_x;
  for (int e = threadIdx.x; e < 16 * 64; e += 256) {
    const int r = ty * 16 + e / 64, c = tx * 64 + (e & 63);
    const long o = (long)r * W + c;
    float s = 0.f;
#pragma unroll
    for (int k = 0; k < 8; ++k) s += p.in[k][o];
#pragma unroll
    for (int k = 0; k < 7; ++k) p.out[k][o] = s + k;
  }
}
// B: float4 lanes, TH x 256 tiles
template <int TH>
__global__ __launch_bounds__(256) void k_tile_vec4(Ptrs p, int H, int W, int tiles_x) {
  const int ty = blockIdx.x / tiles_x, tx = blockIdx.x - ty * tiles_x;
  for (int e = threadIdx.x; e < TH * 64; e += 256) {
    const int r = ty * TH + e / 64, c = tx * 256 + (e & 63) * 4;
    const long o = (long)r * W + c;
    float4 s = make_float4(0, 0, 0, 0);
#pragma unroll
    for (int k = 0; k < 8; ++k) { float4 v = *(const float4*)(p.in[k] + o); s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w; }
#pragma unroll
    for (int k = 0; k < 7; ++k) *(float4*)(p.out[k] + o) = make_float4(s.x + k, s.y, s.z, s.w);
  }
}
// C: plain linear float4 grid-stride (ceiling)
__global__ __launch_bounds__(256) void k_linear_vec4(Ptrs p, long n4) {
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long)gridDim.x * 256) {
    float4 s = make_float4(0, 0, 0, 0);
#pragma unroll
    for (int k = 0; k < 8; ++k) { float4 v = ((const float4*)p.in[k])[i]; s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w; }
#pragma unroll
    for (int k = 0; k < 7; ++k) ((float4*)p.out[k])[i] = make_float4(s.x + k, s.y, s.z, s.w);
  }
}
int main() {
  const int H = 6144 * 3, W = 8192;  // three padded planes, like C2
  const long n = (long)H * W;
  Ptrs p;
  for (int k = 0; k < 8; ++k) { float* q; CK(hipMalloc(&q, n * 4)); CK(hipMemset(q, 0, n * 4)); p.in[k] = q; }
  for (int k = 0; k < 7; ++k) { CK(hipMalloc(&p.out[k], n * 4)); }
  hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  const double gb = 15.0 * n * 4 / 1e9;
  auto report = [&](const char* name, float ms) { printf("%-28s %.3f ms  %.0f GB/s\n", name, ms, gb / (ms * 1e-3)); };
  for (int rep = 0; rep < 2; ++rep) {
    float ms;
    CK(hipEventRecord(a)); for (int i = 0; i < 10; ++i) k_tile_scalar<<<(H / 16) * (W / 64), 256>>>(p, H, W, W / 64); CK(hipEventRecord(b)); CK(hipEventSynchronize(b)); CK(hipEventElapsedTime(&ms, a, b)); report("scalar 16x64 tiles", ms / 10);
    CK(hipEventRecord(a)); for (int i = 0; i < 10; ++i) k_tile_vec4<4><<<(H / 4) * (W / 256), 256>>>(p, H, W, W / 256); CK(hipEventRecord(b)); CK(hipEventSynchronize(b)); CK(hipEventElapsedTime(&ms, a, b)); report("float4 4x256 tiles", ms / 10);
    CK(hipEventRecord(a)); for (int i = 0; i < 10; ++i) k_tile_vec4<16><<<(H / 16) * (W / 256), 256>>>(p, H, W, W / 256); CK(hipEventRecord(b)); CK(hipEventSynchronize(b)); CK(hipEventElapsedTime(&ms, a, b)); report("float4 16x256 tiles", ms / 10);
    CK(hipEventRecord(a)); for (int i = 0; i < 10; ++i) k_linear_vec4<<<2048, 256>>>(p, n / 4); CK(hipEventRecord(b)); CK(hipEventSynchronize(b)); CK(hipEventElapsedTime(&ms, a, b)); report("float4 linear grid-stride", ms / 10);
  }
  return 0;
}
