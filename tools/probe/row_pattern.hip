// row_pattern.hip -- what does HBM deliver for the access pattern of the half-length row kernels, without any arithmetic?
// One workgroup of 256 lanes per row: reads the row's 4097 complex64 (32.8 KB, pitch 4112), optionally a second time in
// mirrored order (the tangling of lpc_gd_v2_kernels.h), optionally 16 KB of a real row (y), idles for `sleep` x 2048
// cycles (the two transforms), writes 32.8 KB to a second array.  9120 rows (3040 x 3), like C3's residual rows.
//   mode bit 0: 16-byte lanes (dwordx4) instead of 8-byte lanes   bit 1: mirrored second read   bit 2: y row
//   bit 3: + 32 more 8-byte loads per lane from a 32-KB table shared by all workgroups (the twiddle loads: L1 / L2 hits)
// usage: row_pattern <mode> <sleep units> [reps]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
constexpr int M = 4096, PITCH = 4112, ROWS = 9120, W = 4056;

template <int MODE>
__global__ __launch_bounds__(256) void k_rows(const float2* __restrict__ Sin, float2* __restrict__ Sout,
                                              const float* __restrict__ Y, int sleep, const float2* __restrict__ TW) {
  const int j = threadIdx.x;
  const float2* in = Sin + (size_t)blockIdx.x * PITCH;
  float2* out = Sout + (size_t)blockIdx.x * PITCH;
  float2 a[16], b[16], y[8];
  float acc = 0.f;
  if (MODE & 1) {
#pragma unroll
    for (int m = 0; m < 8; ++m) {
      const float4 t = ((const float4*)in)[j + 256 * m];
      a[2 * m] = make_float2(t.x, t.y); a[2 * m + 1] = make_float2(t.z, t.w);
    }
  } else {
#pragma unroll
    for (int m = 0; m < 16; ++m) a[m] = in[j + 256 * m];
  }
  if (MODE & 2) {
#pragma unroll
    for (int m = 0; m < 16; ++m) b[m] = in[M - j - 256 * m];
  }
  if (MODE & 4) {
    const float* yr = Y + (size_t)blockIdx.x * W;
#pragma unroll
    for (int m = 0; m < 8; ++m) {
      const int c = 2 * (j + 256 * m);
      y[m] = c < W ? *(const float2*)(yr + c) : make_float2(0.f, 0.f);
    }
  }
  if (MODE & 2) {
#pragma unroll
    for (int m = 0; m < 16; ++m) acc += b[m].x - b[m].y;
  }
  if (MODE & 8) {
    float2 t[32];
#pragma unroll
    for (int m = 0; m < 32; ++m) t[m] = TW[(j * (m + 1)) & 4095];
#pragma unroll
    for (int m = 0; m < 32; ++m) acc += t[m].x * t[m].y;
  }
  if (MODE & 4) {
#pragma unroll
    for (int m = 0; m < 8; ++m) acc += y[m].x + y[m].y;
  }
#pragma unroll
  for (int m = 0; m < 16; ++m) a[m].x += acc * 1e-30f;
  for (int i = 0; i < sleep; ++i) __builtin_amdgcn_s_sleep(32);
  if (MODE & 1) {
#pragma unroll
    for (int m = 0; m < 8; ++m) ((float4*)out)[j + 256 * m] = make_float4(a[2 * m].x, a[2 * m].y, a[2 * m + 1].x, a[2 * m + 1].y);
  } else {
#pragma unroll
    for (int m = 0; m < 16; ++m) out[j + 256 * m] = a[m];
  }
  if (j == 0) out[M] = a[0];
}

int main(int argc, char** argv) {
  const int mode = argc > 1 ? std::atoi(argv[1]) : 0, sleep = argc > 2 ? std::atoi(argv[2]) : 0, reps = argc > 3 ? std::atoi(argv[3]) : 20;
  float2 *A, *B, *TW; float* Y;
  const size_t sb = (size_t)ROWS * PITCH * sizeof(float2);
  CK(hipMalloc(&A, sb)); CK(hipMalloc(&B, sb)); CK(hipMalloc(&Y, (size_t)ROWS * W * 4)); CK(hipMalloc(&TW, 4096 * 8)); CK(hipMemset(TW, 0, 4096 * 8));
  CK(hipMemset(A, 0, sb)); CK(hipMemset(B, 0, sb)); CK(hipMemset(Y, 0, (size_t)ROWS * W * 4));
  auto launch = [&]() {
    switch (mode) {
      case 0: hipLaunchKernelGGL(k_rows<0>, dim3(ROWS), dim3(256), 0, 0, A, B, Y, sleep, TW); break;
      case 1: hipLaunchKernelGGL(k_rows<1>, dim3(ROWS), dim3(256), 0, 0, A, B, Y, sleep, TW); break;
      case 2: hipLaunchKernelGGL(k_rows<2>, dim3(ROWS), dim3(256), 0, 0, A, B, Y, sleep, TW); break;
      case 3: hipLaunchKernelGGL(k_rows<3>, dim3(ROWS), dim3(256), 0, 0, A, B, Y, sleep, TW); break;
      case 4: hipLaunchKernelGGL(k_rows<4>, dim3(ROWS), dim3(256), 0, 0, A, B, Y, sleep, TW); break;
      case 5: hipLaunchKernelGGL(k_rows<5>, dim3(ROWS), dim3(256), 0, 0, A, B, Y, sleep, TW); break;
      case 6: hipLaunchKernelGGL(k_rows<6>, dim3(ROWS), dim3(256), 0, 0, A, B, Y, sleep, TW); break;
      case 7: hipLaunchKernelGGL(k_rows<7>, dim3(ROWS), dim3(256), 0, 0, A, B, Y, sleep, TW); break;
      case 14: hipLaunchKernelGGL(k_rows<14>, dim3(ROWS), dim3(256), 0, 0, A, B, Y, sleep, TW); break;
      case 12: hipLaunchKernelGGL(k_rows<12>, dim3(ROWS), dim3(256), 0, 0, A, B, Y, sleep, TW); break;
      default: hipLaunchKernelGGL(k_rows<8>, dim3(ROWS), dim3(256), 0, 0, A, B, Y, sleep, TW); break;
    }
  };
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int i = 0; i < 3; ++i) launch();
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0, 0));
  for (int i = 0; i < reps; ++i) launch();
  CK(hipEventRecord(e1, 0));
  CK(hipDeviceSynchronize());
  float ms = 0;
  CK(hipEventElapsedTime(&ms, e0, e1));
  ms /= reps;
  const double bytes = (double)ROWS * (2.0 * (M + 1) * 8 + ((mode & 4) ? W * 4.0 : 0.0));
  std::printf("mode %d (%s lanes%s%s) sleep %d: %.4f ms per launch, %.3f GB unique -> %.2f TB/s\n", mode, (mode & 1) ? "16-byte" : "8-byte",
              (mode & 2) ? ", mirrored re-read" : "", (mode & 4) ? ((mode & 8) ? ", y row, +32 table loads" : ", y row") : ((mode & 8) ? ", +32 table loads" : ""), sleep, ms, bytes / 1e9, bytes / ms / 1e9);
  CK(hipGetLastError());
  return 0;
}
