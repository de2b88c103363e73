// seq_pattern.hip -- calibration of the memory-side counters on the access pattern of C4's sequential ADMM middle
// (k_cols_mid_admm_seq, lenslesspicam_amd/csrc/lpc_kernels.h): VERDICT r04 item 2, MI355X_MICROARCH.md "HBM" -- the
// 2 x FETCH_SIZE correction is calibrated for 16-byte-lane streaming only.
//
// A workgroup of 512 lanes owns ONE tile of T = 8 image columns x N = 540 rows of a half-spectrum plane (row pitch
// 496 complex values = 3968 bytes, 64-byte row segments at 8 bytes per lane, element e = tid + 512 k -> row e / 8, column
// e % 8): it reads that tile of two arrays (SA, SB) and writes both back, columns at or beyond Wc = 481 are read but not
// written; blocks are handed out FRAMES FASTEST (block b -> frame b % frames, then column tile, then PSF plane), exactly
// like the kernel.  No arithmetic besides a scale, no LDS: the bytes are known exactly.
//   mode 0  the pattern above                                     unique bytes: read 2 * P * N * 488 * 8, written 2 * P * N * 481 * 8
//   mode 1  + the shared H tile of the plane's PSF channel (read) + 3 * N * 488 * 8 unique (every XCD fetches its own copy)
//   mode 2  the same four streams as whole planes, 16 bytes per lane, fully coalesced (the guide's calibrated pattern)
//   mode 3  mode 0 with tiles of 16 columns (128-byte row segments)
// usage: seq_pattern <mode> [frames=64] [reps=20]      (run under rocprofv3 --pmc ...; prints the byte counts)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

constexpr int N = 540, PITCH = 496, WC = 481, DC = 3;

template <int T, bool WITH_H>
__global__ __launch_bounds__(512) void k_tiles(float2* __restrict__ SA, float2* __restrict__ SB,
                                               const float2* __restrict__ Hs, int ntile, int nfr, float sc) {
  constexpr int NELEM = N * T, EM = (NELEM + 511) / 512;
  const int tid = threadIdx.x;
  const unsigned bx = blockIdx.x;
  const int fr = (int)(bx % (unsigned)nfr), rest = (int)(bx / (unsigned)nfr);
  const int tile = rest % ntile, pp = rest / ntile;
  const long pl = (long)fr * DC + pp;
  const int c0 = tile * T;
  float2* ba = SA + pl * (long)N * PITCH + c0;
  float2* bb = SB + pl * (long)N * PITCH + c0;
  const float2* hb = Hs + (long)pp * N * PITCH + c0;
  float2 a[EM], b[EM], h[EM];
#pragma unroll
  for (int k = 0; k < EM; ++k) {
    const int e = tid + k * 512, ec = e < NELEM ? e : 0;
    b[k] = bb[(ec / T) * PITCH + ec % T];
  }
#pragma unroll
  for (int k = 0; k < EM; ++k) {
    const int e = tid + k * 512, ec = e < NELEM ? e : 0;
    a[k] = ba[(ec / T) * PITCH + ec % T];
  }
  if (WITH_H) {
#pragma unroll
    for (int k = 0; k < EM; ++k) {
      const int e = tid + k * 512, ec = e < NELEM ? e : 0;
      h[k] = hb[(ec / T) * PITCH + ec % T];
    }
  }
#pragma unroll
  for (int k = 0; k < EM; ++k) {
    const int e = tid + k * 512;
    if (e < NELEM && c0 + e % T < WC) {
      float2 x = a[k], y = b[k];
      if (WITH_H) { x.x += h[k].x * sc; y.y += h[k].y * sc; }
      ba[(e / T) * PITCH + e % T] = make_float2(x.x * sc, x.y);
      bb[(e / T) * PITCH + e % T] = make_float2(y.x * sc, y.y);
    }
  }
}

__global__ __launch_bounds__(256) void k_stream(float4* __restrict__ SA, float4* __restrict__ SB, long n4, float sc) {
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long)gridDim.x * 256) {
    float4 a = SA[i], b = SB[i];
    a.x *= sc; b.x *= sc;
    SA[i] = a; SB[i] = b;
  }
}

int main(int argc, char** argv) {
  const int mode = argc > 1 ? std::atoi(argv[1]) : 0;
  const int frames = argc > 2 ? std::atoi(argv[2]) : 64;
  const int reps = argc > 3 ? std::atoi(argv[3]) : 20;
  const int P = frames * DC;
  const size_t plane = (size_t)N * PITCH, bytes = plane * P * sizeof(float2);
  float2 *SA, *SB, *Hs;
  CK(hipMalloc(&SA, bytes)); CK(hipMalloc(&SB, bytes)); CK(hipMalloc(&Hs, plane * DC * sizeof(float2)));
  CK(hipMemset(SA, 0, bytes)); CK(hipMemset(SB, 0, bytes)); CK(hipMemset(Hs, 0, plane * DC * sizeof(float2)));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const int T = mode == 3 ? 16 : 8, ntile = (WC + T - 1) / T;
  auto launch = [&]() {
    if (mode == 0) hipLaunchKernelGGL((k_tiles<8, false>), dim3(ntile * P), dim3(512), 0, 0, SA, SB, Hs, ntile, frames, 1.0001f);
    else if (mode == 1) hipLaunchKernelGGL((k_tiles<8, true>), dim3(ntile * P), dim3(512), 0, 0, SA, SB, Hs, ntile, frames, 1.0001f);
    else if (mode == 3) hipLaunchKernelGGL((k_tiles<16, false>), dim3(ntile * P), dim3(512), 0, 0, SA, SB, Hs, ntile, frames, 1.0001f);
    else hipLaunchKernelGGL(k_stream, dim3(4096), dim3(256), 0, 0, (float4*)SA, (float4*)SB, (long)(bytes / 16), 1.0001f);
  };
  for (int i = 0; i < 3; ++i) launch();
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0, 0));
  for (int i = 0; i < reps; ++i) launch();
  CK(hipEventRecord(e1, 0));
  CK(hipDeviceSynchronize());
  float ms = 0;
  CK(hipEventElapsedTime(&ms, e0, e1));
  ms /= reps;
  double rd, wr;
  if (mode == 2) { rd = 2.0 * bytes; wr = 2.0 * bytes; }
  else {
    const int cols_read = ntile * T;                  // 488 (T = 8) / 496 (T = 16) columns are loaded
    rd = 2.0 * P * N * cols_read * 8.0 + (mode == 1 ? (double)DC * N * cols_read * 8.0 : 0.0);
    wr = 2.0 * P * N * WC * 8.0;
  }
  std::printf("mode %d frames %d: %.4f ms per launch; unique bytes read %.4f GB written %.4f GB total %.4f GB -> %.2f TB/s\n",
              mode, frames, ms, rd / 1e9, wr / 1e9, (rd + wr) / 1e9, (rd + wr) / ms / 1e9);
  CK(hipGetLastError());
  return 0;
}
