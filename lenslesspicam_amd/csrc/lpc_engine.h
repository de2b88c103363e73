// lpc_engine.h -- what the translation units of the engine share: the handle, error plumbing, the launcher and the
// workgroup-shape dispatchers, and the host functions that cross translation units.
//
// The library is split so that the device compiler works on several units in parallel:
//   lpc_engine.cpp  plans, geometry, HBM workspace, the C ABI, the image-domain ADMM kernels, set-up / layout /
//                   evaluation / preparation kernels
//   lpc_rows.cpp    every row-pass launch (real <-> half-spectrum transforms, incl. the fused ADMM rows)
//   lpc_cols.cpp    every column-pass launch (pass A, the fused middles)
//   lpc_gd.cpp, lpc_gd_update.cpp, lpc_gd_update_fwd.cpp   the gradient-descent family's fused row kernels
//   lpc_jit.cpp     plan modules: find / compile / load (lpc_plan.h)
//   lpc_module.cpp  NOT part of the library: the source of a plan module (compile-time-plan kernels of one frame shape)
#pragma once
#include "lpc_kernels.h"
#include "lpc.h"
#include "lpc_plan.h"

#ifndef LPC_SRC_FP
#define LPC_SRC_FP "dev"     // fingerprint of the sources (build.py): a module must be built from the same ones
#endif

#include <algorithm>
#include <climits>
#include <string>
#include <type_traits>
#include <unordered_set>
#include <vector>

// --------------------------------------------------------------------------- errors --
int fail(const std::string& msg);   // records the message for lpc_last_error() (thread-local), returns 1
#define LPC_RT(expr)                                                                      \
  do {                                                                                    \
    lpcError_t e_ = (expr);                                                               \
    if (e_ != lpcSuccess)                                                                 \
      return fail(std::string(#expr) + " failed: " + rt::err_string(e_));                 \
  } while (0)
#define LPC_OK(expr)          \
  do {                        \
    int r_ = (expr);          \
    if (r_) return r_;        \
  } while (0)


// Workgroup shape for an FFT tile of `nelem` complex points: NT threads x EMAX points per thread:
// the fewest threads that hold the tile with <= 16 points per thread.  Measured on MI355X
// (profiles/r01b_notes.md): the alternatives "twice the threads, half the points" (same LDS, twice
// the waves) and "half the threads, 32 points" are both slower.
// LDS holds 160 KiB per workgroup: 16384 complex64 points (128 KiB) or 8192 complex128 points
static constexpr int kMaxTilePoints = (int)(131072 / sizeof(real2));
template <class F>
static inline int dispatch_cfg(int nelem, F&& f) {
  using std::integral_constant;
  if (nelem <= 1024) return f(integral_constant<int, 256>{}, integral_constant<int, 4>{});
  if (nelem <= 2048) return f(integral_constant<int, 256>{}, integral_constant<int, 8>{});
  if (nelem <= 4096) return f(integral_constant<int, 256>{}, integral_constant<int, 16>{});
  if (nelem <= 8192) return f(integral_constant<int, 512>{}, integral_constant<int, 16>{});
  if (nelem <= kMaxTilePoints) return f(integral_constant<int, 1024>{}, integral_constant<int, 16>{});
  return fail("FFT tile of " + std::to_string(nelem) + " points exceeds the LDS budget (" +
              std::to_string(kMaxTilePoints) + ")");
}

// row kernels: (NT, EMAX) by row length, the LDS-skew flag and the radix-2-folding flag of the plan
template <class F>
static inline int dispatch_row(int Wp, int skew, bool r2, F&& f) {
  using std::integral_constant;
  return dispatch_cfg(Wp, [&](auto NT, auto EM) {
    if (skew && r2) return f(NT, EM, integral_constant<bool, true>{}, integral_constant<bool, true>{});
    if (skew) return f(NT, EM, integral_constant<bool, true>{}, integral_constant<bool, false>{});
    if (r2) return f(NT, EM, integral_constant<bool, false>{}, integral_constant<bool, true>{});
    return f(NT, EM, integral_constant<bool, false>{}, integral_constant<bool, false>{});
  });
}

struct KernelTimer {
#if !defined(LPC_SIMT_EMU)
  std::vector<std::pair<hipEvent_t, hipEvent_t>> ev[LPC_K_COUNT];
  size_t used[LPC_K_COUNT] = {0};
#endif
  bool on = false;
  unsigned mask = ~0u;      // bit k: launches of kernel id k are bracketed (lpc_profile_enable)
};

struct lpc_engine {
  lpc_config cfg{};
  PlaneGeom g{};
  int N1 = 1, N2 = 1;  // column split Hp = N1*N2 (N1 == 1: single pass)
  int T = 16;          // image columns per column-pass tile
  Fft1dPlan planW{}, planA{}, planB{};
  Fft1dPlan planWi{};   // inverse-row plan with the radix-2 stage FIRST (rows_r2 only)
  Fft1dPlan planWh{};   // length Wp/2: ADMM rows, one real row per half-length transform (rows_half)
  bool rows_half = false;
  EngineOpts opt;          // lpc_config::options
  PlanSpec spec;           // the compile-time-plan kernels this handle runs (lpc_plan.h) ...
  const struct LpcModule* mod = nullptr;   // ... and the loaded plan module that holds them (null: run-time plans only)
  std::string mod_note;    // why there is no module, for lpc_plan_info
  bool xhalf_rows = false; // ADMM: xi / a = mu1 X - xi computed by the forward row kernel of the module
  bool k1_rows = false;    // ... and the TV / W half too: three launches per iteration (small frames, option k1_rows)
  bool xi_window = false;  // ... which then skips xi / HV_old outside the sensor window (AdmmScalars::xiw)
  bool hv_skip = false;    // ... and rows wholly outside it skip the H V row transforms in both directions (AdmmScalars::skipa)
  bool mid_reg = true;  // register-resident fused middle where the pass-B length allows (option mid_lds=1: off)
  bool rows_r2 = false; // row plans end in a radix-2 stage: fold it into the Hermitian (un)tangling
  ColPass passA{}, passB{};
  int P = 0, Ppsf = 0, Pdata = 0;
  std::vector<void*> allocs;
  size_t total_bytes = 0;

  // spectral constants
  real2* Hs = nullptr;     // [Ppsf] PSF spectrum, permuted row order, norm applied
  real2* Hs_t = nullptr;   // ... and |G| below: copies in the pair-line layout for the module's 8-column middle (PlaneGeom::slay)
  real* Gabs_t = nullptr;
  // ... and the sequential middle's point-wise constants, precombined per (PSF, step sizes) (lpc_kernels.h: k_mid_consts)
  void* midc = nullptr;       // MidConst (mid_pc 1) or real2 (mid_pc 2: real phases) per element
  real* midrd = nullptr;
  double midc_par[3] = {0, 0, 0};   // the step sizes the tables were made for
  bool midc_valid = false;
  real* Gabs = nullptr;    // ADMM: |PsiT Psi| spectrum, ONE plane (identical for every channel)
  // ... and, when that plane is a sum of a row term and a column term (the reference's finite-difference gram is:
  // (2 - 2 cos th_r) + (2 - 2 cos th_c)), the two vectors the middles read instead of it: Ga[row] + Gb[col]
  real* Ga = nullptr;      // [Hp], the engine's (permuted) spectrum row order
  real* Gb = nullptr;      // [cpitch]
  real* Gpart = nullptr;   // partial maxima of the separability check
  int g_sep = 0;
  std::vector<double> sched[4];  // optional per-iteration mu1, mu2, mu3, tau (unrolled ADMM)
  double last_par[4] = {0, 0, 0, 0};  // parameters of the most recent iteration
  real2* phr = nullptr;    // [Hp] ifftshift phase, stored row order
  real2* phc = nullptr;    // [Wc]
  real2* twH = nullptr;
  real2* tws_row = nullptr;   // stage twiddles of the module's row plan in lane order (lpc_sfft.h: SPlan::tws_off)
  // work spectra: [2][P] planes (ADMM uses both halves, others the first)
  real2* S = nullptr;
  // ADMM state (padded real planes)
  real *V[2] = {nullptr, nullptr}, *HVb[2] = {nullptr, nullptr}, *xi = nullptr, *rho = nullptr,
        *Rsp = nullptr, *Aarr = nullptr;
  real *eta0[2] = {nullptr, nullptr}, *eta1[2] = {nullptr, nullptr};  // ping-pong (halo reads)
  int vcur = 0, ecur = 0, hcur = 0;  // HVb[hcur] = H V of the current estimate, HVb[hcur^1] = of the previous one
  // the reference clamps the image estimate IN PLACE whenever _form_image runs (admm.py:331-338); only the W-update ever
  // sees that clamped copy, and it is a pure function of V: vw_cur = the next iteration's W sees clamp(V),
  // vw_old = the previous iteration's W saw clamp(V_old) (needed to recompute W_old) -- AdmmScalars::clamp_cur / _old
  bool vw_cur = false, vw_old = false;
  // GD family state (un-padded planes)
  real *gx = nullptr, *gaux = nullptr;  // x and (p | xk_prev)
  real* galpha = nullptr;               // [C] device
  real* gx0 = nullptr;                  // [C] default start value per channel
  real2* S2 = nullptr;                  // second spectrum buffer (row-inverse+forward is out of place)
  double tk = 1.0, nest_mu = 0.9, nest_p = 0.0;
  // unrolled FISTA (unrolled_fista.py:91-106): per-iteration step alpha[i][c] and momentum factor coef[i]
  std::vector<real> fista_coef;
  real* galpha_sched = nullptr;  // device [n][C]
  size_t galpha_sched_cap = 0;   // elements allocated for it (re-used by later schedules that fit)
  int fista_sched_n = 0;
  // common
  real* Y = nullptr;         // data planes, un-padded [Pdata][H][W]
  real* init_est = nullptr;  // planar copy of the initial estimate (or null)
  real* psf_planar = nullptr;
  bool has_init = false, psf_set = false, data_set = false, first = true;
  bool gd_fwd_done = false;    // the row spectra of H x's input are already in S (written by the fused update kernel)
  bool gd_fuse_fwd = false;    // gradient-descent family: update kernel + next forward rows in one launch
  bool gd_v2 = false;          // ... its two fused row kernels in their second form (lpc_gd_v2_kernels.h; option gd_v2)
  bool split_pending = false;  // lpc_iterate_begin ran, lpc_iterate_end has not yet
  // plug-and-play ADMM (lpc_admm_pnp_begin / _end): explicit state in the arrays the fused path uses for the TV duals
  //   eta0[0] = eta, eta1[0] = U, eta0[1] = X, eta1[1] = W   (all image-shaped)
  bool pnp_mode = false, pnp_pending = false;
  long iters_done = 0;
  KernelTimer timer;
  lpcStream_t stream = nullptr;
};
typedef lpc_engine Engine;

template <class Tp>
static inline int dev_alloc(Engine* e, Tp** out, size_t count) {
  void* p = nullptr;
  size_t bytes = count * sizeof(Tp);
  LPC_RT(rt::dev_malloc(&p, bytes));
  e->allocs.push_back(p);
  e->total_bytes += bytes;
  *out = (Tp*)p;
  return 0;
}

// hipFuncAttributeMaxDynamicSharedMemorySize is a per-DEVICE attribute of a kernel: the (device, function) pairs that have
// it live in the CORE library (lpc_engine.cpp) -- launch_k is instantiated inside every plan module too, and a
// thread_local set there would register a TLS destructor that pins the module: dlclose() would never unload it
int big_smem_once(const void* fn, size_t smem);

// generic launcher (+ optional event bracketing of hot-loop kernels)
template <class K, class... A>
static inline int launch_k(Engine* e, int kid, K kernel, dim3 grid, int nt, size_t smem, A... args) {
  if (smem > 48 * 1024) LPC_OK(big_smem_once((const void*)kernel, smem));
#if !defined(LPC_SIMT_EMU)
  const bool timed = e->timer.on && kid >= 0 && ((e->timer.mask >> kid) & 1u);
  size_t slot = 0;
  if (timed) {
    auto& v = e->timer.ev[kid];
    slot = e->timer.used[kid]++;
    if (slot >= v.size()) {
      hipEvent_t a, b;
      LPC_RT(hipEventCreate(&a));
      LPC_RT(hipEventCreate(&b));
      v.push_back({a, b});
    }
    LPC_RT(hipEventRecord(v[slot].first, e->stream));
  }
  hipLaunchKernelGGL(kernel, grid, dim3(nt), smem, e->stream, args...);
  if (timed) LPC_RT(hipEventRecord(e->timer.ev[kid][slot].second, e->stream));
#else
  (void)kid;
  lpc_emu::launch(grid, dim3(nt), smem, [=]() { kernel(args...); });
#endif
  LPC_RT(rt::last_error());
  return 0;
}

static inline dim3 grid1d(long n, int nt, long planes = 1) {
  long b = (n + nt - 1) / nt;
  if (b > 4096) b = 4096;
  if (b < 1) b = 1;
  return dim3((unsigned)b, (unsigned)planes, 1);
}

static inline RealSrc src_unpadded(const Engine* e, const real* base) {
  const PlaneGeom& g = e->g;
  RealSrc s;
  s.base = base; s.plane_stride = g.uplane; s.pitch = g.W; s.nrows = g.H; s.ncols = g.W; s.col0 = g.sw;
  s.out_row0 = g.sh;
  return s;
}
static inline RealSrc src_padded(const Engine* e, const real* base) {
  const PlaneGeom& g = e->g;
  RealSrc s;
  s.base = base; s.plane_stride = g.rplane; s.pitch = g.rpitch; s.nrows = g.Hp; s.ncols = g.Wp; s.col0 = 0;
  s.out_row0 = 0;
  return s;
}
static inline RealDst dst_padded(const Engine* e, real* base) {
  const PlaneGeom& g = e->g;
  RealDst d;
  d.base = base; d.plane_stride = g.rplane; d.pitch = g.rpitch; d.nrows = g.Hp; d.row0 = 0; d.col0 = 0;
  d.ncols = g.Wp;
  return d;
}
static inline RealDst dst_cropped(const Engine* e, real* base) {
  const PlaneGeom& g = e->g;
  RealDst d;
  d.base = base; d.plane_stride = g.uplane; d.pitch = g.W; d.nrows = g.H; d.row0 = g.sh; d.col0 = g.sw;
  d.ncols = g.W;
  return d;
}

// ---- plan module: launchers of the compile-time-plan kernels of one frame shape (lpc_module.cpp) -------------------
// An entry is null when the module does not hold that kernel; the core then launches its run-time-plan kernel.
struct GdScalars;
struct LpcModule {
  int (*rows_fwd_single)(Engine*, const RealSrc*, real2* S, int nplanes, int kid);
  int (*rows_inv_single)(Engine*, const real2* S, const RealDst*, int nplanes, int kid);
  int (*admm_rows_fwd)(Engine*);
  int (*admm_rows_fwd_x)(Engine*, const AdmmScalars*, const K1Rows* k1);   // k1: + the TV / W half (k1_rows)
  int (*admm_rows_inv)(Engine*, real* Vout, real* HVout, int skip_hv_outside);
  int (*gd_rows_mid)(Engine*);
  int (*gd_rows_update)(Engine*, const GdScalars*, const real* alpha);
  int (*gd_rows_update_fwd)(Engine*, const GdScalars*, const real* alpha);
  int (*cols_passA)(Engine*, const ColPass*, real2* S, int nplanes, int inverse, int kid);
  int (*admm_mid)(Engine*, const ColPass*, const AdmmScalars*, real sb_outside_scale);
  int k1_rows;    // admm_rows_fwd_x takes the TV / W half of the image-domain work as well (k_rfwd_arrays_x<.., K1>)
  int mid_pc;     // its sequential middle reads the precombined constants (Engine::midc / midrd)
  int slay;       // its ADMM row kernels and fused middle keep the work spectra in pair lines (PlanSpec::slay)
  int gd_v2;      // the module holds k_gd_resid_v2 / k_gd_update_fwd_v2 for its row plan (lpc_gd_v2_kernels.h)
};
// lpc_jit.cpp: the module of `spec` -- from the process cache, from disk, or (allow_compile) compiled now; null + `why`
const LpcModule* get_plan_module(const PlanSpec& spec, const EngineOpts& opt, bool allow_compile, std::string* why);
void release_plan_module(const LpcModule* mod);     // a handle that got a module from get_plan_module is done with it
int build_plan_module(const PlanSpec& spec, const EngineOpts& opt, std::string* path_or_error);   // compile only (no load); no-op when the module is on disk

// ---- host functions that cross translation units ------------------------------------------------------------
// lpc_rows.cpp
int rows_fwd_single(Engine* e, const RealSrc& src, real2* S, int nplanes, int kid);
int rows_inv_single(Engine* e, const real2* S, const RealDst& dst, int nplanes, int kid);
int admm_rows_fwd(Engine* e);                                   // e->Rsp, e->Aarr -> the two work spectra
int admm_rows_fwd_x(Engine* e, const AdmmScalars& sc, const K1Rows* k1 = nullptr);          // e->Rsp and (xi, HV, HV_old, y) -> the two work spectra
int admm_rows_inv(Engine* e, real* Vout, real* HVout, bool skip_hv_outside = false);          // the two work spectra -> V, H V
// lpc_cols.cpp
int cols_passA(Engine* e, real2* S, int nplanes, bool inverse, int zr0, int zr1, int kid, bool crop_rows_only = false,
               real sb_outside_scale = (real)0.);
int cols_passB_fwd(Engine* e, real2* S, int nplanes, int zr0, int zr1);
int conv_middle(Engine* e, real2* S, int nplanes, bool adjoint, int zr0, int zr1, bool crop_rows_only = false);
int admm_cols(Engine* e, const AdmmScalars& sc);   // sc.skipa: forward pass A rescales the kept rows of SB                // [pass A] -> fused ADMM middle -> [inverse pass A]
// lpc_gd.cpp, lpc_gd_update.cpp, lpc_gd_update_fwd.cpp (one kernel family each)
int gd_rows_mid(Engine* e);                                     // irfft rows -> residual -> rfft rows (S -> S2)
int gd_rows_update(Engine* e, const GdScalars& sc, const real* alpha);   // irfft rows -> fused projected update
int gd_rows_update_fwd(Engine* e, const GdScalars& sc, const real* alpha);   // ... -> next iteration's forward rows (S)
