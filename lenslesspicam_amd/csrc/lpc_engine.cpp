// lpc_engine.cpp -- host side of the engine: plans, HBM workspace, launch sequences and the
// C ABI declared in include/lpc.h.  Compiled as HIP for gfx950 (product) or, for the CPU
// test-suite only, as plain C++ with -DLPC_SIMT_EMU (see lpc_rt.h).  The row / column / gradient-descent
// launches live in lpc_rows.cpp, lpc_cols.cpp and lpc_gd.cpp (see lpc_engine.h).
#include "lpc_engine.h"
#include <mutex>
#include <unordered_map>
#include "lpc_gd_kernels.h"
#include "lpc_metric_kernels.h"
#include "lpc_prep_kernels.h"

// --------------------------------------------------------------------------- errors --
static thread_local std::string g_last_error;
int fail(const std::string& msg) {
  g_last_error = msg;
  return 1;
}

// -------------------------------------------------------------------------- helpers --
static int next_5smooth(int n) {  // scipy.fftpack.next_fast_len (rfft_convolve.py:112)
  int m = n < 1 ? 1 : n;
  for (;; ++m) {
    int r = m;
    while (r % 2 == 0) r /= 2;
    while (r % 3 == 0) r /= 3;
    while (r % 5 == 0) r /= 5;
    if (r == 1) return m;
  }
}


// ------------------------------------------------------------------------ FFT plans --
static int upload(Engine* e, void* dst, const void* src, size_t bytes) {
  LPC_RT(rt::copy_h2d_async(dst, src, bytes, e->stream));
  LPC_RT(rt::stream_sync(e->stream));
  return 0;
}

int big_smem_once(const void* fn, size_t smem) {
  static std::mutex mu;
  static std::unordered_map<uint64_t, size_t> granted;      // (function, device) -> dynamic LDS the kernel may use
  int dev = 0;
  LPC_RT(rt::current_device(&dev));
  const uint64_t key = (uint64_t)(uintptr_t)fn ^ ((uint64_t)(dev + 1) << 56);
  std::lock_guard<std::mutex> lock(mu);
  size_t& have = granted[key];
  if (have < smem) {            // first launch, or a later one (another handle, a wider tile) that needs more
    // (a kernel with static LDS of its own -- the stamped timing builds, lpc_rt.h: LPC_STAMP -- cannot have the whole
    // 160 KiB as dynamic LDS: ask for what this launch needs then)
    size_t want = smem > 65536 ? (size_t)160 * 1024 : (size_t)65536;
    if (rt::set_max_dyn_smem(fn, want) != lpcSuccess) {
      (void)rt::last_error();
      want = smem;
      LPC_RT(rt::set_max_dyn_smem(fn, want));
    }
    have = want;
  }
  return 0;
}

static int make_twiddles(Engine* e, int n, real2** out) {
  std::vector<real2> h((size_t)std::max(n, 1));
  for (int q = 0; q < n; ++q) {
    const double a = -2.0 * M_PI * (double)q / (double)n;
    h[q] = make_real2((real)std::cos(a), (real)std::sin(a));
  }
  LPC_OK(dev_alloc(e, out, h.size()));
  return upload(e, *out, h.data(), h.size() * sizeof(real2));
}

// stage twiddles of a compile-time plan in lane order: layout and purpose in lpc_sfft.h (SPlan::tws_off)
static int make_stage_twiddles(Engine* e, const StaticFft& f, real2** out) {
  *out = nullptr;
  std::vector<real2> h;
  int ns = f.rad[0];
  auto w = [&](long q) {
    const double a = -2.0 * M_PI * (double)(q % f.n) / (double)f.n;
    return make_real2((real)std::cos(a), (real)std::sin(a));
  };
  for (int st = 1; st < f.nst; ++st) {
    const int R = f.rad[st], nb = f.n / R, step = f.n / (ns * R);
    const size_t base = h.size();
    if (R == 8 || R == 16) {         // base powers {q, 2q}, {4q, 8q}: [pair][j][2]
      h.resize(base + (size_t)4 * nb);
      for (int hh = 0; hh < 2; ++hh)
        for (int j = 0; j < nb; ++j)
          for (int i = 0; i < 2; ++i)
            h[base + ((size_t)hh * nb + j) * 2 + i] = w((long)(j % ns) * step * (1L << (2 * hh + i)));
    } else {                         // every power: [m - 1][j]
      h.resize(base + (size_t)(R - 1) * nb);
      for (int m = 1; m < R; ++m)
        for (int j = 0; j < nb; ++j) h[base + (size_t)(m - 1) * nb + j] = w((long)(j % ns) * step * m);
    }
    ns *= R;
  }
  if (h.empty()) return 0;
  LPC_OK(dev_alloc(e, out, h.size()));
  return upload(e, *out, h.data(), h.size() * sizeof(real2));
}

static int plan_from_radices(Engine* e, Fft1dPlan& p, int n, const std::vector<int>& rad) {
  p.n = n;
  p.nst = 0;
  if ((int)rad.size() > LPC_MAX_STAGES) return fail("too many FFT stages");
  int ns = 1;
  for (size_t s = 0; s < rad.size(); ++s) {
    p.radix[s] = rad[s];
    p.ns[s] = ns;
    p.nsdiv[s] = make_fastdiv((unsigned)ns);
    p.twstep[s] = n / (ns * rad[s]);
    ns *= rad[s];
  }
  if (ns != n) return fail("internal: radices do not multiply to the length");
  p.nst = (int)rad.size();
  p.skew_ok = 1;  // see lpc_fft.h: every butterfly stride must be a multiple of 8
  for (int st = 0; st < p.nst; ++st) {
    const int nb = n / p.radix[st];
    if (nb % 8 != 0) p.skew_ok = 0;
    if (!(p.ns[st] % 8 == 0 || (p.ns[st] == 1 && p.radix[st] % 8 == 0))) p.skew_ok = 0;
  }
  // diagnostic only (results are garbage): no butterflies at all, every pass degenerates to "tile in, tile out"
  // through LDS -- times the memory access pattern of the passes alone (profiles/r01b_notes.md)
#ifdef LPC_DEBUG_KNOBS   // never in the product build: the results are garbage by construction
  if (std::getenv("LPC_DEBUG_NOFFT")) p.nst = 0;
#endif
  real2* tw = nullptr;
  LPC_OK(make_twiddles(e, n, &tw));
  p.tw = tw;
  return 0;
}

// radices of a length-n transform (5-smooth): few, fat stages -- the number of radix-6 stages (each pairs a 2 with a 3)
// that minimises the stage count; ties keep more radix-8 stages.  8s first (the twiddle-free first stage should be fat).
static bool plan_radices(int n, std::vector<int>& rad) {
  rad.clear();
  int r = n, a = 0, b3 = 0;
  while (r % 2 == 0) { r /= 2; ++a; }
  while (r % 3 == 0) { r /= 3; ++b3; }
  int c5 = 0;
  { int t = r; while (t % 5 == 0) { t /= 5; ++c5; } }
  int best_k6 = 0, best_cnt = 1 << 30;
  for (int k6 = 0; k6 <= std::min(a, b3); ++k6) {
    const int a2 = a - k6;
    const int cnt = k6 + a2 / 3 + (a2 % 3 ? 1 : 0) + (b3 - k6) + c5;
    if (cnt < best_cnt) { best_cnt = cnt; best_k6 = k6; }
  }
  a -= best_k6;
  for (int i = 0; i < a / 3; ++i) rad.push_back(8);
  a %= 3;
  for (int i = 0; i < best_k6; ++i) rad.push_back(6);
  b3 -= best_k6;
  if (a == 2) rad.push_back(4);
  if (a == 1) rad.push_back(2);
  while (r % 5 == 0) { r /= 5; rad.push_back(5); }
  for (int i = 0; i < b3; ++i) rad.push_back(3);
  return r == 1;
}

static int build_plan(Engine* e, Fft1dPlan& p, int n) {
  p.n = n;
  p.nst = 0;
  std::vector<int> rad;
  if (!plan_radices(n, rad)) return fail("length " + std::to_string(n) + " is not 5-smooth");
  return plan_from_radices(e, p, n, rad);
}

// same rule as plan_from_radices(): the i + i/8 LDS skew stays affine in every stage
static bool radices_skew_ok(int n, const std::vector<int>& rad) {
  int ns = 1;
  for (int r : rad) {
    if ((n / r) % 8 != 0) return false;
    if (!(ns % 8 == 0 || (ns == 1 && r % 8 == 0))) return false;
    ns *= r;
  }
  return true;
}

// LDS layout of a compile-time row plan (lpc_fft.h: lds_slot): i + i/8 where the plan keeps it affine.  (The conflict-free
// xor layout and i + i/16 were built and measured in rounds 4 / 5: LDS busy 47 % -> 20 %, kernel time unchanged -- the row
// kernels wait for the vector-memory path, not for LDS; profiles/HISTORY.md)
static int row_layout(int n, const std::vector<int>& rad) { return radices_skew_ok(n, rad) ? 1 : 0; }

// choose the column split Hp = N1*N2 and the tile width
static void choose_split(const EngineOpts& opt, int Hp, int Wc, int* N1, int* N2, int* T, bool prefer24 = false,
                         bool admm_f32 = false) {
  int t = 16;
  if (opt.col_t > 0) t = opt.col_t;  // option col_t
  while (t > 1 && t / 2 >= Wc) t /= 2;  // tiny images: do not waste lanes on empty columns
  int budget = kMaxTilePoints;          // points per LDS tile, worst case two arrays (ADMM middle)
  if (opt.tile_budget > 0) budget = std::max(64, opt.tile_budget);  // option tile_budget (tests)
  for (int tt = t; tt >= (t >= 8 ? 8 : t); tt /= 2) {
    if ((long)Hp * 2 * tt <= budget) { *N1 = 1; *N2 = Hp; *T = tt; return; }
    if (tt == 1) break;
  }
  int best1 = 1, best2 = Hp, bestcost = 1 << 30;
  for (int d = 1; d <= Hp; ++d) {
    if (Hp % d) continue;
    const int n2 = d, n1 = Hp / d;
    if ((long)n2 * 2 * t > budget / 2 || (long)n1 * t > budget / 2) continue;
    const int cost = std::max(n1, 2 * n2);
    // ADMM (float32), equal cost: the shorter pass A and the longer LDS middle -- 6144 rows as 96 x 64 instead of 128 x 48:
    // pass A 0.466 / 0.460 -> 0.448 / 0.434 ms, middle +0.012 ms, iteration -1.1 % (same box, three instances each)
    if (cost < bestcost || (admm_f32 && cost == bestcost && n2 > best2)) { bestcost = cost; best1 = n1; best2 = n2; }
  }
  // A 24-point pass B runs the fused middle in registers (k_cols_mid_admm_reg / k_cols_mid_mul_reg<8,3>) with the
  // fewest registers; worth it as long as pass A stays short.  Measured (r01b_notes.md): 2160 rows, 90 x 24 vs
  // 72 x 30: ADMM 72.8 vs 68.3 it/s, FISTA 2134 vs 2013 it/s; 6144 rows, 256 x 24 vs 128 x 48: ADMM 191 vs 204 it/s.
  // (ADMM: up to a 96-point pass A only -- 3072 rows run 1.6 % faster as 64 x 48 with the LDS middle than as 128 x 24,
  // 82.1 against 83.4-83.7 ms per 100 iterations of a 1520 x 2028 x 3 frame, r03z_ab.log)
  if (prefer24 && Hp % 24 == 0 && Hp / 24 <= (admm_f32 ? 96 : 128) && Hp / 24 >= 2 && (long)(Hp / 24) * t <= budget / 2) {
    best2 = 24; best1 = Hp / 24;
  }
  if (opt.split_n2 > 0) {  // option split_n2: force the length of the fused middle transform
    const int n2 = opt.split_n2;
    if (n2 > 0 && Hp % n2 == 0 && (long)n2 * 2 * t <= budget && (long)(Hp / n2) * t <= budget) { best2 = n2; best1 = Hp / n2; }
  }
  *N1 = best1; *N2 = best2; *T = t;
}

// stored row p = k1*N2 + k2  <->  frequency k = k1 + N1*k2
static inline int stored_row_freq(const Engine* e, int p) { return (p / e->N2) + e->N1 * (p % e->N2); }

// ---- the launch plan: everything that is decided once per handle, no device work -----------------------------------
static void set_static_fft(StaticFft& f, int n, const std::vector<int>& rad, int T, int nt, int em) {
  f = StaticFft{};
  if ((int)rad.size() > LPC_SPEC_MAX_ST) return;    // (cannot happen for n <= 16384 with these radices: leaves n == 0)
  f.n = n; f.nst = (int)rad.size();
  for (int i = 0; i < f.nst; ++i) f.rad[i] = rad[(size_t)i];
  f.T = T; f.nt = nt; f.em = em;
}
static inline int round_up64(int v) { return (v + 63) / 64 * 64; }
// option row_rad: "16.16.8" replaces `rad` when it is a factorisation of n into radices that have a butterfly
// (lpc_fft.h: Dft<R>)
static void override_radices(const std::string& opt, int n, std::vector<int>& rad) {
  if (opt.empty()) return;
  std::vector<int> r;
  long prod = 1;
  size_t i = 0;
  while (i < opt.size()) {
    size_t j = opt.find('.', i);
    if (j == std::string::npos) j = opt.size();
    const int v = std::atoi(opt.substr(i, j - i).c_str());
    static const int ok[] = {2, 3, 4, 5, 6, 8, 9, 10, 12, 15, 16, 18, 20, 24, 30};
    if (std::find(std::begin(ok), std::end(ok), v) == std::end(ok)) return;
    r.push_back(v);
    prod *= v;
    i = j + 1;
  }
  if (prod == n && (int)r.size() <= LPC_SPEC_MAX_ST) rad = r;
}

// compute units the launch plan is sized for: the device's when there is one, an MI355X's for the device-less
// lpc_plan_module() path (build.py pre-building modules in a container without a GPU)
static int plan_cu_count() {
#if defined(LPC_SIMT_EMU)
  return 256;       // (the emulator chooses the plans the MI355X would)
#endif
  int n = 0;
  if (rt::device_count(&n) != lpcSuccess || n <= 0) return 256;
  return rt::cu_count();
}

// `allow_static`: choose compile-time plans (-> e->spec, served by a plan module) wherever the kernels exist; false: the
// run-time plans of the core library alone.  Sets N1, N2, T, rows_half and the spec; touches nothing on the device.
static void choose_plan(Engine* e, bool allow_static) {
  const lpc_config& c = e->cfg;
  const PlaneGeom& g = e->g;
  const EngineOpts& o = e->opt;
  const bool admm = c.algo == LPC_ALGO_ADMM, f32 = sizeof(real) == 4;
  e->spec = PlanSpec{};
  e->spec.family = admm ? LPC_FAM_ADMM : LPC_FAM_GD;
  e->spec.f64 = f32 ? 0 : 1;
  e->mid_reg = true;
  // (a 24-point register middle for ADMM in float32 only: 2 x 24 complex128 values do not fit a lane's registers)
  // (the gradient-descent family keeps 128 x 48 at 6144 rows: its 48-point middle lives in registers)
  choose_split(o, g.Hp, g.Wc, &e->N1, &e->N2, &e->T, !admm || f32, admm && f32);
  // the column kernels of a plan module address their tiles with 24-bit row-index x row-step products (k_cols): the step
  // between two rows of one column transform must stay below 2^24 bytes (12 MP: 48 rows x 32.8 KB = 1.6 MB)
  const long col_step = (long)(e->N1 > 1 ? e->N2 : 1) * g.cpitch * (long)sizeof(real2);
  const bool st_cols = allow_static && col_step < (1L << 24) && g.Hp < (1 << 24) &&
                       (unsigned long long)g.Hp * g.cpitch * sizeof(real2) < (1ULL << 32);   // ... and offsets are 32-bit
  // Single-pass ADMM columns whose two-spectra tile allows only 8 image columns (DiffuserCam-sized frames, 540 padded
  // rows): the fused middle takes the two spectra one after the other through the tile (k_cols_mid_admm_seq), one
  // parked in registers while the other is transformed ... when the batch is large enough to fill the chip with
  // workgroups that each hold one spectrum (64 frames 1.20 -> 0.96 ms per launch; ONE frame 0.032 -> 0.042 ms: 93
  // workgroups for 256 CUs): four 39-KB workgroups of 512 lanes x 9 points per CU inside 64 VGPRs.  (Whole 6144-point
  // columns two at a time through the same kernel -- one launch instead of three per column step -- were built and measured
  // in round 4: 2.25 ms against 1.455 ms; 16-column tiles: 30 % slower.  profiles/HISTORY.md)
  const bool seq = admm && f32 && st_cols && e->N1 == 1 && e->T == 8 && g.Wc > 8 && (long)g.Hp * 16 <= kMaxTilePoints &&
                   o.col_t == 0 && o.mid_seq != 0 && ((long)e->P * ((g.Wc + 15) / 16) >= 512 || o.mid_seq == 1);
  // Row passes: one real row per half-length complex transform (k_r*_half kernels) once the
  // paired tile is so large that fewer than 5 workgroups fit a CU's 160 KiB of LDS.  Measured (r01b_notes.md):
  // 8192 columns +3 % it/s, 3840 columns (C5) +1.8 %; 960 columns (C4) -5 %: the short transforms leave most
  // of a 256-thread group idle.
  // The gradient-descent family switches earlier (its irfft -> residual -> rfft kernel runs two transforms per
  // workgroup): 2048 columns FISTA +6.8 %, ADMM -1 %.
  // ... and with compile-time plans at every even width: its paired-row kernels exist on run-time plans only, and a
  // half-length transform on its own plan beats them (same-box A/B, profiles/r03_notes.md: FISTA 270x480x3 3.85 -> 3.35 ms
  // per 60 iterations, 380x507x3 4.40 -> 3.75 ms).  ADMM keeps the size rule with either kind of plan (540 x 960: paired
  // 0.298 vs half 0.311 ms per 5 iterations; 768 x 1024: 0.374 vs 0.364; 3072 x 4096: paired 43.6 vs half 44.6 ms per 50).
  // Round 6: ADMM on compile-time plans keeps PAIRED rows up to 4096 columns -- paired rows take the TV / W half of the
  // image-domain work (two quads per lane: three launches per iteration, r_sp never stored), which half-length rows cannot
  // (four quads per lane: slower than the tiled kernel).  Same box, paired + fused against half-length + tiled kernel
  // (profiles/r06_notes.md): 16 x 1080p (3840 columns) 175.3 -> 170.7 ms per 20 iterations, two of its planes 54.5 -> 52.3,
  // 1520 x 2028 x 3 (4096) 30.18 -> 29.23 ms per 40; 5000 / 5760 columns -1.3 / -0.5 %, 6000 +1.4 %, 8192 (12 MP) +3 %,
  // 5120 = 8.8.8.2.5 +10 %: half-length rows above 4096.
  const bool half_ok = g.Wp % 2 == 0 && g.Wp >= 4;
  const bool admm_wide = (allow_static && g.Wp % 4 == 0 && o.k1_rows != 0) ? g.Wp > 4096
                                                                            : 5 * LPC_ROW_SMEM_BYTES(g.Wp, 1) > 160 * 1024;
  const bool wide = admm ? admm_wide : (g.Wp >= 2048 || (allow_static && g.Wp >= 128));
  e->rows_half = half_ok && wide;
  if (o.rows_half == 0) e->rows_half = false;
  if (o.rows_half == 1 && half_ok) e->rows_half = true;
  if (!allow_static) return;

  PlanSpec& sp = e->spec;
  std::vector<int> rad;
  // the X half of the image-domain work moves into the forward rows when the stencil half can run as the tiled
  // four-pixel-lane kernel (k_admm_spatial_v4<.., XHALF = false>): padded width a multiple of 4
  const bool xhalf = admm && g.Wp % 4 == 0;
  // ---- rows
  if (e->rows_half) {
    const int n = g.Wp / 2;
    plan_radices(n, rad);
    if (n == 4096) rad = {16, 16, 16};   // one butterfly per thread and stage, one LDS round trip fewer than 8.8.8.8
                                          // (same-box A/B, profiles/r02_notes.md: inverse rows 0.518 -> 0.487 ms)
    if (n == 2048 && admm) rad = {16, 16, 8};   // same idea, 256 threads x 8 points: 3072 x 4096 frames 44.4 -> 43.1 ms per
                                                 // 50 iterations (profiles/r03k_ab.log; the paired 2048-point rows of
                                                 // 1536 x 2048 frames and the 1024-point rows are faster on 8.8.8.x)
    if (n == 1920 && admm) rad = {16, 8, 15};   // 1080p frames, three stages on 128 threads x 15 points: 4 of C5's planes
                                                 // 49.1 -> 46.4 ms per 20 iterations (r03u_ab.log; 16.15.8 47.9, 24.10.8
                                                 // 48.4, 20.12.8 47.4; the gradient-descent family is FASTER on 8.8.6.5)
    override_radices(o.row_rad, n, rad);
    int nt = std::min(1024, std::max(64, round_up64(n / rad[0])));   // every lane owns a first-stage butterfly
    if (n == 2048 && admm && rad[0] == 16) nt = 256;
    // twice the lanes for 4096 = 16.16.16 (512 x 8 points: every other lane has no butterfly, but the tangling, the
    // loads and the stores get twice the waves): 12 MP FISTA 75.8 -> 73.4 ms per 40 iterations, ADMM 135.4 -> 134.3
    // (r03v_ab.log; 1024 lanes: 81.4 / 148.2); likewise the gradient-descent family's 1024 = 8.8.8.2 on 256 lanes
    // (1536 x 2048 frames: 4.43 -> 4.30 ms per 60 iterations); 2048-point rows are faster on 256 in both families
    if (n == 4096 && rad[0] == 16) nt = 512;
    if (n == 1024 && !admm) nt = 256;
    set_static_fft(sp.row, n, rad, 1, nt, (n + nt - 1) / nt);
    if (sp.row.n && sp.row.em <= 16) {
      sp.row_kind = LPC_ROWS_HALF;
      sp.row_sk = row_layout(n, rad);
      sp.row_x = xhalf;
    }
  } else if (admm) {    // paired rows: ADMM's own kernels only (set-up transforms keep the run-time plan)
    const int n = g.Wp;
    plan_radices(n, rad);
    // no folded radix-2 stage on compile-time plans: 8 ... 2 -> 4 ... 4
    if (rad.size() >= 2 && rad.back() == 2) {
      for (size_t i = rad.size() - 1; i-- > 0;)
        if (rad[i] == 8) { rad[i] = 4; rad.back() = 4; std::stable_sort(rad.begin(), rad.end(), [](int a, int b) { return (a == 8) > (b == 8); }); break; }
    }
    override_radices(o.row_rad, n, rad);
    int nt = std::min(1024, std::max(64, round_up64(n / rad[0])));
    // short rows (960 = 8.8.5.3: 120 first-stage butterflies): 128 threads x 8 points for batches, where every lane
    // then owns a butterfly of the stage that issues the global loads (forward rows 0.642 -> 0.576 ms at 64 frames);
    // ONE frame is faster on 256 x 4 (0.460 vs 0.470 ms per 5 iterations, profiles/r02_notes.md)
    // Round 5: ... unless the rows can take the TV / W half of the image-domain work as well (Engine::k1_rows: one
    // quad per lane and row, i.e. 256 lanes here) -- three launches per iteration beat the better row shape at every batch
    // size (64 frames 33.1 -> 31.0 ms per 20 iterations, 8 frames 4.27 -> 4.00 ms; profiles/r05_notes.md section 5)
    if (nt < 256 && n >= 512) {
      const bool batch = (long)e->P * g.Hp >= 8192;
      const bool k1r = xhalf && o.k1_rows != 0 && n % 4 == 0 && n / 4 <= 256;   // (lpc_module.cpp: kK1Rows)
      if (o.prow_nt128 == 0 || (o.prow_nt128 < 0 && (!batch || k1r))) nt = 256;
    }
    set_static_fft(sp.row, n, rad, 1, nt, (n + nt - 1) / nt);
    if (sp.row.n && sp.row.em <= 16) {
      sp.row_kind = LPC_ROWS_PAIRED;
      sp.row_sk = row_layout(n, rad);
      sp.row_x = xhalf;
    }
  }
  // ---- pass A of a split column transform: 32 columns per tile (256-byte row segments at its long row stride) while
  // the fused middle keeps 16 -- the two passes tile the columns independently.  Same-box A/B at 12 MP with T = 32 for
  // both (profiles/r02_notes.md): pass A 0.578 / 0.575 -> 0.530 / 0.510 ms, the middle 0.655 -> 0.71 ms.
  if (st_cols && e->N1 > 1) {
    int T = e->T;
    if (e->T == 16 && g.Wc >= 256 && o.col_t == 0) T = 32;
    if (o.passa_t > 0) T = o.passa_t;
    while (T > 1 && (long)e->N1 * T > kMaxTilePoints) T /= 2;
    plan_radices(e->N1, rad);
    if (e->N1 == 90) rad = {10, 9};      // two stages instead of 6.5.3: 16 x 1080p planes 48.1 -> 46.9 ms per 20 iterations
                                          // (r03k_ab.log; 9.10, 18.5, 30.3 are slower, and 128 = 16.8 is slower than 8.8.2 at 12 MP)
    const int pts = e->N1 * T;
    int nt = T >= 32 ? 512 : 256;
    while (nt < 1024 && (pts + nt - 1) / nt > 16) nt *= 2;
    set_static_fft(sp.passA, e->N1, rad, T, nt, (pts + nt - 1) / nt);
    if (sp.passA.em > 16) sp.passA = StaticFft{};
  }
  // ---- ADMM's fused middle in LDS (a 24-point pass B lives in registers: k_cols_mid_admm_reg, core library)
  const bool reg_mid = e->N1 > 1 && e->mid_reg && f32 && e->N2 == 24;
  if (admm && st_cols && !reg_mid) {
    const int n = e->N2, T = e->T;
    plan_radices(n, rad);
    // 540 = 30.18 side by side (two fat register butterflies, one LDS trip; 184 registers, one workgroup per CU);
    // one spectrum at a time: 6.10.9 inside a 128-register budget = TWO workgroups per CU overlapping one another's
    // loads and barriers -- 0.650 ms per launch at 64 frames against 0.84 ms for 30.18 and 0.95 ms for 6.6.5.3
    // (on 512 lanes the order 10.6.9 is 2 % faster than 6.10.9 -- 0.453 vs 0.464 ms at 64 frames, two instances each,
    // profiles/r04u_ab_shard5.log; 9.10.6 0.479, 10.9.6 0.482, 6.9.10 0.498)
    if (n == 540) rad = seq ? std::vector<int>{10, 6, 9} : std::vector<int>{30, 18};
    // A launch of fewer workgroups than the chip holds at once (one DiffuserCam frame: 183 tiles on 256 CUs) lasts as
    // long as ONE workgroup takes: twice the lanes on half the points each shorten that chain -- 540 x 16 points on 1024
    // lanes as 6.10.9 (every stage has >= 864 butterflies; 30.18 has 288 / 480): C1's middle 21.2 -> 19.1 us, the
    // 5-iteration call 0.243 -> 0.234 ms (profiles/r04t_ab_c1.log; 30.18 on 1024 lanes 19.8 us, 768 lanes 19.6 us).
    // Only while every workgroup has a CU of its own (256 on an MI355X): two frames = 366 tiles are 6 % SLOWER that way
    // (0.370 -> 0.392 ms, r04t_ab_c1c.log).
    const bool one_wave_of_tiles = !seq && e->N1 == 1 && (long)e->P * ((g.Wc + T - 1) / T) <= plan_cu_count() && n * 2 * T > 8192;
    if (one_wave_of_tiles && n == 540) rad = {6, 10, 9};
    const int pts = n * (seq ? T : 2 * T);
    int nt = pts <= 4096 ? 256 : (pts <= 9216 ? 512 : 1024);
    // one spectrum at a time: 8 columns x 540 points on 512 lanes x 9 points (round 3: 256 x 17) -- the middle of a batch
    // of 8 / 16 / 32 / 64 frames 79.6 -> 75.7 / 140 -> 136 / 262 -> 255 / 505 -> 491 us, the 8-frame shard's 20-iteration
    // call 4.55 -> 4.44 ms (profiles/r04u_ab_shard2.log; 384 lanes 103 us, 1024 lanes 77.9 us, 16 columns x 1024: 84 us);
    // three instances of each at 64 / 8 frames (r04u_ab_shard4.log): 256 lanes 0.499 ms / 78.5 us, 512 lanes 0.493 / 74.2,
    // 512 lanes inside 64 VGPRs 0.464 / 70.6
    if (seq) nt = pts <= 9 * 256 ? 256 : (pts <= 18 * 512 ? 512 : 1024);
    if (one_wave_of_tiles) nt = 1024;
    set_static_fft(sp.mid, n, rad, T, nt, (pts + nt - 1) / nt);
    if (sp.mid.n && sp.mid.em <= 18) {
      sp.mid_kind = seq ? LPC_MID_SEQ : LPC_MID_PAIR;
      if (seq) {   // waves per SIMD the register allocation must allow: as many workgroups as the LDS holds
        // both tiles' loads up front (round 4: large batches only; round 6: the 8-frame shard too, 3.86 -> 3.83 ms per call)
        sp.mid_pre = o.mid_pre >= 0 ? (o.mid_pre ? 1 : 0) : 1;
        const size_t lds = (size_t)n * (T + 1) * sizeof(real2);        // tile + the plan's twiddles behind it
        const int wgs = (int)std::max<size_t>(1, std::min<size_t>(8, (160 * 1024) / lds));
        // (512 lanes: 8 = a 64-VGPR allocation, four workgroups per CU as the LDS allows -- 68 registers without the
        // bound, i.e. three; the middle of 64 / 8 frames 0.493 -> 0.464 ms / 74.2 -> 70.6 us)
        sp.mid_minw = std::min(8, std::max(1, (wgs * nt + 255) / 256));
      }
    } else {
      sp.mid = StaticFft{};
    }
  }
  // pair-line work spectra (lpc_kernels.h: spec_col): paired rows + a single-pass middle of 8-column tiles, float32 (a tile row
  // of 8 complex128 columns is a whole line already)
  sp.slay = (admm && f32 && sp.row_kind == LPC_ROWS_PAIRED && sp.mid_kind != LPC_MID_RUNTIME && e->N1 == 1 && sp.mid.T == 8 &&
             o.spec_lay != 0) ? 1 : 0;
  // ... the sequential middle's point-wise constants precombined (k_mid_consts); even padded sizes: the ifftshift phases are
  // +-1, one complex constant per element instead of two (mid_pc = 2)
  sp.mid_pc = (sp.slay && sp.mid_kind == LPC_MID_SEQ && o.mid_pc != 0) ? ((g.Hp % 2 == 0 && g.Wp % 2 == 0) ? 2 : 1) : 0;
}

// frame geometry (rfft_convolve.py:110-117) and the launch plan -- no device work (also serves lpc_plan_module)
static int setup_shape(Engine* e, bool* want_static_out) {
  const lpc_config& c = e->cfg;
  PlaneGeom& g = e->g;
  g.H = c.height; g.W = c.width;
  g.Hp = next_5smooth(2 * g.H - 1);
  g.Wp = next_5smooth(2 * g.W - 1);
  g.Wc = g.Wp / 2 + 1;
  g.sh = (g.Hp - g.H) / 2;
  g.sw = (g.Wp - g.W) / 2;
  g.rpitch = (g.Wp + 3) / 4 * 4;
  g.cpitch = (g.Wc + 15) / 16 * 16;
  g.rplane = (long)g.Hp * g.rpitch;
  g.cplane = (long)((g.Hp + 1) & ~1) * g.cpitch;      // whole row pairs (PlaneGeom::slay)
  g.slay = 0;
  g.uplane = (long)g.H * g.W;
  g.DC = c.depth * c.channels;
  g.C = c.channels;
  g.rev = 0;
  e->Ppsf = g.DC;
  e->P = c.batch * g.DC;
  e->Pdata = c.batch * c.channels;
  if (g.Wp > kMaxTilePoints)
    return fail("padded width " + std::to_string(g.Wp) + " > " + std::to_string(kMaxTilePoints) + " is not supported");
  // compile-time plans live in a plan module (lpc_plan.h): look for it, build it if allowed, else run-time plans
  const bool want_static = !e->opt.no_static && (long)g.Hp * g.Wp >= e->opt.jit_min_points;
  choose_plan(e, want_static);
  *want_static_out = want_static;
  return 0;
}

static int setup_geometry(Engine* e) {
  bool want_static = false;
  LPC_OK(setup_shape(e, &want_static));
  const lpc_config& c = e->cfg;
  const PlaneGeom& g = e->g;
  e->mod = nullptr;
  if (!want_static) e->mod_note = e->opt.no_static ? "no_static" : "small frame";
  else if (e->spec.any()) {
    e->mod = get_plan_module(e->spec, e->opt, e->opt.jit != 0, &e->mod_note);
    if (!e->mod) {      // lpc_plan_info() names the module a deployment without a compiler would have to ship
      const std::string key = plan_spec_key(e->spec);
      if (e->mod_note.find(key) == std::string::npos) e->mod_note = "module " + key + ": " + e->mod_note;
      choose_plan(e, false);
    }
  }
  if (!e->mod) e->spec = PlanSpec{};
  e->g.slay = (e->mod && e->mod->slay) ? 1 : 0;
  const bool admm = c.algo == LPC_ALGO_ADMM;
  LPC_OK(build_plan(e, e->planW, g.Wp));
  e->rows_r2 = e->planW.nst >= 2 && e->planW.radix[e->planW.nst - 1] == 2;
  if (e->rows_r2) {
    std::vector<int> rad{2};
    for (int st = 0; st + 1 < e->planW.nst; ++st) rad.push_back(e->planW.radix[st]);
    LPC_OK(plan_from_radices(e, e->planWi, g.Wp, rad));
    e->planWi.skew_ok = 0;
  }
  if (e->rows_half) LPC_OK(build_plan(e, e->planWh, g.Wp / 2));
  e->tws_row = nullptr;
  if (e->mod && e->spec.row_kind != LPC_ROWS_RUNTIME) LPC_OK(make_stage_twiddles(e, e->spec.row, &e->tws_row));
  LPC_OK(build_plan(e, e->planB, e->N2));
  if (e->N1 > 1) LPC_OK(build_plan(e, e->planA, e->N1));
  // The half of the image-domain work that needs no neighbours rides in the module's forward row kernel: the blocks of
  // `a` compute xi' and a = mu1 X - xi' from xi, HV, HV_old, y themselves (-2R per iteration), the tiled kernel keeps
  // the stencil half at its own occupancy (without a module: the full stand-alone kernel) ...
  e->xhalf_rows = admm && e->mod && e->mod->admm_rows_fwd_x;
  // ... narrow frames (paired rows of one quad per lane: padded widths up to 1024) hand it the TV / W half too: three
  // launches per iteration, r_sp never stored.  One small frame is a chain of launch boundaries and memory latencies
  // (C1 -7.6 %), a batch saves the trip of r_sp through memory and the tiled kernel's launch (C4 -6.3 %);
  // profiles/r05_notes.md section 5 (option k1_rows=0: off)
  // (round 6: rows of TWO quads per lane as well -- padded widths up to 2048: the reference's own profile frame 760 x 1014
  // gray 0.458 -> 0.442 ms per 5 iterations, 8 frames of 600 x 800 x 3 22.1 -> 20.8 ms per 20; profiles/r06_notes.md.  FOUR
  // quads per lane -- 12-MP half-length rows -- are slower than the tiled kernel: not built)
  e->k1_rows = e->xhalf_rows && g.Wp % 4 == 0 && e->mod->k1_rows != 0 && e->opt.k1_rows != 0;
  // ... outside the sensor window that half works from HV alone (AdmmScalars::xiw; option xi_full: every pixel alike) ...
  e->xi_window = e->xhalf_rows && !e->opt.xi_full;
  // ... and rows wholly outside it skip the H V row transforms in both directions: the kept rows of SB are rescaled by
  // forward pass A (any plan) or, for single-pass columns, by the module's fused middle (option hv_full: off)
  e->hv_skip = e->xi_window && !e->opt.hv_full && e->mod->admm_rows_inv && (e->N1 > 1 || e->mod->admm_mid);
  e->gd_fuse_fwd = c.algo >= LPC_ALGO_GD && e->mod && e->mod->gd_rows_update_fwd && !e->opt.gd_no_fuse_fwd;
  // the second form of the fused row kernels: 8-byte accesses to y / x need an even window offset and frame width
  e->gd_v2 = c.algo >= LPC_ALGO_GD && e->mod && e->mod->gd_v2 && e->tws_row && e->opt.gd_v2 != 0 &&
             ((g.sw | g.W) & 1) == 0 && g.W >= 2;
  if (e->opt.gd_rev < 0)     // EngineOpts::gd_rev
    // all three (the row kernels and the register middle alternate with the forward-walking pass A, so every kernel
    // starts where its predecessor finished): 12 MP FISTA 75.4 / 74.1 / 73.8 -> 74.5 / 73.0 / 72.8 ms per 40 iterations on
    // three instances of one box against the middle alone (r03z_ab.log); no effect at 1080p, where nothing is reversed
    e->opt.gd_rev = ((size_t)g.cplane * e->P * sizeof(real2) > ((size_t)200 << 20)) ? 7 : 0;
  LPC_OK(make_twiddles(e, g.Hp, &e->twH));
  const int ntc = (g.Wc + e->T - 1) / e->T;
  ColPass& A = e->passA;
  A.N = e->N1; A.G = e->N2; A.istride = e->N2; A.gstride = 1; A.T = e->T; A.ntile_c = ntc;
  A.tw_mode = 0; A.zr0 = 0; A.zr1 = g.Hp; A.twH = e->twH; A.need0 = 0; A.needn = g.Hp;
  A.sc_plane0 = INT_MAX; A.sc_r0 = 0; A.sc_r1 = g.Hp; A.sc = (real)1.;
  A.tdiv = make_fastdiv((unsigned)e->T); A.tcdiv = make_fastdiv((unsigned)ntc);
  A.swz = 0;
  A.rev = 0;
  A.ga = A.gb = nullptr;
  ColPass& B = e->passB;
  B = A;
  B.N = e->N2; B.G = e->N1; B.istride = 1; B.gstride = e->N2;
  if (e->spec.passA.n) {       // the module's pass A tiles the columns on its own (choose_plan)
    A.T = e->spec.passA.T;
    A.ntile_c = (g.Wc + A.T - 1) / A.T;
    A.tdiv = make_fastdiv((unsigned)A.T);
    A.tcdiv = make_fastdiv((unsigned)A.ntile_c);
  }
  // ifftshift phases: out[i] = in[(i + n/2) mod n]  <=>  multiply bin k by exp(+2 pi i k (n/2) / n)
  std::vector<real2> pr((size_t)g.Hp), pc((size_t)g.Wc);
  for (int p = 0; p < g.Hp; ++p) {
    const long k = stored_row_freq(e, p);
    const double a = 2.0 * M_PI * (double)((k * (g.Hp / 2)) % g.Hp) / (double)g.Hp;
    pr[p] = make_real2((real)std::cos(a), (real)std::sin(a));
  }
  for (int k = 0; k < g.Wc; ++k) {
    const double a = 2.0 * M_PI * (double)(((long)k * (g.Wp / 2)) % g.Wp) / (double)g.Wp;
    pc[k] = make_real2((real)std::cos(a), (real)std::sin(a));
  }
  LPC_OK(dev_alloc(e, &e->phr, pr.size()));
  LPC_OK(dev_alloc(e, &e->phc, pc.size()));
  LPC_OK(upload(e, e->phr, pr.data(), pr.size() * sizeof(real2)));
  LPC_OK(upload(e, e->phc, pc.data(), pc.size() * sizeof(real2)));
  return 0;
}

// full forward 2-D transform of a real source into S (used for the PSF and the TV gram)
static int fft2_forward_setup(Engine* e, const RealSrc& src, real2* S, int nplanes) {
  const PlaneGeom& g = e->g;
  const int zr0 = src.out_row0, zr1 = src.out_row0 + src.nrows;
  LPC_OK(rows_fwd_single(e, src, S, nplanes, -1));
  if (e->N1 > 1) {
    LPC_OK(cols_passA(e, S, nplanes, false, zr0, zr1, -1));
    LPC_OK(cols_passB_fwd(e, S, nplanes, 0, g.Hp));
  } else {
    LPC_OK(cols_passB_fwd(e, S, nplanes, zr0, zr1));
  }
  return 0;
}

// planar real (padded or not) -> convolution with H / H* -> planar real, same kind
static int convolve_planar(Engine* e, const real* xin, real* xout, int nplanes, bool padded_io, bool adjoint) {
  const PlaneGeom& g = e->g;
  if (padded_io) {
    LPC_OK(rows_fwd_single(e, src_padded(e, xin), e->S, nplanes, LPC_K_ROW_FWD));
    LPC_OK(conv_middle(e, e->S, nplanes, adjoint, 0, g.Hp));
    LPC_OK(rows_inv_single(e, e->S, dst_padded(e, xout), nplanes, LPC_K_ROW_INV));
  } else {
    LPC_OK(rows_fwd_single(e, src_unpadded(e, xin), e->S, nplanes, LPC_K_ROW_FWD));
    LPC_OK(conv_middle(e, e->S, nplanes, adjoint, g.sh, g.sh + g.H, true));
    LPC_OK(rows_inv_single(e, e->S, dst_cropped(e, xout), nplanes, LPC_K_ROW_INV));
  }
  return 0;
}

// ------------------------------------------------------------------ layout helpers --
static int hwc_to_planar(Engine* e, const real* src, real* dst, int nimg, int rows, int cols, int pitch,
                         long dplane, int src_channels = 0) {
  const long n = (long)rows * cols * e->cfg.channels;
  return launch_k(e, -1, k_hwc_to_planar<256>, grid1d(n, 256, nimg), 256, 0, src, dst, rows, cols,
                  e->cfg.channels, pitch, dplane, src_channels > 0 ? src_channels : e->cfg.channels);
}
// channel count of a caller's buffer: the handle's own, or 1 (broadcast) -- anything else would make the kernels
// read past the end of the buffer
static int check_channels(const Engine* e, int ch, const char* who) {
  if (ch == e->cfg.channels || ch == 1) return 0;
  return fail(std::string(who) + ": buffer has " + std::to_string(ch) + " channel(s), the PSF " +
              std::to_string(e->cfg.channels) + " (only 1 -> C broadcasts)");
}
static int planar_to_hwc(Engine* e, real* src, real* dst, int nimg, int rows, int cols, int pitch, long splane,
                         int row0, int col0, int clamp) {
  const long n = (long)rows * cols * e->cfg.channels;
  // clamp: 0 none, 1 everywhere, 2 inside the sensor window only (rows / cols then span the padded frame)
  const PlaneGeom& g = e->g;
  return launch_k(e, -1, k_planar_to_hwc<256>, grid1d(n, 256, nimg), 256, 0, src, dst, rows, cols,
                  e->cfg.channels, pitch, splane, row0, col0, clamp, 0, clamp == 2 ? g.sh : 0,
                  clamp == 2 ? g.sh + g.H : rows, clamp == 2 ? g.sw : 0, clamp == 2 ? g.sw + g.W : cols);
}

// ------------------------------------------------------------------------------ ADMM --
// parameters of iteration `it` (since reset): the schedule if one is set, else the constructor's
static void admm_params(const Engine* e, long it, double out[4]) {
  const lpc_config& c = e->cfg;
  const double dflt[4] = {c.mu1, c.mu2, c.mu3, c.tau};
  for (int k = 0; k < 4; ++k) {
    const std::vector<double>& v = e->sched[k];
    out[k] = v.empty() ? dflt[k] : v[(size_t)std::min<long>(it, (long)v.size() - 1)];
  }
}

static AdmmScalars admm_scalars(const Engine* e, const double cur[4]) {
  AdmmScalars p;
  p.mu1 = (real)cur[0]; p.mu2 = (real)cur[1]; p.mu3 = (real)cur[2];
  p.thr = (real)(cur[3] / cur[1]);               // admm.py:246: python-double division, then float32
  p.m_in = (real)1.0 / ((real)1.0 + p.mu1);                 // admm.py:193 in float32
  p.m_out = (real)1.0 / ((real)0.0 + p.mu1);
  p.first = e->first ? 1 : 0;
  const double* prev = e->first ? cur : e->last_par;
  p.mu1p = (real)prev[0]; p.mu2p = (real)prev[1]; p.mu3p = (real)prev[2];
  p.thrp = (real)(prev[3] / prev[1]);
  p.m_in_p = (real)1.0 / ((real)1.0 + p.mu1p);
  p.m_out_p = (real)1.0 / ((real)0.0 + p.mu1p);
  p.r_mu2 = (real)(1.0 / (double)p.mu2); p.r_mu3 = (real)(1.0 / (double)p.mu3);      // RN(1/d): see div_by
  p.r_mu2p = (real)(1.0 / (double)p.mu2p); p.r_mu3p = (real)(1.0 / (double)p.mu3p);
  p.clamp_cur = e->vw_cur ? 1 : 0;
  p.clamp_old = e->vw_old ? 1 : 0;
  p.xiw = e->xi_window ? 1 : 0;
  p.xi_store = 1;              // admm_iterate clears it on all but the last iteration of a call
  p.skipa = p.skiphv = 0;      // set by admm_iterate inside a call (AdmmScalars::skipa)
  p.rev = (e->opt.rev_order & 1) ? 1 : 0;
  p.half_in = p.half_out = 0;  // set by admm_iterate between the iterations of one call (AdmmScalars::half_in)
  return p;
}

static const int kGsepBlocks = 512;
static int admm_alloc(Engine* e) {
  const PlaneGeom& g = e->g;
  const size_t rp = (size_t)g.rplane * e->P;
  // the eight arrays reset() zeroes are ONE block, V[0] first: one fill instead of eight (a reset of a DiffuserCam-sized
  // frame was 8 x 5.3 us of launch-bound fills in a 264-us apply(), profiles/r04k_c1_gaps.txt)
  real* zeroed = nullptr;
  LPC_OK(dev_alloc(e, &zeroed, 8 * rp));
  real** zb[] = {&e->V[0], &e->V[1], &e->HVb[0], &e->HVb[1], &e->xi, &e->eta0[0], &e->eta1[0], &e->rho};
  for (int k = 0; k < 8; ++k) *zb[k] = zeroed + (size_t)k * rp;
  real** bufs[] = {&e->eta0[1], &e->eta1[1], &e->Rsp, &e->Aarr};
  for (real** b : bufs) LPC_OK(dev_alloc(e, b, rp));
  LPC_OK(dev_alloc(e, &e->Gabs, (size_t)g.cplane));
  if (g.slay) {
    LPC_OK(dev_alloc(e, &e->Gabs_t, (size_t)g.cplane));
    LPC_OK(dev_alloc(e, &e->Hs_t, (size_t)g.cplane * e->Ppsf));
    if (e->mod && e->mod->mid_pc) {
      real2* c = nullptr;
      LPC_OK(dev_alloc(e, &c, (size_t)g.cplane * e->Ppsf * (e->mod->mid_pc == 1 ? 2 : 1)));
      e->midc = c;
      LPC_OK(dev_alloc(e, &e->midrd, (size_t)g.cplane * e->Ppsf));
    }
  }
  LPC_OK(dev_alloc(e, &e->Ga, (size_t)g.Hp));
  LPC_OK(dev_alloc(e, &e->Gb, (size_t)g.cpitch));
  LPC_OK(dev_alloc(e, &e->Gpart, (size_t)2 * kGsepBlocks));
  return 0;
}

// |PsiT Psi| as row term + column term (ColPass::ga): taken when the plane in e->Gabs separates to float32 round-off
// (the reference's finite-difference gram does, admm.py:385-397; a caller's psi_gram in general does not)
static int admm_split_gram(Engine* e) {
  const PlaneGeom& g = e->g;
  e->g_sep = 0;
  e->midc_valid = false;      // (a new PSF or gram: the middle's precombined constants are remade at the next step)
  if (g.slay)       // the 8-column middle reads the plane in pair lines
    LPC_OK(launch_k(e, -1, k_to_pair_lines<256, real>, grid1d((long)g.Hp * g.cpitch, 256), 256, 0, (const real*)e->Gabs,
                    e->Gabs_t, g.Hp, g.cpitch, g.cplane));
  // Measured (r03z_ab.log): at 12 MP (100-MB plane, 64-byte tile rows fetched as whole lines once per colour plane) the
  // terms take 0.5 GB off the middle's HBM traffic, 0.622 -> 0.563 ms; on DiffuserCam-sized frames the 1-MB plane lives
  // in the L2 and one load beats two (C1 middle 0.0206 -> 0.0221 ms with the terms)
  const int want = e->opt.g_plane >= 0 ? !e->opt.g_plane : ((size_t)g.cplane * sizeof(real) > ((size_t)8 << 20));
  if (!want) return 0;
  const int n = (int)std::max<long>(g.Hp, g.cpitch);
  LPC_OK(launch_k(e, -1, k_gsep_extract, grid1d((long)n, 256), 256, 0, (const real*)e->Gabs, g.Hp, g.Wc,
                  (long)g.cpitch, e->Ga, e->Gb));
  LPC_OK(launch_k(e, -1, k_gsep_check<256>, dim3(kGsepBlocks), 256, 2 * 256 * sizeof(real), (const real*)e->Gabs, g.Hp,
                  g.Wc, (long)g.cpitch, (const real*)e->Ga, (const real*)e->Gb, e->Gpart));
  std::vector<real> part((size_t)2 * kGsepBlocks);
  LPC_RT(rt::copy_d2h_async(part.data(), e->Gpart, part.size() * sizeof(real), e->stream));
  LPC_RT(rt::stream_sync(e->stream));
  double err = 0., top = 0.;
  for (int b = 0; b < kGsepBlocks; ++b) { err = std::max(err, (double)part[2 * b]); top = std::max(top, -(double)part[2 * b + 1]); }
  const double eps = sizeof(real) == 4 ? 1e-6 : 1e-13;        // a few ulp of the largest entry: FFT round-off of the gram
  e->g_sep = (top > 0. && err <= eps * top) ? 1 : 0;
  return 0;
}

static int admm_setup_constants(Engine* e) {
  // R_divmat = 1/(mu1 |H* H| + mu2 |PsiT Psi| + mu3)  (admm.py:186-190) is formed inside the middle
  // kernel; here only |PsiT Psi| is prepared.  The gram spectrum is produced by the engine's own forward
  // transform of the 5-point stencil (admm.py:385-397) so that it lands in the permuted row order.
  const PlaneGeom& g = e->g;
  real* stencil = e->Rsp;  // scratch: one padded plane
  LPC_RT(rt::memset_async(stencil, 0, (size_t)g.rplane * sizeof(real), e->stream));
  std::vector<real> host((size_t)g.rplane, (real)0.);
  // gram[0,0]=4; [0,1]=[0,-1]=[1,0]=[-1,0]=-1 with python negative indexing (later writes win)
  host[0] = (real)4.;
  host[(size_t)(1 % g.Wp)] = -(real)1.;
  host[(size_t)(g.Wp - 1)] = -(real)1.;
  host[(size_t)(1 % g.Hp) * g.rpitch] = -(real)1.;
  host[(size_t)(g.Hp - 1) * g.rpitch] = -(real)1.;
  LPC_OK(upload(e, stencil, host.data(), host.size() * sizeof(real)));
  real2* Gs = e->S;  // scratch spectrum plane
  LPC_OK(fft2_forward_setup(e, src_padded(e, stencil), Gs, 1));
  LPC_OK(launch_k(e, -1, k_abs_complex<256>, grid1d((long)g.cplane, 256), 256, 0, (const real2*)Gs, e->Gabs,
                  (long)g.cplane));
  return admm_split_gram(e);
}

static int admm_reset(Engine* e) {
  const PlaneGeom& g = e->g;
  const size_t rb = (size_t)g.rplane * e->P * sizeof(real);
  // V[1], HVb[0], HVb[1], xi, eta0[0], eta1[0], rho: contiguous behind V[0] (admm_alloc)
  if (e->has_init) LPC_RT(rt::memset_async(e->V[1], 0, 7 * rb, e->stream));
  else LPC_RT(rt::memset_async(e->V[0], 0, 8 * rb, e->stream));
  e->vcur = 0;
  e->ecur = 0;
  e->hcur = 0;
  e->vw_cur = e->vw_old = false;
  if (e->has_init) {
    LPC_RT(rt::copy_d2d_async(e->V[0], e->init_est, rb, e->stream));
    // admm.py:172-176: forward_out = convolve(V0)
    LPC_OK(convolve_planar(e, e->V[0], e->HVb[0], e->P, true, false));
  }
  e->first = true;
  e->pnp_mode = e->pnp_pending = false;
  e->iters_done = 0;
  return 0;
}

// (r_sp, a) in e->Rsp / e->Aarr  ->  Vout = irfft2(R_div (rfft2 r_sp + s H* rfft2 a)),  HVout = H Vout:
// forward rows, [pass A], fused middle, [inverse pass A], inverse rows
static int admm_spectral_step(Engine* e, const AdmmScalars& sc, real* Vout, real* HVout, bool xhalf = false,
                              const K1Rows* k1 = nullptr) {
  if (xhalf) LPC_OK(admm_rows_fwd_x(e, sc, k1));
  else LPC_OK(admm_rows_fwd(e));
  LPC_OK(admm_cols(e, sc));
  return admm_rows_inv(e, Vout, HVout, sc.skiphv != 0);
}


static int admm_iterate(Engine* e, int n_iter) {
  const PlaneGeom& g = e->g;
  constexpr int TH = 16, TW = 64, NT = 256;
  const size_t k1_smem = (size_t)(2 * (TH + 2) * (TW + 2) + (TH + 1) * TW + TH * (TW + 1)) * sizeof(real);
  const unsigned tiles_x = (g.Wp + TW - 1) / TW, tiles_y = (g.Hp + TH - 1) / TH;
  const dim3 k1_grid(tiles_x * tiles_y, e->P, 1);
  // 16-byte-lane kernel whenever the padded width allows aligned four-pixel lanes (every BASELINE size does)
  const bool vec4 = g.Wp % 4 == 0;
  constexpr int TH4 = 8, TW4 = 256;
  const unsigned tiles_x4 = (g.Wp + TW4 - 1) / TW4, tiles_y4 = (g.Hp + TH4 - 1) / TH4;
  const dim3 k1_grid4(tiles_x4 * tiles_y4, e->P, 1);
  const size_t k1_smem4 = (size_t)2 * (TH4 + 2) * (TW4 + 8) * sizeof(real);
  // the TV / W half alone (X half inside the forward rows) is lighter per pixel: 4-row tiles, one row per wave -- three
  // alternations on one box (r02as): 0.940 -> 0.901 ms at 12 MP (6.03 TB/s), C4 0.641 -> 0.614 ms, C5 unchanged
  constexpr int TH4X = 4;
  const dim3 k1_grid4x(tiles_x4 * ((g.Hp + TH4X - 1) / TH4X), e->P, 1);
  const size_t k1_smem4x = (size_t)2 * (TH4X + 2) * (TW4 + 8) * sizeof(real);
  bool sb_rows_valid = false;   // AdmmScalars::skipa may rely on the rows of SB only after a step of this very call
  // K1Rows::xcd_order.  (The XCD-aware block orders assume the MI355X's 8 XCDs x 32 CUs and its dispatch rule "workgroup w
  // on XCD w % 8"; any other part gets launch order: the orders are permutations, results are the same.)
  const int k1_xcd_order = plan_cu_count() != 256 ? 0
                           : (long)paired_rows_grid(g, false) * e->P <= 8192 ? -1 : std::max(0, e->opt.k1_group);
  for (int it = 0; it < n_iter; ++it) {
    real* Vc = e->V[e->vcur];
    real* Vo = e->V[e->vcur ^ 1];
    double par[4];
    admm_params(e, e->iters_done, par);
    AdmmScalars sc = admm_scalars(e, par);
    sc.xi_store = (it + 1 == n_iter || !sc.xiw) ? 1 : 0;
    // rows wholly outside the sensor window: once an iteration of THIS call has run, SB still holds their row spectra
    // (sb_rows_valid: set below, local to the call -- no other entry point can have touched the work spectrum in
    // between); the last iteration runs complete (it stores xi out there), and the last three write H V there:
    // xi = mu1p (HV - HV_old) of the final X half and every read-out after the call need HV_{n-2}, HV_{n-1}, HV_n whole
    sc.skipa = (e->hv_skip && sb_rows_valid && !sc.xi_store) ? 1 : 0;
    sc.skiphv = (e->hv_skip && it + 3 < n_iter) ? 1 : 0;
    // duals half-applied between the iterations of this call (AdmmScalars::half_in; option k1_half=0: never): the
    // first iteration reads plain duals, the last one writes them -- nothing outside this loop sees the other form
    const bool k1_half = vec4 && e->xhalf_rows && e->opt.k1_half != 0;
    sc.half_in = (k1_half && it > 0) ? 1 : 0;
    sc.half_out = (k1_half && it + 1 < n_iter) ? 1 : 0;
    // small frames: the forward rows take the TV / W half as well (Engine::k1_rows) -- same buffers, same ping-pong
    const bool k1r = e->k1_rows && vec4;
    const K1Rows k1 = {Vc, Vo, e->eta0[e->ecur], e->eta1[e->ecur], e->eta0[e->ecur ^ 1], e->eta1[e->ecur ^ 1], e->rho,
                       k1_xcd_order};
    if (k1r) {
    } else if (sc.half_in)
      LPC_OK(launch_k(e, LPC_K_SPATIAL, k_admm_spatial_v4<TH4X, NT, false, true>, k1_grid4x, NT, k1_smem4x / 2, g, sc, (const real*)Vc,
                      (const real*)Vo, (const real*)e->HVb[e->hcur], (const real*)e->HVb[e->hcur ^ 1], e->xi, (const real*)e->eta0[e->ecur],
                      (const real*)e->eta1[e->ecur], e->eta0[e->ecur ^ 1], e->eta1[e->ecur ^ 1], e->rho,
                      (const real*)e->Y, e->Rsp, e->Aarr, tiles_x4));
    else if (vec4 && e->xhalf_rows)
      LPC_OK(launch_k(e, LPC_K_SPATIAL, k_admm_spatial_v4<TH4X, NT, false>, k1_grid4x, NT, k1_smem4x, g, sc, (const real*)Vc,
                      (const real*)Vo, (const real*)e->HVb[e->hcur], (const real*)e->HVb[e->hcur ^ 1], e->xi, (const real*)e->eta0[e->ecur],
                      (const real*)e->eta1[e->ecur], e->eta0[e->ecur ^ 1], e->eta1[e->ecur ^ 1], e->rho,
                      (const real*)e->Y, e->Rsp, e->Aarr, tiles_x4));
    else if (vec4)
      LPC_OK(launch_k(e, LPC_K_SPATIAL, k_admm_spatial_v4<TH4, NT>, k1_grid4, NT, k1_smem4, g, sc, (const real*)Vc,
                      (const real*)Vo, (const real*)e->HVb[e->hcur], (const real*)e->HVb[e->hcur ^ 1], e->xi, (const real*)e->eta0[e->ecur],
                      (const real*)e->eta1[e->ecur], e->eta0[e->ecur ^ 1], e->eta1[e->ecur ^ 1], e->rho,
                      (const real*)e->Y, e->Rsp, e->Aarr, tiles_x4));
    else
    LPC_OK(launch_k(e, LPC_K_SPATIAL, k_admm_spatial<TH, TW, NT>, k1_grid, NT, k1_smem, g, sc, (const real*)Vc,
                    (const real*)Vo, (const real*)e->HVb[e->hcur], (const real*)e->HVb[e->hcur ^ 1], e->xi, (const real*)e->eta0[e->ecur],
                    (const real*)e->eta1[e->ecur], e->eta0[e->ecur ^ 1], e->eta1[e->ecur ^ 1], e->rho,
                    (const real*)e->Y, e->Rsp, e->Aarr, tiles_x));
    e->vw_old = e->vw_cur;          // this iteration's "V as W saw it" becomes the next one's "V_old as W_old saw it"
    e->vw_cur = false;
    e->ecur ^= 1;
    e->first = false;
    // (hcur still names the CURRENT H V here: the X half inside the forward rows reads HVb[hcur] and HVb[hcur ^ 1]
    // before the inverse rows of this same step overwrite HVb[hcur ^ 1] -- stream order)
    LPC_OK(admm_spectral_step(e, sc, Vo, e->HVb[e->hcur ^ 1], vec4 && e->xhalf_rows, k1r ? &k1 : nullptr));
    e->vcur ^= 1;  // Vo now holds the new image estimate
    e->hcur ^= 1;  // ... and the other H V buffer its forward model
    sb_rows_valid = true;   // the inverse column passes of this step left rfft(H V row) / Wp in every row of SB
    for (int k = 0; k < 4; ++k) e->last_par[k] = par[k];
    ++e->iters_done;
  }
  return 0;
}

#include "lpc_gd_engine.inc"

// process-wide defaults from the environment first (the ONE launch-plan variable the library reads), then the handle's own
static std::string parse_all_opts(const char* handle_opts, EngineOpts& o) {
  std::string err = parse_engine_opts(std::getenv("LPC_OPTIONS"), o);
  return err.empty() ? parse_engine_opts(handle_opts, o) : err;
}

// =============================================================================== C ABI ==
extern "C" {

const char* lpc_last_error(void) { return g_last_error.c_str(); }
const char* lpc_backend(void) { return rt::backend_name(); }
const char* lpc_real_name(void) { return LPC_REAL_NAME; }

int lpc_create(const lpc_config* cfg, lpc_handle* out) {
  if (!cfg || !out) return fail("lpc_create: null argument");
  *out = nullptr;
  if (cfg->height < 1 || cfg->width < 1) return fail("lpc_create: bad spatial size");
  if (cfg->channels != 1 && cfg->channels != 3) return fail("PSF must either be rgb (3) or grayscale (1)");
  if (cfg->depth < 1 || cfg->batch < 1) return fail("lpc_create: depth and batch must be >= 1");
  if (cfg->algo < LPC_ALGO_CONV || cfg->algo > LPC_ALGO_FISTA) return fail("lpc_create: unknown algo");
  if (cfg->norm < 0 || cfg->norm > 2) return fail("lpc_create: unknown norm");
  int ndev = 0;
  if (rt::device_count(&ndev) != lpcSuccess || ndev < 1)
    return fail("no HIP device: the engine has no CPU path");
  Engine* e = new Engine();
  e->cfg = *cfg;
  e->cfg.options = nullptr;            // (the caller's string is not kept)
  {   // process-wide defaults from the environment first, then the handle's own
    std::string err = parse_all_opts(cfg->options, e->opt);
    if (!err.empty()) { delete e; return fail("lpc_create: " + err); }
  }
  e->tk = cfg->fista_tk; e->nest_mu = cfg->nesterov_mu; e->nest_p = cfg->nesterov_p;
  int rc = setup_geometry(e);
  const PlaneGeom& g = e->g;
  if (!rc) rc = dev_alloc(e, &e->Hs, (size_t)g.cplane * e->Ppsf);
  if (!rc) rc = dev_alloc(e, &e->psf_planar, (size_t)g.uplane * e->Ppsf);
  const int nspec = cfg->algo == LPC_ALGO_ADMM ? 2 : 1;
  if (!rc) rc = dev_alloc(e, &e->S, (size_t)g.cplane * e->P * nspec);
  if (!rc && cfg->algo != LPC_ALGO_CONV) rc = dev_alloc(e, &e->Y, (size_t)g.uplane * e->Pdata);
  if (!rc && cfg->algo == LPC_ALGO_CONV) {  // staging planes for the channels-last <-> planar hop
    const size_t n = (size_t)(cfg->pad ? g.uplane : g.rplane) * e->P;
    rc = dev_alloc(e, &e->gaux, n);
    if (!rc) rc = dev_alloc(e, &e->gx, n);
  }
  if (!rc && cfg->algo == LPC_ALGO_ADMM) rc = admm_alloc(e);
  if (!rc && cfg->algo >= LPC_ALGO_GD) rc = gd_alloc(e);
  if (rc) {
    lpc_destroy(e);
    return rc;
  }
  *out = e;
  return 0;
}

// the plan module lpc_create(cfg) would use: its key, and (build != 0) compile it now if it is not on disk.  No device
// needed: build.py pre-builds the modules of BASELINE.json's shapes with it in the GPU-less build container.
int lpc_plan_module(const lpc_config* cfg, int build, char* key_buf, size_t n) {
  if (!cfg) return fail("lpc_plan_module: null config");
  if (cfg->height < 1 || cfg->width < 1 || cfg->depth < 1 || cfg->batch < 1) return fail("lpc_plan_module: bad size");
  Engine tmp;
  tmp.cfg = *cfg;
  std::string err = parse_all_opts(cfg->options, tmp.opt);
  if (!err.empty()) return fail("lpc_plan_module: " + err);
  bool want_static = false;
  LPC_OK(setup_shape(&tmp, &want_static));
  const bool any = want_static && tmp.spec.any();
  if (key_buf && n) std::snprintf(key_buf, n, "%s", any ? plan_spec_key(tmp.spec).c_str() : "");
  if (!any || !build) return 0;
  std::string path;
  if (build_plan_module(tmp.spec, tmp.opt, &path) != 0) return fail(path);
  return 0;
}

int lpc_destroy(lpc_handle e) {
  if (!e) return 0;
  (void)rt::stream_sync(e->stream);
  for (void* p : e->allocs) (void)rt::dev_free(p);
  release_plan_module(e->mod);
  e->mod = nullptr;
#if !defined(LPC_SIMT_EMU)
  for (auto& v : e->timer.ev)
    for (auto& pr : v) { (void)hipEventDestroy(pr.first); (void)hipEventDestroy(pr.second); }
#endif
  delete e;
  return 0;
}

int lpc_padded_shape(lpc_handle e, int* Hp, int* Wp, int* sh, int* sw) {
  if (!e) return fail("null handle");
  if (Hp) *Hp = e->g.Hp;
  if (Wp) *Wp = e->g.Wp;
  if (sh) *sh = e->g.sh;
  if (sw) *sw = e->g.sw;
  return 0;
}

int lpc_workspace_bytes(lpc_handle e, size_t* bytes) {
  if (!e || !bytes) return fail("null argument");
  *bytes = e->total_bytes;
  return 0;
}

int lpc_set_psf(lpc_handle e, const real* dev_psf, void* stream) {
  if (!e || !dev_psf) return fail("lpc_set_psf: null argument");
  e->stream = (lpcStream_t)stream;
  const PlaneGeom& g = e->g;
  // (D,H,W,C) -> planar [D*C][H][W]
  LPC_OK(hwc_to_planar(e, dev_psf, e->psf_planar, e->cfg.depth, g.H, g.W, g.W, g.uplane));
  LPC_OK(fft2_forward_setup(e, src_unpadded(e, e->psf_planar), e->Hs, e->Ppsf));
  double sc = 1.0;  // rfft_convolve.py:121 norm= of the PSF spectrum
  if (e->cfg.norm == LPC_NORM_ORTHO) sc = 1.0 / std::sqrt((double)g.Hp * (double)g.Wp);
  if (e->cfg.norm == LPC_NORM_FORWARD) sc = 1.0 / ((double)g.Hp * (double)g.Wp);
  if (sc != 1.0) {
    const long n = (long)g.cplane * e->Ppsf;
    LPC_OK(launch_k(e, -1, k_scale_complex<256>, grid1d(n, 256), 256, 0, e->Hs, n, (real)sc));
  }
  e->psf_set = true;
  if (g.slay)
    LPC_OK(launch_k(e, -1, k_to_pair_lines<256, real2>, grid1d((long)g.Hp * g.cpitch, 256, e->Ppsf), 256, 0,
                    (const real2*)e->Hs, e->Hs_t, g.Hp, g.cpitch, g.cplane));
  if (e->cfg.algo == LPC_ALGO_ADMM) LPC_OK(admm_setup_constants(e));
  if (e->cfg.algo >= LPC_ALGO_GD) LPC_OK(gd_setup_constants(e));
  if (e->cfg.algo != LPC_ALGO_CONV) return lpc_reset(e, stream);
  return 0;
}

int lpc_convolve(lpc_handle e, const real* dev_x, real* dev_out, int n, int x_channels, int adjoint, void* stream) {
  if (!e || !dev_x || !dev_out) return fail("lpc_convolve: null argument");
  LPC_OK(check_channels(e, x_channels, "lpc_convolve"));
  if (!e->psf_set) return fail("lpc_convolve: PSF not set");
  if (n < 1 || n > e->cfg.batch) return fail("lpc_convolve: n exceeds the configured batch");
  if (e->cfg.algo != LPC_ALGO_CONV) return fail("lpc_convolve: handle was not created with LPC_ALGO_CONV");
  e->stream = (lpcStream_t)stream;
  const PlaneGeom& g = e->g;
  const int nplanes = n * g.DC;
  const bool padded_io = !e->cfg.pad;
  // staging planes live in the tail of the handle's scratch
  real* xin = (real*)e->gaux;
  real* xout = (real*)e->gx;
  const int nimg = n * e->cfg.depth;
  if (padded_io) {
    LPC_OK(hwc_to_planar(e, dev_x, xin, nimg, g.Hp, g.Wp, g.rpitch, g.rplane, x_channels));
    LPC_OK(convolve_planar(e, xin, xout, nplanes, true, adjoint != 0));
    LPC_OK(planar_to_hwc(e, xout, dev_out, nimg, g.Hp, g.Wp, g.rpitch, g.rplane, 0, 0, 0));
  } else {
    LPC_OK(hwc_to_planar(e, dev_x, xin, nimg, g.H, g.W, g.W, g.uplane, x_channels));
    LPC_OK(convolve_planar(e, xin, xout, nplanes, false, adjoint != 0));
    LPC_OK(planar_to_hwc(e, xout, dev_out, nimg, g.H, g.W, g.W, g.uplane, 0, 0, 0));
  }
  return 0;
}

int lpc_convolve_spectrum(lpc_handle e, const real* dev_x, real* dev_out, int n, int x_channels, int adjoint,
                          void* stream) {
  if (!e || !dev_x || !dev_out) return fail("lpc_convolve_spectrum: null argument");
  LPC_OK(check_channels(e, x_channels, "lpc_convolve_spectrum"));
  if (!e->psf_set) return fail("lpc_convolve_spectrum: PSF not set");
  if (n < 1 || n > e->cfg.batch) return fail("lpc_convolve_spectrum: n exceeds the configured batch");
  if (e->cfg.algo != LPC_ALGO_CONV) return fail("lpc_convolve_spectrum: handle was not created with LPC_ALGO_CONV");
  e->stream = (lpcStream_t)stream;
  const PlaneGeom& g = e->g;
  const int nplanes = n * g.DC, nimg = n * e->cfg.depth;
  real* xin = (real*)e->gaux;
  if (!e->cfg.pad) {
    LPC_OK(hwc_to_planar(e, dev_x, xin, nimg, g.Hp, g.Wp, g.rpitch, g.rplane, x_channels));
    LPC_OK(fft2_forward_setup(e, src_padded(e, xin), e->S, nplanes));
  } else {
    LPC_OK(hwc_to_planar(e, dev_x, xin, nimg, g.H, g.W, g.W, g.uplane, x_channels));
    LPC_OK(fft2_forward_setup(e, src_unpadded(e, xin), e->S, nplanes));
  }
  return launch_k(e, -1, k_spectrum_mul_to_hwc<256>, grid1d((long)g.Hp * g.Wc * g.C, 256, nimg), 256, 0, g,
                  (const real2*)e->S, (const real2*)e->Hs, adjoint ? 1 : 0, (real2*)dev_out, e->N1, e->N2);
}

int lpc_set_data(lpc_handle e, const real* dev_data, int data_channels, void* stream) {
  if (!e || !dev_data) return fail("lpc_set_data: null argument");
  if (e->cfg.algo == LPC_ALGO_CONV) return fail("lpc_set_data: operator-only handle");
  LPC_OK(check_channels(e, data_channels, "lpc_set_data"));
  e->stream = (lpcStream_t)stream;
  const PlaneGeom& g = e->g;
  LPC_OK(hwc_to_planar(e, dev_data, e->Y, e->cfg.batch, g.H, g.W, g.W, g.uplane, data_channels));
  e->data_set = true;
  return 0;
}

int lpc_set_initial_estimate(lpc_handle e, const real* dev_est, void* stream) {
  if (!e) return fail("null handle");
  if (e->cfg.algo == LPC_ALGO_CONV) return fail("lpc_set_initial_estimate: operator-only handle");
  e->stream = (lpcStream_t)stream;
  const PlaneGeom& g = e->g;
  if (!dev_est) { e->has_init = false; return 0; }
  const bool admm = e->cfg.algo == LPC_ALGO_ADMM;
  const size_t n = (size_t)(admm ? g.rplane : g.uplane) * e->P;
  if (!e->init_est) LPC_OK(dev_alloc(e, &e->init_est, n));
  const int nimg = e->cfg.batch * e->cfg.depth;
  if (admm) LPC_OK(hwc_to_planar(e, dev_est, e->init_est, nimg, g.Hp, g.Wp, g.rpitch, g.rplane));
  else LPC_OK(hwc_to_planar(e, dev_est, e->init_est, nimg, g.H, g.W, g.W, g.uplane));
  e->has_init = true;
  return 0;
}

int lpc_reset(lpc_handle e, void* stream) {
  if (!e) return fail("null handle");
  if (!e->psf_set) return fail("lpc_reset: PSF not set");
  e->stream = (lpcStream_t)stream;
  if (e->cfg.algo == LPC_ALGO_ADMM) return admm_reset(e);
  if (e->cfg.algo >= LPC_ALGO_GD) return gd_reset(e);
  return 0;
}

int lpc_set_momentum(lpc_handle e, double p, double mu, double tk) {
  if (!e) return fail("null handle");
  e->nest_p = p; e->nest_mu = mu;
  if (tk > 0) e->tk = tk;
  return gd_apply_momentum_reset(e);
}

int lpc_set_admm_schedule(lpc_handle e, int n, const double* mu1, const double* mu2, const double* mu3,
                          const double* tau) {
  if (!e) return fail("null handle");
  if (e->cfg.algo != LPC_ALGO_ADMM) return fail("lpc_set_admm_schedule: not an ADMM handle");
  for (auto& v : e->sched) v.clear();
  if (n <= 0) return 0;
  if (!mu1 || !mu2 || !mu3 || !tau) return fail("lpc_set_admm_schedule: null array");
  for (int i = 0; i < n; ++i) {
    if (!(mu1[i] > 0) || !(mu2[i] > 0) || !(mu3[i] > 0)) return fail("lpc_set_admm_schedule: step sizes must be > 0");
    e->sched[0].push_back(mu1[i]); e->sched[1].push_back(mu2[i]);
    e->sched[2].push_back(mu3[i]); e->sched[3].push_back(tau[i]);
  }
  return 0;
}

int lpc_set_fista_schedule(lpc_handle e, int n, const real* alpha, const real* coef, void* stream) {
  if (!e) return fail("null handle");
  if (e->cfg.algo != LPC_ALGO_FISTA) return fail("lpc_set_fista_schedule: not a FISTA handle");
  e->stream = (lpcStream_t)stream;
  e->fista_coef.clear();
  e->fista_sched_n = 0;
  if (n <= 0) return 0;
  if (!alpha || !coef) return fail("lpc_set_fista_schedule: null array");
  const int C = e->cfg.channels;
  if (e->galpha_sched && e->galpha_sched_cap < (size_t)n * C) {   // grown: give the old table back
    (void)rt::stream_sync(e->stream);
    e->allocs.erase(std::remove(e->allocs.begin(), e->allocs.end(), (void*)e->galpha_sched), e->allocs.end());
    e->total_bytes -= e->galpha_sched_cap * sizeof(real);
    (void)rt::dev_free(e->galpha_sched);
    e->galpha_sched = nullptr;
  }
  if (!e->galpha_sched) {
    LPC_OK(dev_alloc(e, &e->galpha_sched, (size_t)n * C));
    e->galpha_sched_cap = (size_t)n * C;
  }
  LPC_OK(upload(e, e->galpha_sched, alpha, (size_t)n * C * sizeof(real)));
  e->fista_coef.assign(coef, coef + n);
  e->fista_sched_n = n;
  return 0;
}

int lpc_iterate(lpc_handle e, int n_iter, void* stream) {
  if (!e) return fail("null handle");
  if (n_iter < 0) return fail("lpc_iterate: negative iteration count");
  if (!e->psf_set) return fail("lpc_iterate: PSF not set");
  if (!e->data_set) return fail("Must set data with `set_data()`");
  e->stream = (lpcStream_t)stream;
  if (e->split_pending) return fail("lpc_iterate: a split iteration is in flight (lpc_iterate_end missing)");
  if (e->pnp_mode) return fail("lpc_iterate: the handle runs plug-and-play iterations since the last reset");
  if (e->cfg.algo == LPC_ALGO_ADMM) return admm_iterate(e, n_iter);
  if (e->cfg.algo >= LPC_ALGO_GD) return gd_iterate(e, n_iter);
  return fail("lpc_iterate: operator-only handle");
}

// ---- plug-and-play hook (section 8f row N4): one iteration split at the projection ----
int lpc_iterate_begin(lpc_handle e, void* stream) {
  if (!e) return fail("null handle");
  if (e->cfg.algo < LPC_ALGO_GD) return fail("lpc_iterate_begin: gradient-descent family only");
  if (!e->psf_set) return fail("lpc_iterate_begin: PSF not set");
  if (!e->data_set) return fail("Must set data with `set_data()`");
  if (e->split_pending) return fail("lpc_iterate_begin: the previous split iteration was not finished");
  if (e->fista_sched_n > 0) return fail("lpc_iterate_begin: not available with an unrolled schedule");
  e->stream = (lpcStream_t)stream;
  return gd_iterate(e, 1, 1);
}

int lpc_iterate_end(lpc_handle e, const real* dev_projected, void* stream) {
  if (!e || !dev_projected) return fail("lpc_iterate_end: null argument");
  if (e->cfg.algo < LPC_ALGO_GD) return fail("lpc_iterate_end: gradient-descent family only");
  if (!e->split_pending) return fail("lpc_iterate_end: no split iteration in flight");
  e->stream = (lpcStream_t)stream;
  return gd_finish_split(e, dev_projected);
}

// ---- plug-and-play ADMM (section 8f row N4): one iteration split at the U-update ----
static int pnp_check(lpc_handle e, const char* who) {
  if (!e) return fail("null handle");
  if (e->cfg.algo != LPC_ALGO_ADMM) return fail(std::string(who) + ": ADMM handles only");
  if (!e->psf_set) return fail(std::string(who) + ": PSF not set");
  if (!e->data_set) return fail("Must set data with `set_data()`");
  if (!e->sched[0].empty()) return fail(std::string(who) + ": not available with an unrolled schedule");
  return 0;
}

int lpc_admm_pnp_begin(lpc_handle e, int use_dual, real* dev_denoiser_in, void* stream) {
  LPC_OK(pnp_check(e, "lpc_admm_pnp_begin"));
  if (!dev_denoiser_in) return fail("lpc_admm_pnp_begin: null argument");
  if (e->pnp_pending) return fail("lpc_admm_pnp_begin: the previous split iteration was not finished");
  if (!e->pnp_mode && e->iters_done != 0)
    return fail("lpc_admm_pnp_begin: fused iterations already ran since the last reset");
  e->stream = (lpcStream_t)stream;
  e->pnp_mode = true;
  const PlaneGeom& g = e->g;
  const int nimg = e->cfg.batch * e->cfg.depth;
  real* src = e->V[e->vcur];                    // admm.py:242: denoiser(image_est)
  if (use_dual) {                               // admm.py:237-240: denoiser(U + eta / mu2)
    const long n = (long)g.rplane * e->P;
    LPC_OK(launch_k(e, -1, k_pnp_input<256>, grid1d(n, 256), 256, 0, e->Rsp, (const real*)e->eta1[0],
                    (const real*)e->eta0[0], (real)e->cfg.mu2, n));
    src = e->Rsp;
  }
  LPC_OK(planar_to_hwc(e, src, dev_denoiser_in, nimg, g.Hp, g.Wp, g.rpitch, g.rplane, 0, 0, 0));
  e->pnp_pending = true;
  return 0;
}

int lpc_admm_pnp_end(lpc_handle e, int use_dual, const real* dev_U, void* stream) {
  LPC_OK(pnp_check(e, "lpc_admm_pnp_end"));
  if (!dev_U) return fail("lpc_admm_pnp_end: null argument");
  if (!e->pnp_pending) return fail("lpc_admm_pnp_end: no split iteration in flight");
  e->stream = (lpcStream_t)stream;
  const PlaneGeom& g = e->g;
  const int nimg = e->cfg.batch * e->cfg.depth;
  real *eta = e->eta0[0], *U = e->eta1[0], *X = e->eta0[1], *W = e->eta1[1];
  LPC_OK(hwc_to_planar(e, dev_U, U, nimg, g.Hp, g.Wp, g.rpitch, g.rplane));
  double par[4];
  admm_params(e, e->iters_done, par);
  const AdmmScalars sc = admm_scalars(e, par);
  const dim3 grid = grid1d((long)g.Hp * g.Wp, 256, e->P);
  real* Vc = e->V[e->vcur];
  real* Vn = e->V[e->vcur ^ 1];
  real* HVn = e->HVb[e->hcur ^ 1];
  LPC_OK(launch_k(e, LPC_K_SPATIAL, k_pnp_pre<256>, grid, 256, 0, g, sc, use_dual ? 1 : 0, (const real*)Vc,
                  (const real*)e->HVb[e->hcur], (const real*)e->xi, (const real*)e->rho, (const real*)U,
                  (const real*)eta, (const real*)e->Y, X, W, e->Rsp, e->Aarr));
  LPC_OK(admm_spectral_step(e, sc, Vn, HVn));
  LPC_OK(launch_k(e, LPC_K_SPATIAL, k_pnp_post<256>, grid, 256, 0, g, sc, use_dual ? 1 : 0, (const real*)Vn,
                  (const real*)HVn, (const real*)X, (const real*)W, (const real*)U, e->xi, eta, e->rho));
  e->vcur ^= 1;
  e->hcur ^= 1;
  e->pnp_pending = false;
  e->first = false;
  ++e->iters_done;
  return 0;
}

// ---- ADMM with a caller-supplied sparsifying operator (admm.py:104-120): one iteration around the caller's Psi / Psi^T ----
int lpc_set_psi_gram(lpc_handle e, const real* dev_gabs, void* stream) {
  if (!e || !dev_gabs) return fail("lpc_set_psi_gram: null argument");
  if (e->cfg.algo != LPC_ALGO_ADMM) return fail("lpc_set_psi_gram: ADMM handles only");
  e->stream = (lpcStream_t)stream;
  const PlaneGeom& g = e->g;
  LPC_RT(rt::memset_async(e->Gabs, 0, (size_t)g.cplane * sizeof(real), e->stream));
  LPC_OK(launch_k(e, -1, k_permute_spectrum_rows<256>, grid1d((long)g.Hp * g.Wc, 256), 256, 0, dev_gabs, e->Gabs, g.Hp,
                  g.Wc, g.cpitch, e->N1, e->N2));
  return admm_split_gram(e);
}

int lpc_admm_psi_step(lpc_handle e, const real* dev_psit, void* stream) {
  LPC_OK(pnp_check(e, "lpc_admm_psi_step"));
  if (!dev_psit) return fail("lpc_admm_psi_step: null argument");
  if (e->pnp_pending) return fail("lpc_admm_psi_step: a plug-and-play iteration is in flight");
  if (!e->pnp_mode && e->iters_done != 0)
    return fail("lpc_admm_psi_step: fused iterations already ran since the last reset");
  e->stream = (lpcStream_t)stream;
  e->pnp_mode = true;       // explicit state from here on; lpc_iterate refuses until the next reset
  const PlaneGeom& g = e->g;
  const int nimg = e->cfg.batch * e->cfg.depth;
  real *T = e->eta1[0], *X = e->eta0[1], *W = e->eta1[1];
  LPC_OK(hwc_to_planar(e, dev_psit, T, nimg, g.Hp, g.Wp, g.rpitch, g.rplane));
  double par[4];
  admm_params(e, e->iters_done, par);
  const AdmmScalars sc = admm_scalars(e, par);
  const dim3 grid = grid1d((long)g.Hp * g.Wp, 256, e->P);
  real* Vc = e->V[e->vcur];
  real* Vn = e->V[e->vcur ^ 1];
  real* HVn = e->HVb[e->hcur ^ 1];
  LPC_OK(launch_k(e, LPC_K_SPATIAL, k_pnp_pre<256>, grid, 256, 0, g, sc, 2, (const real*)Vc,
                  (const real*)e->HVb[e->hcur], (const real*)e->xi, (const real*)e->rho, (const real*)T,
                  (const real*)e->eta0[0], (const real*)e->Y, X, W, e->Rsp, e->Aarr));
  LPC_OK(admm_spectral_step(e, sc, Vn, HVn));
  LPC_OK(launch_k(e, LPC_K_SPATIAL, k_pnp_post<256>, grid, 256, 0, g, sc, 0, (const real*)Vn, (const real*)HVn,
                  (const real*)X, (const real*)W, (const real*)T, e->xi, e->eta0[0], e->rho));   // xi and rho (eta is the caller's)
  e->vcur ^= 1;
  e->hcur ^= 1;
  e->first = false;
  ++e->iters_done;
  return 0;
}

int lpc_form_image(lpc_handle e, real* dev_out, void* stream) {
  if (!e || !dev_out) return fail("lpc_form_image: null argument");
  e->stream = (lpcStream_t)stream;
  const PlaneGeom& g = e->g;
  const int nimg = e->cfg.batch * e->cfg.depth;
  if (e->cfg.algo == LPC_ALGO_ADMM && e->has_init && e->iters_done == 0 && e->init_est) {
    // Right after reset() the reference's state still ALIASES the stored initial estimate (admm.py:154-155,
    // `self._image_est = self._initial_est`), so this read-out's in-place clamp (admm.py:337) lands in the initial
    // estimate too: every later reset() starts from the clamped one.  (apply(plot/save=...) does exactly this
    // before its loop, recon.py:563-566.)
    LPC_OK(launch_k(e, -1, k_clamp_window_inplace<256>, grid1d((long)g.H * g.W, 256, e->P), 256, 0, g, e->init_est));
  }
  if (e->cfg.algo == LPC_ALGO_ADMM && e->pnp_mode) {   // explicit state: the clamp really is in place
    LPC_OK(planar_to_hwc(e, e->V[e->vcur], dev_out, nimg, g.H, g.W, g.rpitch, g.rplane, g.sh, g.sw, 1));
    return launch_k(e, -1, k_clamp_window_inplace<256>, grid1d((long)g.H * g.W, 256, e->P), 256, 0, g,
                    e->V[e->vcur]);
  }
  if (e->cfg.algo == LPC_ALGO_ADMM) {  // crop + clamp (admm.py:331-338)
    LPC_OK(planar_to_hwc(e, e->V[e->vcur], dev_out, nimg, g.H, g.W, g.rpitch, g.rplane, g.sh, g.sw, 1));
    // ... which the reference applies IN PLACE to its state: the W-updates of the next two iterations see the clamped
    // estimate (everything else keeps using the un-clamped V, exactly like the reference's cached _Psi_out /
    // _forward_out do).  clamp(V) is recomputed where it is needed (AdmmScalars::clamp_cur / clamp_old): no copy.
    e->vw_cur = true;
    return 0;
  }
  if (e->cfg.algo >= LPC_ALGO_GD)    // projection (gd.py:136-140)
    return planar_to_hwc(e, e->gx, dev_out, nimg, g.H, g.W, g.W, g.uplane, 0, 0, 1);
  return fail("lpc_form_image: operator-only handle");
}

int lpc_get_state(lpc_handle e, const char* name, real* dev_out, void* stream) {
  if (!e || !name || !dev_out) return fail("lpc_get_state: null argument");
  e->stream = (lpcStream_t)stream;
  const PlaneGeom& g = e->g;
  const std::string nm(name);
  const int nimg = e->cfg.batch * e->cfg.depth;
  if (e->cfg.algo >= LPC_ALGO_GD) return gd_get_state(e, nm, dev_out);
  if (e->cfg.algo != LPC_ALGO_ADMM) return fail("lpc_get_state: operator-only handle");
  auto out_padded = [&](real* src) {
    return planar_to_hwc(e, src, dev_out, nimg, g.Hp, g.Wp, g.rpitch, g.rplane, 0, 0, 0);
  };
  if (e->pnp_mode) {   // explicit state arrays; U and eta are image-shaped here
    if (nm == "image_est") return out_padded(e->V[e->vcur]);
    if (nm == "forward_out") return out_padded(e->HVb[e->hcur]);
    if (nm == "xi") return out_padded(e->xi);
    if (nm == "rho") return out_padded(e->rho);
    if (nm == "eta") return out_padded(e->eta0[0]);
    if (nm == "U") return out_padded(e->eta1[0]);
    if (nm == "X") return out_padded(e->eta0[1]);
    if (nm == "W") return out_padded(e->eta1[1]);
    return fail("lpc_get_state: unknown name '" + nm + "'");
  }
  if (nm == "image_est")     // after a read-out: the clamped estimate, like the reference's attribute
    return planar_to_hwc(e, e->V[e->vcur], dev_out, nimg, g.Hp, g.Wp, g.rpitch, g.rplane, 0, 0, e->vw_cur ? 2 : 0);
  if (nm == "forward_out") return out_padded(e->HVb[e->hcur]);
  // the rest needs the pending dual update applied: materialise what was asked for into the two padded arrays that are
  // idle between iterations (r_sp and a: no allocation, no host synchronisation)
  int w0 = -1, w1 = -1;
  if (nm == "xi") w0 = 0;
  else if (nm == "rho") w0 = 3;
  else if (nm == "W") w0 = 6;
  else if (nm == "X") w0 = 7;
  else if (nm == "eta") { w0 = 1; w1 = 2; }
  else if (nm == "U") { w0 = 4; w1 = 5; }
  else return fail("lpc_get_state: unknown name '" + nm + "'");
  double par[4];
  admm_params(e, e->iters_done, par);
  AdmmScalars sc = admm_scalars(e, par);
  sc.clamp_old = e->vw_old ? 1 : 0;
  LPC_OK(launch_k(e, -1, k_admm_flush<256>, grid1d((long)g.Hp * g.Wp, 256, e->P), 256, 0, g, sc,
                  (const real*)e->V[e->vcur], (const real*)e->V[e->vcur ^ 1], (const real*)e->HVb[e->hcur],
                  (const real*)e->HVb[e->hcur ^ 1], (const real*)e->Y, (const real*)e->xi, (const real*)e->eta0[e->ecur],
                  (const real*)e->eta1[e->ecur], (const real*)e->rho, e->Rsp, w1 >= 0 ? e->Aarr : (real*)nullptr, w0,
                  w1 >= 0 ? w1 : 0));
  if (w1 < 0) return out_padded(e->Rsp);
  const long n = (long)g.Hp * g.Wp * e->cfg.channels;
  return launch_k(e, -1, k_planar2_to_hwc2<256>, grid1d(n, 256, nimg), 256, 0, (const real*)e->Rsp,
                  (const real*)e->Aarr, dev_out, g.Hp, g.Wp, e->cfg.channels, g.rpitch, g.rplane);
}

// ---- evaluation reductions (section 8f row N2): nothing here synchronises with the host ----
int lpc_reconstruction_error(lpc_handle e, const real* dev_pred, const real* dev_data, int normalize,
                             real* dev_out, void* stream) {
  if (!e || !dev_pred || !dev_out) return fail("lpc_reconstruction_error: null argument");
  if (!e->psf_set) return fail("lpc_reconstruction_error: PSF not set");
  if (!dev_data && !e->data_set) return fail("lpc_reconstruction_error: no data (lpc_set_data or dev_data)");
  e->stream = (lpcStream_t)stream;
  const PlaneGeom& g = e->g;
  const size_t up = (size_t)g.uplane * e->P;
  e->gd_fwd_done = false;     // the work spectrum is about to be overwritten
  // two un-padded planar staging arrays out of buffers that are dead between iterations
  real *xin = nullptr, *xout = nullptr;
  if (e->cfg.algo == LPC_ALGO_ADMM) { xin = e->Rsp; xout = e->Aarr; }
  else if (e->cfg.algo == LPC_ALGO_CONV) { xin = e->gaux; xout = e->gx; }
  else { xin = (real*)e->S2; xout = xin + up; }
  const int nimg = e->cfg.batch * e->cfg.depth;
  LPC_OK(hwc_to_planar(e, dev_pred, xin, nimg, g.H, g.W, g.W, g.uplane));
  LPC_OK(convolve_planar(e, xin, xout, e->P, false, false));   // H x, cropped (recon.py:632-638)
  // partials live in the work spectrum, free again once the convolution has been enqueued
  const int nblk = (int)std::max<long>(1, std::min<long>(64, g.uplane / 1024));
  if ((4 * (size_t)e->P * nblk + 2 * (size_t)nimg + 8) * sizeof(real) + (size_t)e->P * nblk * sizeof(double) >
      (size_t)g.cplane * e->P * sizeof(real2))
    return fail("lpc_reconstruction_error: frame too small for the reduction scratch");
  real* mm = (real*)e->S;                                     // [P][nblk](max, min)
  real* rng = mm + 2 * (size_t)e->P * nblk;                   // [B*D](min, max - min)
  double* part = (double*)(rng + 2 * (size_t)nimg + 2);       // [P][nblk]
  part = (double*)(((uintptr_t)part + 7) & ~(uintptr_t)7);
  if (normalize) {
    LPC_OK(launch_k(e, -1, k_plane_minmax<256>, dim3(nblk, e->P), 256, 2 * 256 * sizeof(real), g,
                    (const real2*)nullptr, (const real*)xout, 1, mm));
    LPC_OK(launch_k(e, -1, k_item_range, dim3((nimg + 63) / 64), 64, 0, (const real*)mm, nblk,
                    e->cfg.channels, nimg, rng));
  }
  LPC_OK(launch_k(e, -1, k_sqerr<256>, dim3(nblk, e->P), 256, 256 * sizeof(double), g, (const real*)xout,
                  dev_data ? dev_data : (const real*)e->Y, dev_data ? 1 : 0,
                  (const real*)(normalize ? rng : nullptr), part));
  const double npix = (double)e->cfg.depth * g.H * g.W * e->cfg.channels;   // recon.py:258
  LPC_OK(launch_k(e, -1, k_sqerr_finish, dim3((e->cfg.batch + 63) / 64), 64, 0, (const double*)part, nblk,
                  g.DC, e->cfg.batch, npix, dev_out));
  return 0;
}

int lpc_image_metrics(const real* dev_true, const real* dev_est, long n, int n_items, int normalize,
                      real* dev_out, void* stream) {
  if (!dev_true || !dev_est || !dev_out) return fail("lpc_image_metrics: null argument");
  if (n < 1 || n_items < 1) return fail("lpc_image_metrics: empty input");
  Engine tmp;                       // launch context only (stream); owns nothing
  tmp.stream = (lpcStream_t)stream;
  Engine* e = &tmp;
  const int nblk = (int)std::max<long>(1, std::min<long>(1024, n / 4096));
  const size_t nr = (size_t)n_items * nblk;
  void* scratch = nullptr;          // [2 nr] + [2 nr] min/max partials, 2 x [2 items] ranges, [nr] doubles
  const size_t bytes = (4 * nr + 4 * (size_t)n_items) * sizeof(real) + 16 + nr * sizeof(double);
  LPC_RT(rt::dev_malloc_async(&scratch, bytes, e->stream));
  real* pt = (real*)scratch;
  real* px = pt + 2 * nr;
  real* rt_ = px + 2 * nr;
  real* rx = rt_ + 2 * n_items;
  double* part = (double*)(((uintptr_t)(rx + 2 * n_items) + 7) & ~(uintptr_t)7);
  int rc = launch_k(e, -1, k_flat_minmax<256>, dim3(nblk, n_items), 256, 2 * 256 * sizeof(real), dev_true, n, pt);
  if (!rc) rc = launch_k(e, -1, k_flat_minmax<256>, dim3(nblk, n_items), 256, 2 * 256 * sizeof(real), dev_est, n, px);
  if (!rc) rc = launch_k(e, -1, k_flat_range, dim3((n_items + 63) / 64), 64, 0, (const real*)pt, nblk, n_items, rt_);
  if (!rc) rc = launch_k(e, -1, k_flat_range, dim3((n_items + 63) / 64), 64, 0, (const real*)px, nblk, n_items, rx);
  if (!rc) rc = launch_k(e, -1, k_pair_sqdiff<256>, dim3(nblk, n_items), 256, 256 * sizeof(double), dev_true,
                         dev_est, n, (const real*)rt_, (const real*)rx, normalize, part);
  if (!rc) rc = launch_k(e, -1, k_pair_finish, dim3((n_items + 63) / 64), 64, 0, (const double*)part, nblk,
                         n_items, n, (const real*)rt_, normalize, dev_out);
  (void)rt::dev_free_async(scratch, e->stream);   // stream-ordered: freed after the kernels, no host sync
  return rc;
}

// ---- raw-frame preparation (section 8f row N3): handle-free, results stay on the device ----
static int prep_geom(const lpc_prep_config* c, PrepGeom* g, const char* who) {
  if (!c) return fail(std::string(who) + ": null config");
  if (c->height < 1 || c->width < 1) return fail(std::string(who) + ": bad spatial size");
  if (c->channels != 1 && c->channels != 3) return fail(std::string(who) + ": channels must be 1 or 3");
  if (c->raw_type < LPC_RAW_U8 || c->raw_type > LPC_RAW_F64) return fail(std::string(who) + ": unknown raw type");
  g->H = c->height; g->W = c->width; g->Cin = c->channels;
  g->flip_ud = c->flip_ud != 0; g->flip_lr = c->flip_lr != 0; g->rev = c->bgr_input != 0 && c->channels == 3;
  g->gray = c->gray != 0; g->raw_type = c->raw_type; g->p0 = c->bg_pix0; g->p1 = c->bg_pix1;
  g->single = c->single_psf != 0 && c->channels == 3; g->normalize = c->normalize != 0;
  g->Cout = (g->gray && g->Cin == 3) ? 1 : g->Cin;
  return 0;
}

int lpc_preprocess_frames(const lpc_prep_config* cfg, const void* dev_raw, int n, const real* dev_bg,
                          real* dev_out, void* stream) {
  PrepGeom g;
  LPC_OK(prep_geom(cfg, &g, "lpc_preprocess_frames"));
  if (!dev_raw || !dev_out || n < 1) return fail("lpc_preprocess_frames: null / empty argument");
  Engine tmp;
  tmp.stream = (lpcStream_t)stream;
  Engine* e = &tmp;
  const long npx = (long)g.H * g.W;
  const int nblk = (int)std::max<long>(1, std::min<long>(512, npx / 4096));
  void* scratch = nullptr;
  LPC_RT(rt::dev_malloc_async(&scratch, ((size_t)n * g.Cin * nblk + 5 * (size_t)n) * sizeof(real), e->stream));
  real* partial = (real*)scratch;
  real* par = partial + (size_t)n * g.Cin * nblk;
  int rc = launch_k(e, -1, k_prep_chanmax<256>, dim3(nblk, n * g.Cin), 256, 2 * 256 * sizeof(real), g, dev_raw,
                    partial);
  if (!rc) rc = launch_k(e, -1, k_prep_frame_params, dim3((n + 63) / 64), 64, 0, g, (const real*)partial, nblk,
                         dev_bg, n, par);
  if (!rc) rc = launch_k(e, -1, k_prep_frame<256>, grid1d(npx, 256, n), 256, 0, g, dev_raw, (const real*)par,
                         dev_out);
  (void)rt::dev_free_async(scratch, e->stream);
  return rc;
}

int lpc_preprocess_psf(const lpc_prep_config* cfg, const void* dev_raw, int depth, real* dev_psf_out,
                       real* dev_bg_out, void* stream) {
  PrepGeom g;
  LPC_OK(prep_geom(cfg, &g, "lpc_preprocess_psf"));
  if (!dev_raw || !dev_psf_out || depth < 1) return fail("lpc_preprocess_psf: null / empty argument");
  const bool has_bg = g.p1 > g.p0;
  if (has_bg && (g.p0 < 0 || g.p1 > g.H || g.p1 > g.W)) return fail("lpc_preprocess_psf: bg_pix outside the frame");
  int rep = g.Cin;
  if (g.single) {
    rep = cfg->out_channels;
    if (rep != 1 && rep != 3) return fail("lpc_preprocess_psf: out_channels must be 1 or 3 with single_psf");
  }
  Engine tmp;
  tmp.stream = (lpcStream_t)stream;
  Engine* e = &tmp;
  const long npx = (long)depth * g.H * g.W;
  const int nblk = (int)std::max<long>(1, std::min<long>(1024, npx / 4096));
  void* scratch = nullptr;
  LPC_RT(rt::dev_malloc_async(&scratch, (size_t)nblk * sizeof(double) + ((size_t)nblk + 8) * sizeof(real),
                              e->stream));
  double* psum = (double*)scratch;
  real* pmax = (real*)(psum + nblk);
  real* bgv = pmax + nblk;   // [3]
  real* nrm = bgv + 3;       // [1]
  int rc = 0;
  if (has_bg)
    rc = launch_k(e, -1, k_prep_psf_bg<256>, dim3(g.Cin), 256, 256 * sizeof(double), g, dev_raw, depth, bgv);
  if (!rc) rc = launch_k(e, -1, k_prep_psf_energy<256>, dim3(nblk), 256, 256 * sizeof(double) + 2 * 256 * sizeof(real),
                         g, dev_raw, depth, (const real*)bgv, (int)has_bg, psum, pmax);
  if (!rc) rc = launch_k(e, -1, k_prep_psf_finish, dim3(1), 64, 0, g, (const double*)psum, (const real*)pmax, nblk,
                         (const real*)bgv, (int)has_bg, nrm, dev_bg_out);
  if (!rc) rc = launch_k(e, -1, k_prep_psf<256>, grid1d(npx, 256), 256, 0, g, dev_raw, depth, (const real*)bgv,
                         (int)has_bg, (const real*)nrm, rep, dev_psf_out);
  (void)rt::dev_free_async(scratch, e->stream);
  return rc;
}

int lpc_resize_aa(const real* dev_in, int n, int H, int W, int C, int Hout, int Wout, real* dev_out, void* stream) {
  if (!dev_in || !dev_out) return fail("lpc_resize_aa: null argument");
  if (n < 1 || H < 1 || W < 1 || C < 1 || Hout < 1 || Wout < 1) return fail("lpc_resize_aa: bad size");
  Engine tmp;
  tmp.stream = (lpcStream_t)stream;
  Engine* e = &tmp;
  const long nin = (long)n * H * W * C;
  const int nblk = (int)std::max<long>(1, std::min<long>(1024, nin / 4096));
  void* scratch = nullptr;        // [n][H][Wout][C] intermediate, then 2 * nblk partials and the (max, min) pair
  const size_t mid = (size_t)n * H * Wout * C;
  LPC_RT(rt::dev_malloc_async(&scratch, (mid + 2 * (size_t)nblk + 2) * sizeof(real), e->stream));
  real* tmpbuf = (real*)scratch;
  real* part = tmpbuf + mid;
  real* rng = part + 2 * (size_t)nblk;
  int rc = launch_k(e, -1, k_flat_minmax<256>, dim3(nblk, 1), 256, 2 * 256 * sizeof(real), dev_in, nin, part);
  if (!rc) rc = launch_k(e, -1, k_flat_range, dim3(1), 64, 0, (const real*)part, nblk, 1, rng);
  // image.py:59-64: the last spatial axis first (aten's separable kernel), then the rows
  if (!rc) rc = launch_k(e, -1, k_resize_aa_axis<256>, grid1d((long)mid, 256), 256, 0, dev_in, tmpbuf, (long)n * H, W, Wout,
                         (long)C, (const real*)nullptr);
  if (!rc) rc = launch_k(e, -1, k_resize_aa_axis<256>, grid1d((long)n * Hout * Wout * C, 256), 256, 0, (const real*)tmpbuf,
                         dev_out, (long)n, H, Hout, (long)Wout * C, (const real*)rng);
  (void)rt::dev_free_async(scratch, e->stream);
  return rc;
}

int lpc_profile_enable(lpc_handle e, int on) {
  if (!e) return fail("null handle");
#if !defined(LPC_SIMT_EMU)
  for (int k = 0; k < LPC_K_COUNT; ++k) e->timer.used[k] = 0;
#endif
  e->timer.on = on != 0;
  e->timer.mask = on > 1 ? (unsigned)on >> 1 : ~0u;      // 1: every hot-loop kernel; otherwise bit k + 1 selects kernel id k
  return 0;
}

int lpc_profile_read(lpc_handle e, double* avg_ms, long* launches) {
  if (!e || !avg_ms || !launches) return fail("null argument");
  for (int k = 0; k < LPC_K_COUNT; ++k) { avg_ms[k] = 0.0; launches[k] = 0; }
#if !defined(LPC_SIMT_EMU)
  LPC_RT(hipStreamSynchronize(e->stream));
  for (int k = 0; k < LPC_K_COUNT; ++k) {
    double tot = 0.0;
    for (size_t i = 0; i < e->timer.used[k]; ++i) {
      float ms = 0.f;  // HIP API type, not the engine's arithmetic type
      LPC_RT(hipEventElapsedTime(&ms, e->timer.ev[k][i].first, e->timer.ev[k][i].second));
      tot += ms;
    }
    launches[k] = (long)e->timer.used[k];
    avg_ms[k] = e->timer.used[k] ? tot / (double)e->timer.used[k] : 0.0;
  }
#endif
  return 0;
}

int lpc_kernel_bytes(lpc_handle e, int kid, double* bytes) {
  if (!e || !bytes) return fail("null argument");
  const PlaneGeom& g = e->g;
  const double eb = (double)sizeof(real);             // 4 (liblpc) or 8 (liblpc_f64)
  const double R = eb * g.Hp * g.Wp * e->P;           // padded real arrays, all planes
  const double S = 2 * eb * g.Hp * g.Wc * e->P;       // half spectra
  const double R0 = eb * g.H * g.W * e->Pdata;
  const double Sc = 2 * eb * g.Hp * g.Wc * e->Ppsf;   // spectral constants
  const bool split = e->N1 > 1;
  const double fr = (double)g.H / (double)g.Hp;
  double b = 0.0;
  if (e->cfg.algo == LPC_ALGO_ADMM) {
    switch (kid) {
      // SURVEY 8(d) figure for the stand-alone kernel (reads 8R+R0, writes 7R; the kernel itself moves 14R + R0: X is
      // recomputed instead of stored).  Fused into the forward rows it reads 8R + R0 (V, V_old, HV, HV_old, xi, eta0,
      // eta1, rho; y) and writes xi, eta0, eta1, rho (4R) + the two row spectra (2S): r_sp and a never reach HBM.
      // X half in the forward rows (default with compile-time row plans): the tiled kernel reads V, V_old, eta0, eta1,
      // rho and writes eta0, eta1, rho, r_sp = 9R (SURVEY's 15R + R0 minus its X part: reads HV, X, xi, y, writes xi, X,
      // a); the row kernel reads r_sp (R) and xi, HV, HV_old, y (3R + R0), writes xi (R) and the two spectra (2S).
      // ... and without V_old once the duals travel half-applied between the iterations of a call (k1_half): 8R
      // ... k1_rows (small frames): not launched; the forward rows read V, eta0, eta1, rho (+ V_old without k1_half)
      // instead of r_sp and write eta0, eta1, rho: + 6R (7R)
      case LPC_K_SPATIAL: b = e->k1_rows ? 0.0 : e->xhalf_rows ? ((e->opt.k1_half != 0 && g.Wp % 4 == 0) ? 8.0 : 9.0) * R
                                            : 15.0 * R + R0; break;
      // ... with xi confined to the sensor window (AdmmScalars::xiw) the row kernel reads r_sp, HV everywhere (2R) and
      // xi, HV_old / writes xi only over the window (3 window-sized arrays per plane) and y: 2R + 3 Rw + R0 + 2S
      // ... and with the H V row transforms skipped on rows wholly outside the window (AdmmScalars::skipa, steady state
      // of a long call; fr = H / Hp): rows fwd (1 + fr) R + 3 Rw + R0 + (1 + fr) S, rows inv (1 + fr) (S + R)
      case LPC_K_ROW_FWD: b = (e->hv_skip ? (1.0 + fr) * R + 3.0 * eb * g.H * g.W * e->P + R0 + (1.0 + fr) * S
                                  : e->xi_window ? 2.0 * R + 3.0 * eb * g.H * g.W * e->P + R0 + 2.0 * S
                                  : e->xhalf_rows ? 5.0 * R + R0 + 2.0 * S : 2.0 * R + 2.0 * S)
                                 + (e->k1_rows ? (e->opt.k1_half != 0 ? 6.0 : 7.0) * R : 0.0); break;
      case LPC_K_COL_A_FWD: b = split ? 4.0 * S : 0.0; break;
      case LPC_K_COL_MID: b = 4.0 * S + Sc + (e->g_sep ? 0. : eb * g.Hp * g.Wc); break;  // + H (complex) + |G| (real, one plane; two vectors when it separates)
      case LPC_K_COL_A_INV: b = split ? 4.0 * S : 0.0; break;
      case LPC_K_ROW_INV: b = e->hv_skip ? (1.0 + fr) * (S + R) : 2.0 * S + 2.0 * R; break;
      default: return fail("bad kernel id");
    }
  } else if (e->cfg.algo >= LPC_ALGO_GD) {
    return gd_kernel_bytes(e, kid, bytes);
  } else {
    return fail("lpc_kernel_bytes: operator-only handle");
  }
  *bytes = b;
  return 0;
}

int lpc_plan_info(lpc_handle e, char* buf, size_t n) {
  if (!e || !buf || n == 0) return fail("null argument");
  const PlaneGeom& g = e->g;
  const PlanSpec& sp = e->spec;
  auto radstr = [](const StaticFft& f) {
    std::string r;
    for (int i = 0; i < f.nst; ++i) r += (i ? "." : "") + std::to_string(f.rad[i]);
    return r;
  };
  std::string s = "padded " + std::to_string(g.Hp) + "x" + std::to_string(g.Wp);
  s += e->rows_half ? "; rows: half-length " + std::to_string(g.Wp / 2) : "; rows: paired " + std::to_string(g.Wp);
  const bool rows_static = e->mod && sp.row_kind && (e->rows_half || e->cfg.algo == LPC_ALGO_ADMM);
  if (rows_static) s += " [static " + radstr(sp.row) + ", " + std::to_string(sp.row.nt) + " threads]";
  if (e->gd_v2) s += " (fused rows: second form, " + std::to_string(sp.row.n / sp.row.rad[0]) + " lanes)";
  s += "; columns: " + (e->N1 > 1 ? std::to_string(e->N1) + " x " + std::to_string(e->N2) + " split" : std::string("single pass ") + std::to_string(e->N2));
  s += ", T = " + std::to_string(e->T);
  if (e->mod && sp.passA.n) s += ", pass A [static " + radstr(sp.passA) + ", T = " + std::to_string(sp.passA.T) + "]";
  if (e->cfg.algo == LPC_ALGO_ADMM) {
    const bool reg = e->N1 > 1 && e->mid_reg && sizeof(real) == 4 && e->N2 == 24;
    s += reg ? ", middle in registers"
             : (e->mod && sp.mid_kind ? ", LDS middle [static " + radstr(sp.mid) + (sp.mid_kind == LPC_MID_SEQ ? ", one spectrum at a time" : "") + (g.slay ? ", pair-line spectra]" : "]")
                                      : ", LDS middle");
    s += e->k1_rows ? "; TV / W half and X half inside the forward rows (three launches per iteration)"
         : e->xhalf_rows ? "; tiled TV / W kernel + X half inside the forward rows" : "; stand-alone image-domain kernel";
    if (e->xi_window) s += e->hv_skip ? " (xi inside the sensor window only, H V row transforms skipped outside it)"
                                      : " (xi inside the sensor window only)";
  }
  if (e->cfg.algo == LPC_ALGO_ADMM && e->g_sep) s += "; gram as row + column terms";
  s += e->mod ? "; plan module " + plan_spec_key(sp) : "; run-time plans (" + e->mod_note + ")";
  std::snprintf(buf, n, "%s", s.c_str());
  return 0;
}

int lpc_model_bytes(lpc_handle e, double* bytes) {
  if (!e || !bytes) return fail("null argument");
  const PlaneGeom& g = e->g;
  const double eb = (double)sizeof(real);
  const double R = eb * g.Hp * g.Wp * e->P, S = 2 * eb * g.Hp * g.Wc * e->P;
  if (e->cfg.algo == LPC_ALGO_ADMM) *bytes = 19.0 * R + eb * g.H * g.W * e->Pdata + 13.5 * S;
  else if (e->cfg.algo >= LPC_ALGO_GD) *bytes = (e->cfg.algo == LPC_ALGO_GD ? 6.0 : 8.0) * eb * g.H * g.W * e->P + 14.0 * S;
  else return fail("lpc_model_bytes: operator-only handle");
  return 0;
}

}  // extern "C"
