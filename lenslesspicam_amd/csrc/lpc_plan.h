// lpc_plan.h -- engine options and the description of a PLAN MODULE (host only).
//
// Every transform the hot loop runs has two implementations:
//   * run-time plans (lpc_fft.h): one kernel per workgroup shape serves any 5-smooth length, radices and strides are
//     kernel arguments -- always available, compiled into liblpc.so;
//   * compile-time plans (lpc_sfft.h): the same kernel sources instantiated on SPlan<radices...>, 20-25 % faster per
//     byte (half the registers, no index arithmetic).  A frame shape needs its OWN instantiations, so they live in a
//     small shared object per shape -- a "plan module" (lpc_module.cpp compiled with the -D flags that
//     plan_spec_defines() produces, ~3 s of hipcc) -- which lpc_create() loads from <libdir>/modules/, compiling it
//     first when it is missing (lpc_jit.cpp).  build.py pre-builds the modules of BASELINE.json's shapes; any other
//     5-smooth frame (the reference accepts them all, rfft_convolve.py:110-117) gets the same kernels on first use.
// PlanSpec is everything that is a template argument in a module; its key names the file.
#pragma once
#include <cctype>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#define LPC_SPEC_MAX_ST 8

struct StaticFft {            // one compile-time transform: length, radices, workgroup shape
  int n = 0;                  // 0: not static (run-time plan)
  int nst = 0;
  int rad[LPC_SPEC_MAX_ST] = {0};
  int nt = 0, em = 0;         // threads per workgroup x points per thread (nt * em >= n * tile columns)
  int T = 1;                  // tile columns (column passes)
};

// (macros, not enumerators: lpc_module.cpp tests them in #if)
#define LPC_FAM_ADMM 1
#define LPC_FAM_GD 2            // gradient-descent family and the bare operator
#define LPC_ROWS_RUNTIME 0
#define LPC_ROWS_HALF 1
#define LPC_ROWS_PAIRED 2
#define LPC_MID_RUNTIME 0
#define LPC_MID_PAIR 1          // ADMM LDS middle: both spectra side by side
#define LPC_MID_SEQ 2           // ... one at a time

struct PlanSpec {
  int family = 0;
  int f64 = 0;
  int row_kind = LPC_ROWS_RUNTIME;
  StaticFft row;              // length Wp / 2 (half) or Wp (paired)
  int row_sk = 0;             // LDS layout of the row tile: 0 natural, 1 i + i/8 (lpc_fft.h)
  int row_x = 0;              // ADMM: forward rows with the X half of the image-domain work
  StaticFft passA;            // pass A of a split column transform (T = 32 | 16 | narrower)
  int mid_kind = LPC_MID_RUNTIME;
  StaticFft mid;              // ADMM fused middle in LDS
  int mid_minw = 1;           // __launch_bounds__ second argument of the sequential middle
  int mid_pre = 0;            // sequential middle: both tiles' loads issued before the first transform
  int mid_pc = 0;             // sequential middle on pair-line spectra: point-wise constants precombined (k_mid_consts): 1 two
                              // complex constants per element, 2 one (even padded sizes: the ifftshift phases are +-1)
  int slay = 0;               // ADMM work spectra in pair lines (lpc_kernels.h: spec_col): paired rows + 8-column sequential middle
  bool any() const { return row_kind != LPC_ROWS_RUNTIME || passA.n || mid_kind != LPC_MID_RUNTIME; }
};

static inline std::string fft_key(const StaticFft& f) {
  std::string s = std::to_string(f.n) + "r";
  for (int i = 0; i < f.nst; ++i) s += (i ? "." : "") + std::to_string(f.rad[i]);
  s += "t" + std::to_string(f.T) + "w" + std::to_string(f.nt) + "x" + std::to_string(f.em);
  return s;
}
static inline std::string plan_spec_key(const PlanSpec& s) {
  std::string k = s.f64 ? "f64" : "f32";
  k += s.family == LPC_FAM_ADMM ? "_admm" : "_gd";
  if (s.row_kind) k += std::string(s.row_kind == LPC_ROWS_HALF ? "_rh" : "_rp") + fft_key(s.row) + (s.row_sk == 1 ? "s" : "") + (s.row_x ? "x" : "");
  if (s.passA.n) k += "_a" + fft_key(s.passA);
  if (s.mid_kind) k += std::string(s.mid_kind == LPC_MID_PAIR ? "_mp" : "_ms") + fft_key(s.mid) + "m" + std::to_string(s.mid_minw) + (s.mid_pre ? "p" : "") + (s.slay ? "L" : "") + (s.mid_pc == 1 ? "c" : (s.mid_pc == 2 ? "r" : ""));
  return k;
}
static inline std::string rad_list(const StaticFft& f) {
  std::string s;
  for (int i = 0; i < f.nst; ++i) s += (i ? "," : "") + std::to_string(f.rad[i]);
  return s;
}
// the -D flags lpc_module.cpp is compiled with
static inline std::vector<std::string> plan_spec_defines(const PlanSpec& s) {
  std::vector<std::string> d;
  auto def = [&](const std::string& k, const std::string& v) { d.push_back("-D" + k + "=" + v); };
  auto defi = [&](const std::string& k, int v) { def(k, std::to_string(v)); };
  if (s.f64) d.push_back("-DLPC_DOUBLE");
  defi("LPC_MOD_FAMILY", s.family);
  defi("LPC_MOD_ROW_KIND", s.row_kind);
  if (s.row_kind) {
    def("LPC_MOD_ROW_RAD", rad_list(s.row));
    defi("LPC_MOD_ROW_NT", s.row.nt); defi("LPC_MOD_ROW_EM", s.row.em);
    defi("LPC_MOD_ROW_SK", s.row_sk); defi("LPC_MOD_ROW_X", s.row_x);
  }
  defi("LPC_MOD_PASSA", s.passA.n ? 1 : 0);
  if (s.passA.n) {
    def("LPC_MOD_PASSA_RAD", rad_list(s.passA));
    defi("LPC_MOD_PASSA_NT", s.passA.nt); defi("LPC_MOD_PASSA_EM", s.passA.em); defi("LPC_MOD_PASSA_T", s.passA.T);
  }
  defi("LPC_MOD_MID_KIND", s.mid_kind);
  if (s.mid_kind) {
    def("LPC_MOD_MID_RAD", rad_list(s.mid));
    defi("LPC_MOD_MID_NT", s.mid.nt); defi("LPC_MOD_MID_EM", s.mid.em); defi("LPC_MOD_MID_T", s.mid.T);
    defi("LPC_MOD_MID_MINW", s.mid_minw);
    defi("LPC_MOD_MID_PRE", s.mid_pre);
  }
  defi("LPC_MOD_SLAY", s.slay);
  defi("LPC_MOD_MID_PC", s.mid_pc);
  return d;
}

// ---- options (lpc_config::options, "key=value,key=value"; include/lpc.h lists them) ---------------------------------
struct EngineOpts {
  // -- which kernels exist for a handle
  int no_static = 0;          // run-time plans everywhere: no plan module is looked for
  int jit = 1;                // compile a missing plan module at lpc_create (0: only modules already on disk)
  long jit_min_points = 1L << 16;   // padded frames smaller than this keep the run-time plans (launch-latency regime)
  std::string module_dir;     // where plan modules are looked for and written first (default: <libdir>/modules)
  std::string compiler;       // hipcc to compile a missing module with (default: $ROCM_PATH/bin/hipcc, /opt/rocm/bin/hipcc)
  int module_max = 256;       // module files kept in a directory this library writes to (least recently used go first)
  int module_loaded_max = 64; // modules kept dlopen()ed by a process once no handle uses them
  // -- forcing a plan the chooser takes for other sizes (tests run every kernel family on small frames with these;
  //    every setting gives valid results)
  int rows_half = -1;         // -1: by size; 0: paired rows; 1: one real row per half-length transform (even widths)
  int tile_budget = 0;        // LDS points per column tile (forces the four-step split / 8-column tiles onto small frames)
  int split_n2 = 0;           // length of the fused middle transform of a split column pass
  int col_t = 0;              // image columns per column tile
  int passa_t = 0;            // ... of pass A on a compile-time plan (32 | 16)
  int mid_seq = -1;           // single-pass ADMM middle: -1 by batch size; 1 one spectrum at a time; 0 side by side
  int mid_pre = -1;           // ... one spectrum at a time: 0 the second tile's loads behind the first transform (-1 / 1: up front)
  int prow_nt128 = -1;        // short paired rows on 128 threads: -1 by batch size
  int g_plane = -1;           // ADMM middles read |PsiT Psi| from its plane (1) / as row + column terms when it separates (0);
                              // -1: the terms when the plane is larger than 8 MB (it then misses the L2 once per colour plane)
  int mid_swz = -1;           // side-by-side LDS middle: pairs of half-line column tiles on one XCD (ColPass::swz); -1: when a
                              // tile row is narrower than a 128-byte line and the spectra are not in pair lines
  int spec_lay = -1;          // ADMM work spectra of paired rows + 8-column middles in pair lines (PlanSpec::slay); 0: rows
  std::string row_rad;        // "16.16.8": radices of the compile-time row plan instead of the chooser's (one module, ~3 s);
                              // ignored unless the product is the row length and every radix has a butterfly
  // -- the structure of an ADMM iteration (each 0 / 1 is an older, complete form of the same arithmetic)
  int hv_full = 0;            // every row of H V transformed in every iteration
  int xi_full = 0;            // xi kept on the whole padded frame
  int k1_rows = 1;            // TV / W half inside the paired forward rows where a row is one or two quads per lane (three launches)
  int k1_group = 16;          // ... on launches of more than 8192 row blocks: runs of this many consecutive blocks per XCD
                              // (K1Rows::xcd_order; 0: launch order)
  int mid_pc = 1;             // sequential middle on pair-line spectra: H, |G| and the phases precombined per (PSF, step sizes)
                              // into one 16-byte + one 4-byte load per element (PlanSpec::mid_pc); 0: loaded and combined per element
  int k1_half = 1;            // duals half-applied between the iterations of one call: the tiled kernel does not read V_old
  // -- block orders (permutations: results unchanged)
  int rev_order = 9;          // bit 0 / 1 / 2 / 3: the tiled ADMM kernel / forward pass A / inverse pass A / the LDS middle walk
                              // their grids BACKWARDS: a kernel that starts where its predecessor finished finds the last
                              // ~256 MB that one wrote in the memory-side cache (C4 -3.3 %, C5 -2.5 %, C2 -1.5 %)
  int gd_rev = -1;            // gradient-descent family / operator, backwards: bit 0 residual rows, 1 update rows, 2 the
                              // register middle.  -1: all three when a work spectrum is larger than the memory-side cache
  // -- gradient-descent family
  int gd_no_fuse_fwd = 0;     // update without the next iteration's forward rows
  int gd_v2 = 1;              // half-length rows on a one-radix plan: the fused row kernels of lpc_gd_v2_kernels.h (0: first form)
};

// path-valued options (module_dir, compiler) may contain the separators below as %XX escapes ("%2C" = ','; "%25" = '%')
static inline std::string unescape_opt_path(const std::string& v) {
  std::string r;
  for (size_t i = 0; i < v.size(); ++i) {
    if (v[i] == '%' && i + 2 < v.size() && std::isxdigit((unsigned char)v[i + 1]) && std::isxdigit((unsigned char)v[i + 2])) {
      r += (char)std::strtol(v.substr(i + 1, 2).c_str(), nullptr, 16);
      i += 2;
    } else {
      r += v[i];
    }
  }
  return r;
}

// parses "k=v,k=v" (also ';' or whitespace as separators) into o; returns "" or an error message
static inline std::string parse_engine_opts(const char* str, EngineOpts& o) {
  if (!str) return "";
  std::string s(str);
  size_t i = 0;
  while (i < s.size()) {
    size_t j = s.find_first_of(",; \t\n", i);
    if (j == std::string::npos) j = s.size();
    if (j > i) {
      const std::string tok = s.substr(i, j - i);
      const size_t eq = tok.find('=');
      const std::string k = tok.substr(0, eq), v = eq == std::string::npos ? "1" : tok.substr(eq + 1);
      const long iv = std::atol(v.c_str());
      if (k == "no_static") o.no_static = (int)iv;
      else if (k == "jit") o.jit = (int)iv;
      else if (k == "jit_min_points") o.jit_min_points = iv;
      else if (k == "module_dir") o.module_dir = unescape_opt_path(v);
      else if (k == "compiler") o.compiler = unescape_opt_path(v);
      else if (k == "module_max") o.module_max = (int)iv;
      else if (k == "module_loaded_max") o.module_loaded_max = (int)iv;
      else if (k == "rows_half") o.rows_half = (int)iv;
      else if (k == "tile_budget") o.tile_budget = (int)iv;
      else if (k == "split_n2") o.split_n2 = (int)iv;
      else if (k == "col_t") o.col_t = (int)iv;
      else if (k == "passa_t") o.passa_t = (int)iv;
      else if (k == "mid_seq") o.mid_seq = (int)iv;
      else if (k == "mid_pre") o.mid_pre = (int)iv;
      else if (k == "prow_nt128") o.prow_nt128 = (int)iv;
      else if (k == "g_plane") o.g_plane = (int)iv;
      else if (k == "mid_swz") o.mid_swz = (int)iv;
      else if (k == "spec_lay") o.spec_lay = (int)iv;
      else if (k == "row_rad") o.row_rad = v;
      else if (k == "hv_full") o.hv_full = (int)iv;
      else if (k == "xi_full") o.xi_full = (int)iv;
      else if (k == "k1_rows") o.k1_rows = (int)iv;
      else if (k == "k1_group") o.k1_group = (int)iv;
      else if (k == "k1_half") o.k1_half = (int)iv;
      else if (k == "mid_pc") o.mid_pc = (int)iv;
      else if (k == "rev_order") o.rev_order = (int)iv;
      else if (k == "gd_rev") o.gd_rev = (int)iv;
      else if (k == "gd_no_fuse_fwd") o.gd_no_fuse_fwd = (int)iv;
      else if (k == "gd_v2") o.gd_v2 = (int)iv;
      else return "unknown engine option '" + k + "'";
    }
    i = j + 1;
  }
  return "";
}
