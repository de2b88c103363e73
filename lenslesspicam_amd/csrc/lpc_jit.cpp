// lpc_jit.cpp -- plan modules (lpc_plan.h): find, compile, load.
//
// A module is lpc_module.cpp compiled for ONE PlanSpec into  <module dir>/lpcmod_<backend>_<fingerprint>_<key>.so .
// get_plan_module() serves it from the process cache, else from disk, else -- if allowed -- compiles it first:
//   product library   hipcc -O3 --offload-arch=gfx950 -shared ... lpc_module.cpp        (about 3 s per shape)
//   emulator build    g++ -O2 -DLPC_SIMT_EMU -shared ...                                (tests only, lpc_rt.h)
// The module is linked against this library (it calls fail(), and in the emulator build the fibre scheduler), which the
// loader resolves to the copy that is already in the process.  The fingerprint names the sources the core library was
// built from (build.py): a module from other sources is never picked up, and lpc_module_init() checks it again together
// with sizeof(lpc_engine).  Concurrent builders (one process per GPU) write to private temporaries and rename.
#include "lpc_engine.h"

#include <cerrno>
#include <dlfcn.h>
#include <sys/stat.h>
#include <sys/types.h>
#include <unistd.h>

#include <map>
#include <mutex>

#ifndef LPC_CSRC_REL
#define LPC_CSRC_REL "../csrc"            // relative to the directory of this library: lenslesspicam_amd/_lib -> csrc
#endif
#ifndef LPC_INCLUDE_REL
#define LPC_INCLUDE_REL "../../include"
#endif

#if defined(LPC_SIMT_EMU)
static const char* kBackendTag = "emu";
#else
static const char* kBackendTag = "hip";
#endif

static bool file_exists(const std::string& p) {
  struct stat st;
  return ::stat(p.c_str(), &st) == 0 && S_ISREG(st.st_mode);
}
static bool dir_writable(const std::string& d) {
  if (::mkdir(d.c_str(), 0755) != 0 && errno != EEXIST) return false;
  return ::access(d.c_str(), W_OK | X_OK) == 0;
}

static std::string self_path() {          // the path of this shared object
  Dl_info info;
  if (dladdr((const void*)&file_exists, &info) && info.dli_fname) {
    char buf[4096];
    if (::realpath(info.dli_fname, buf)) return buf;
    return info.dli_fname;
  }
  return "";
}
static std::string dir_of(const std::string& p) {
  const size_t k = p.rfind('/');
  return k == std::string::npos ? "." : p.substr(0, k);
}

static std::string module_file(const PlanSpec& spec) {
  return std::string("lpcmod_") + kBackendTag + "_" + LPC_SRC_FP + "_" + plan_spec_key(spec) + ".so";
}

// directories a module is looked for in (first hit wins) / written to (first writable one)
static std::vector<std::string> module_dirs(const EngineOpts& opt) {
  std::vector<std::string> d;
  if (!opt.module_dir.empty()) d.push_back(opt.module_dir);
  const std::string self = self_path();
  if (!self.empty()) d.push_back(dir_of(self) + "/modules");
  const char* home = std::getenv("HOME");          // a read-only installation still gets its modules: per-user cache
  if (home && *home) d.push_back(std::string(home) + "/.cache/lenslesspicam_amd");
  return d;
}

static std::string find_compiler(const EngineOpts& opt) {
#if defined(LPC_SIMT_EMU)
  (void)opt;
  return "g++";
#else
  if (!opt.compiler.empty()) return opt.compiler;
  std::vector<std::string> cand;
  if (const char* rp = std::getenv("ROCM_PATH")) cand.push_back(std::string(rp) + "/bin/hipcc");
  cand.push_back("/opt/rocm/bin/hipcc");
  for (const std::string& c : cand)
    if (::access(c.c_str(), X_OK) == 0) return c;
  return "";
#endif
}

static std::string shell_quote(const std::string& a) {
  std::string q = "'";
  for (char ch : a) q += (ch == '\'') ? std::string("'\\''") : std::string(1, ch);
  return q + "'";
}

int build_plan_module(const PlanSpec& spec, const EngineOpts& opt, std::string* path_or_error) {
  const std::string self = self_path();
  if (self.empty()) { *path_or_error = "cannot locate the library on disk"; return 1; }
  const std::string lib_dir = dir_of(self);
  const std::string csrc = lib_dir + "/" + LPC_CSRC_REL, inc = lib_dir + "/" + LPC_INCLUDE_REL;
  const std::string src = csrc + "/lpc_module.cpp";
  if (!file_exists(src)) { *path_or_error = "module source not found: " + src; return 1; }
  const std::string cc = find_compiler(opt);
  if (cc.empty()) { *path_or_error = "no hipcc (option compiler=, $ROCM_PATH/bin/hipcc, /opt/rocm/bin/hipcc)"; return 1; }
  std::string out_dir;
  for (const std::string& d : module_dirs(opt))
    if (dir_writable(d)) { out_dir = d; break; }
  if (out_dir.empty()) { *path_or_error = "no writable module directory"; return 1; }
  const std::string out = out_dir + "/" + module_file(spec);
  const std::string tmp = out + ".tmp" + std::to_string((long)::getpid());
  std::string cmd = shell_quote(cc);
#if defined(LPC_SIMT_EMU)
  cmd += " -std=c++17 -O2 -fPIC -shared -DLPC_SIMT_EMU -x c++";
#else
  cmd += " -std=c++17 -O3 --offload-arch=gfx950 -fPIC -shared -x hip";
#endif
  cmd += " -I" + shell_quote(inc) + " -I" + shell_quote(csrc) + " -DLPC_SRC_FP=" + shell_quote(std::string("\"") + LPC_SRC_FP + "\"");
  for (const std::string& d : plan_spec_defines(spec)) cmd += " " + shell_quote(d);
  cmd += " " + shell_quote(src) + " -x none " + shell_quote(self) + " -o " + shell_quote(tmp) + " 2>&1";
  std::string log;
  FILE* p = ::popen(cmd.c_str(), "r");
  if (!p) { *path_or_error = "cannot start the compiler"; return 1; }
  char buf[512];
  while (std::fgets(buf, sizeof buf, p)) { if (log.size() < 4000) log += buf; }
  const int rc = ::pclose(p);
  if (rc != 0 || !file_exists(tmp)) {
    ::unlink(tmp.c_str());
    *path_or_error = "compiling " + module_file(spec) + " failed: " + log;
    return 1;
  }
  if (::rename(tmp.c_str(), out.c_str()) != 0) { ::unlink(tmp.c_str()); *path_or_error = "cannot move the module into place"; return 1; }
  *path_or_error = out;
  return 0;
}

const LpcModule* get_plan_module(const PlanSpec& spec, const EngineOpts& opt, bool allow_compile, std::string* why) {
  static std::mutex mu;
  static std::map<std::string, const LpcModule*> cache;     // modules stay loaded for the life of the process
  static std::map<std::string, std::string> failed;         // ... and a failure is not retried (nor reported twice)
  const std::string file = module_file(spec);
  std::lock_guard<std::mutex> lock(mu);
  auto hit = cache.find(file);
  if (hit != cache.end()) return hit->second;
  auto bad = failed.find(file);
  if (bad != failed.end()) { if (why) *why = bad->second; return nullptr; }
  std::string path;
  for (const std::string& d : module_dirs(opt))
    if (file_exists(d + "/" + file)) { path = d + "/" + file; break; }
  std::string note;
  if (path.empty()) {
    if (!allow_compile) note = "module " + plan_spec_key(spec) + " not built and jit=0";
    else if (build_plan_module(spec, opt, &path) != 0) { note = path; path.clear(); }
  }
  if (!path.empty()) {
    void* h = ::dlopen(path.c_str(), RTLD_NOW | RTLD_LOCAL);
    if (!h) note = std::string("dlopen: ") + ::dlerror();
    else {
      typedef int (*init_fn)(LpcModule*, size_t, const char*);
      init_fn init = (init_fn)::dlsym(h, "lpc_module_init");
      LpcModule* m = new LpcModule();
      if (init && init(m, sizeof(lpc_engine), LPC_SRC_FP) == 0) {
        cache[file] = m;
        return m;
      }
      delete m;
      ::dlclose(h);
      note = "module " + path + " was built from other sources";
    }
  }
  failed[file] = note;
  std::fprintf(stderr, "lenslesspicam_amd: compile-time plans unavailable, using run-time plans (%s)\n", note.c_str());
  if (why) *why = note;
  return nullptr;
}
