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
#include <sys/time.h>
#include <sys/types.h>
#include <unistd.h>

#include <dirent.h>
#include <pthread.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <map>
#include <memory>
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
static bool dir_exists(const std::string& p) {
  struct stat st;
  return ::stat(p.c_str(), &st) == 0 && S_ISDIR(st.st_mode);
}
// mkdir -p, new components 0700 (a per-user cache must not be writable by others: modules are dlopen()ed from it)
static bool make_dirs(const std::string& d) {
  if (d.empty() || dir_exists(d)) return !d.empty();
  const size_t k = d.rfind('/');
  if (k != std::string::npos && k > 0 && !make_dirs(d.substr(0, k))) return false;
  return ::mkdir(d.c_str(), 0700) == 0 || errno == EEXIST;
}
static bool dir_writable(const std::string& d) {
  if (!make_dirs(d)) return false;
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

// ---- the sources a module is compiled from must be the ones this library was built from ------------------------------
// build.py / build_emu.sh hand the library the CRC-32 of its sources (LPC_SRC_CRC: every .h / .cpp / .inc of csrc/ in name
// order, name then content, then include/lpc.h); the JIT recomputes it over the files it is about to compile and refuses
// when they differ -- a module from edited sources would carry this library's fingerprint and pass lpc_module_init().
static uint32_t crc32_update(uint32_t crc, const unsigned char* p, size_t n) {
  static uint32_t table[256];
  static std::once_flag once;
  std::call_once(once, [] {
    for (uint32_t i = 0; i < 256; ++i) {
      uint32_t c = i;
      for (int k = 0; k < 8; ++k) c = (c & 1) ? 0xEDB88320u ^ (c >> 1) : c >> 1;
      table[i] = c;
    }
  });
  crc = ~crc;
  for (size_t i = 0; i < n; ++i) crc = table[(crc ^ p[i]) & 0xff] ^ (crc >> 8);
  return ~crc;
}
static bool crc_file(const std::string& path, const std::string& name, uint32_t* crc) {
  FILE* f = std::fopen(path.c_str(), "rb");
  if (!f) return false;
  *crc = crc32_update(*crc, (const unsigned char*)name.data(), name.size());
  unsigned char buf[1 << 16];
  size_t n;
  while ((n = std::fread(buf, 1, sizeof buf, f)) > 0) *crc = crc32_update(*crc, buf, n);
  std::fclose(f);
  return true;
}
static bool is_source_name(const std::string& n) {
  auto ends = [&](const char* suf) { const size_t k = std::strlen(suf); return n.size() > k && n.compare(n.size() - k, k, suf) == 0; };
  return ends(".h") || ends(".cpp") || ends(".inc");
}
static bool sources_crc(const std::string& csrc, const std::string& inc, uint32_t* out) {
  std::vector<std::string> names;
  DIR* d = ::opendir(csrc.c_str());
  if (!d) return false;
  while (struct dirent* de = ::readdir(d))
    if (is_source_name(de->d_name)) names.push_back(de->d_name);
  ::closedir(d);
  std::sort(names.begin(), names.end());
  uint32_t crc = 0;
  for (const std::string& n : names)
    if (!crc_file(csrc + "/" + n, n, &crc)) return false;
  if (!crc_file(inc + "/lpc.h", "lpc.h", &crc)) return false;
  *out = crc;
  return true;
}

// ---- the module directory is a cache: bounded (option module_max), least recently used first -------------------------
static void prune_module_dir(const std::string& dir, int keep, const std::string& spare) {
  if (keep <= 0) return;
  struct Ent { std::string path; time_t used; bool ours; };
  std::vector<Ent> ents;
  DIR* d = ::opendir(dir.c_str());
  if (!d) return;
  const std::string mine = std::string("lpcmod_") + kBackendTag + "_" + LPC_SRC_FP + "_";
  while (struct dirent* de = ::readdir(d)) {
    const std::string n = de->d_name;
    if (n.compare(0, 7, "lpcmod_") != 0 || n.size() < 4 || n.compare(n.size() - 3, 3, ".so") != 0) continue;
    struct stat st;
    const std::string path = dir + "/" + n;
    if (::stat(path.c_str(), &st) != 0 || path == spare) continue;
    ents.push_back({path, std::max(st.st_atime, st.st_mtime), n.compare(0, mine.size(), mine) == 0});
  }
  ::closedir(d);
  if ((int)ents.size() + 1 <= keep) return;
  // modules of other sources can never be loaded by this library: they go first, then the longest unused
  std::sort(ents.begin(), ents.end(), [](const Ent& a, const Ent& b) { return a.ours != b.ours ? !a.ours : a.used < b.used; });
  for (size_t i = 0; i + (size_t)keep < ents.size() + 1; ++i) ::unlink(ents[i].path.c_str());
}

int build_plan_module(const PlanSpec& spec, const EngineOpts& opt, std::string* path_or_error) {
  // "only if not on disk" (include/lpc.h): a module some earlier call or process built is taken as it is
  for (const std::string& d : module_dirs(opt))
    if (file_exists(d + "/" + module_file(spec))) { *path_or_error = d + "/" + module_file(spec); return 0; }
  const std::string self = self_path();
  if (self.empty()) { *path_or_error = "cannot locate the library on disk"; return 1; }
  const std::string lib_dir = dir_of(self);
  const std::string csrc = lib_dir + "/" + LPC_CSRC_REL, inc = lib_dir + "/" + LPC_INCLUDE_REL;
  const std::string src = csrc + "/lpc_module.cpp";
  if (!file_exists(src)) { *path_or_error = "module source not found: " + src; return 1; }
#if defined(LPC_SRC_CRC)
  {
    uint32_t crc = 0;
    if (!sources_crc(csrc, inc, &crc)) { *path_or_error = "cannot read the sources under " + csrc; return 1; }
    if (crc != (uint32_t)LPC_SRC_CRC) {
      *path_or_error = "the sources under " + csrc + " are not the ones this library was built from (rebuild the library)";
      return 1;
    }
  }
#endif
  const std::string cc = find_compiler(opt);
  // (a bare name -- compiler=hipcc -- is resolved through PATH by the shell below: only a path is checked here)
  if (cc.empty() || (cc.find('/') != std::string::npos && ::access(cc.c_str(), X_OK) != 0)) {
#if !defined(LPC_SIMT_EMU)
    *path_or_error = "no hipcc (" + (cc.empty() ? std::string("option compiler=, $ROCM_PATH/bin/hipcc, /opt/rocm/bin/hipcc") : cc + " is not executable") + ")";
    return 1;
#endif
  }
  std::string out_dir;
  for (const std::string& d : module_dirs(opt))
    if (dir_writable(d)) { out_dir = d; break; }
  if (out_dir.empty()) { *path_or_error = "no writable module directory"; return 1; }
  const std::string out = out_dir + "/" + module_file(spec);
  // private temporary, unique per CALL: ranks of one node (processes) and the threads of one process (build.py's pool,
  // two configurations that map to one key) may all build the same module at once; whoever renames last wins, all
  // copies are identical
  static std::atomic<unsigned long> serial{0};
  const std::string tmp = out + ".tmp" + std::to_string((long)::getpid()) + "." +
                          std::to_string((unsigned long)(uintptr_t)::pthread_self()) + "." + std::to_string(serial++);
  std::string cmd = shell_quote(cc);
#if defined(LPC_SIMT_EMU)
  cmd += " -std=c++17 -O2 -fPIC -shared -DLPC_SIMT_EMU -x c++";
#else
  cmd += " -std=c++17 -O3 --offload-arch=gfx950 -fPIC -shared -x hip";
#endif
  cmd += " -I" + shell_quote(inc) + " -I" + shell_quote(csrc) + " -DLPC_SRC_FP=" + shell_quote(std::string("\"") + LPC_SRC_FP + "\"");
#if defined(LPC_MODULE_EXTRA_DEFS)
  {   // build-time flavour flags of the library (LPC_EXTRA_DEFS, part of the fingerprint) apply to its modules as well
    const std::string extra = LPC_MODULE_EXTRA_DEFS;
    size_t i = 0;
    while (i < extra.size()) {
      size_t j = extra.find_first_of(" \t", i);
      if (j == std::string::npos) j = extra.size();
      if (j > i) cmd += " " + shell_quote(extra.substr(i, j - i));
      i = j + 1;
    }
  }
#endif
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
  prune_module_dir(out_dir, opt.module_max, out);
  if (::rename(tmp.c_str(), out.c_str()) != 0) { ::unlink(tmp.c_str()); *path_or_error = "cannot move the module into place"; return 1; }
  *path_or_error = out;
  return 0;
}

// ---- the process cache ------------------------------------------------------------------------------------------------
// One slot per module file.  The table lock is held only to find the slot; finding / compiling / loading happens under
// the slot's own lock, so a 3-second compile of one shape never blocks lpc_create for another.  Handles count their
// module (release_plan_module from lpc_destroy); an unreferenced module stays loaded until more than `module_loaded_max`
// are, then the longest unused one is dlclose()d.
namespace {
struct Slot {
  std::mutex mu;
  const LpcModule* mod = nullptr;
  void* dl = nullptr;
  std::atomic<int> refs{0};
  unsigned long last_use = 0;
  bool failed = false, compile_tried = false;
  std::string note, fail_env;          // fail_env: compiler + module directory of the handle that failed
  std::chrono::steady_clock::time_point failed_at;
};
std::mutex g_table_mu;
std::map<std::string, std::shared_ptr<Slot>> g_slots;
std::map<const LpcModule*, std::shared_ptr<Slot>> g_by_mod;
unsigned long g_use_clock = 0;

void unload_surplus(int keep_loaded) {     // g_table_mu held
  if (keep_loaded < 1) keep_loaded = 1;
  for (;;) {
    int loaded = 0;
    std::shared_ptr<Slot> victim;
    for (auto& kv : g_slots) {
      Slot& s = *kv.second;
      if (!s.mod) continue;
      ++loaded;
      if (s.refs == 0 && (!victim || s.last_use < victim->last_use)) victim = kv.second;
    }
    if (loaded <= keep_loaded || !victim) return;
    std::unique_lock<std::mutex> lk(victim->mu, std::try_to_lock);
    if (!lk.owns_lock() || victim->refs != 0) return;
    g_by_mod.erase(victim->mod);
    delete victim->mod;
    victim->mod = nullptr;
    if (victim->dl) ::dlclose(victim->dl);
    victim->dl = nullptr;
  }
}
// a shared object cut short (written by a process that died, a full disk): not an ELF header, or its section header
// table lies beyond the end of the file
static bool elf_truncated(const std::string& path) {
  FILE* f = std::fopen(path.c_str(), "rb");
  if (!f) return false;                       // unreadable here is not "bad"
  unsigned char h[64];
  const size_t got = std::fread(h, 1, sizeof(h), f);
  std::fseek(f, 0, SEEK_END);
  const long size = std::ftell(f);
  std::fclose(f);
  if (got < sizeof(h) || std::memcmp(h, "\177ELF", 4) != 0 || h[4] != 2) return true;     // ELFCLASS64
  unsigned long long shoff = 0;
  for (int i = 7; i >= 0; --i) shoff = (shoff << 8) | h[0x28 + i];
  const unsigned shentsize = h[0x3A] | (h[0x3B] << 8), shnum = h[0x3C] | (h[0x3D] << 8);
  return shoff == 0 || (long long)(shoff + (unsigned long long)shentsize * shnum) > (long long)size;
}
}  // namespace

const LpcModule* get_plan_module(const PlanSpec& spec, const EngineOpts& opt, bool allow_compile, std::string* why) {
  const std::string file = module_file(spec);
  std::shared_ptr<Slot> slot;
  {
    std::lock_guard<std::mutex> lock(g_table_mu);
    std::shared_ptr<Slot>& s = g_slots[file];
    if (!s) s = std::make_shared<Slot>();
    slot = s;
  }
  const LpcModule* result = nullptr;
  {
    std::lock_guard<std::mutex> lock(slot->mu);
    Slot& s = *slot;
    if (!s.mod && s.failed) {
      // a failure is remembered (and reported once) -- but not for ever: a handle that may compile retries what an
      // earlier jit=0 handle could not, one that names another compiler or module directory retries at once, and a
      // failed compile / load is retried after a minute (full disk, a module half-written by a dying neighbour, ...)
      const bool retry = (allow_compile && !s.compile_tried) || s.fail_env != opt.compiler + "|" + opt.module_dir ||
                         std::chrono::steady_clock::now() - s.failed_at > std::chrono::seconds(60);
      if (!retry) { if (why) *why = s.note; return nullptr; }
      s.failed = false;
    }
    if (!s.mod) {
      std::string path, note;
      for (const std::string& d : module_dirs(opt))
        if (file_exists(d + "/" + file)) { path = d + "/" + file; break; }
      if (path.empty()) {
        if (!allow_compile) note = "module " + plan_spec_key(spec) + " not built and jit=0";
        else {
          s.compile_tried = true;
          if (build_plan_module(spec, opt, &path) != 0) { note = path; path.clear(); }
        }
      }
      // a file that is on disk but does not load (truncated by a dying neighbour, built from other sources under this
      // fingerprint, ...) must not be found again by every retry: it is removed and -- when this handle may compile --
      // rebuilt once before the failure is recorded
      for (int attempt = 0; attempt < 2 && !path.empty() && !s.mod; ++attempt) {
        // ... only a file that is PROVABLY bad, though: a truncated ELF image, or one whose lpc_module_init is missing or
        // refuses this library.  A dlopen() that fails for a reason of this node or moment (a ROCm runtime library that
        // does not resolve here, no memory) leaves a module directory shared between nodes or ranks untouched.
        bool bad_file = false;
        void* h = ::dlopen(path.c_str(), RTLD_NOW | RTLD_LOCAL);
        if (!h) {
          note = std::string("dlopen: ") + ::dlerror();
          bad_file = elf_truncated(path);
        } else {
          typedef int (*init_fn)(LpcModule*, size_t, const char*);
          init_fn init = (init_fn)::dlsym(h, "lpc_module_init");
          LpcModule* m = new LpcModule();
          if (init && init(m, sizeof(lpc_engine), LPC_SRC_FP) == 0) {
            s.mod = m;
            s.dl = h;
            ::utimes(path.c_str(), nullptr);        // least-recently-USED order for prune_module_dir
          } else {
            delete m;
            ::dlclose(h);
            note = "module " + path + " was built from other sources";
            bad_file = true;
          }
        }
        if (!s.mod) {
          const bool removed = bad_file && ::access(dir_of(path).c_str(), W_OK) == 0 && ::unlink(path.c_str()) == 0;
          path.clear();
          if (removed && allow_compile && attempt == 0) {
            s.compile_tried = true;
            std::string built;
            if (build_plan_module(spec, opt, &built) == 0) path = built;
            else note += "; rebuilding it failed: " + built;
          }
        }
      }
      if (!s.mod) {
        s.failed = true;
        s.fail_env = opt.compiler + "|" + opt.module_dir;
        s.failed_at = std::chrono::steady_clock::now();
        s.note = note;
        std::fprintf(stderr, "lenslesspicam_amd: compile-time plans unavailable, using run-time plans (%s)\n", note.c_str());
        if (why) *why = note;
        return nullptr;
      }
    }
    ++s.refs;
    result = s.mod;
  }
  {
    std::lock_guard<std::mutex> lock(g_table_mu);
    slot->last_use = ++g_use_clock;
    g_by_mod[result] = slot;
    unload_surplus(opt.module_loaded_max);
  }
  return result;
}

void release_plan_module(const LpcModule* mod) {
  if (!mod) return;
  std::lock_guard<std::mutex> lock(g_table_mu);
  auto it = g_by_mod.find(mod);
  if (it == g_by_mod.end()) return;
  if (it->second->refs.load() > 0) --it->second->refs;     // (no slot lock: a compile in that slot must not block a destroy)
}
