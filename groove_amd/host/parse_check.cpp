// parse_check.cpp — the parser half of the host layer (json5.hpp, project.cpp: projects, Welsh patches, WAV headers) as a
// stand-alone program for SANITIZER and fuzz runs on the CPU (`make -C groove_amd/host asan`, tests/test_host_sanitizers.py).
// These files read untrusted input; the reference gets memory safety from Rust (settings/src/songs.rs:84-89: json5::from_str)
// and warns-and-continues on bad content (songs.rs:136-139, 152-156).  The promise checked here is the C ABI's
// (include/groove_hip.h): an error string, never a crash.  Needs no device library (GROOVE_HOST_PARSER_ONLY).
//
//   parse_check project <assets_root|-> file...      one line per file: "ok <n devices> <n notes> <n warnings>" | "error: <message>"
//   parse_check patch file...                         "ok" | "error: <message>"
//   parse_check wav file...                           "ok <frames> <rate>" | "error: <message>"
#include "project.hpp"
#include <cstdio>
#include <cstring>
#include <fstream>
#include <sstream>

extern "C" int gh_welsh_params_from_patch_json(const char* text, groove_welsh_params* out, char* err, size_t err_len);

static std::string slurp(const char* path) {
  std::ifstream f(path, std::ios::binary);
  std::ostringstream ss;
  ss << f.rdbuf();
  return ss.str();
}

int main(int argc, char** argv) {
  if (argc < 2) { std::fprintf(stderr, "usage: parse_check project <assets|-> file... | patch file... | wav file...\n"); return 2; }
  const std::string mode = argv[1];
  if (mode == "project") {
    if (argc < 3) return 2;
    const std::string assets = std::strcmp(argv[2], "-") ? argv[2] : "";
    for (int i = 3; i < argc; ++i) {
      try {
        const groove_host::ProjectDesc p = groove_host::parse_project(slurp(argv[i]), assets);
        const std::string d = groove_host::describe(p);
        std::printf("ok %zu %zu %zu %zu\n", p.devices.size(), p.notes.size(), p.warnings.size(), d.size());
      } catch (const std::exception& e) {
        std::printf("error: %s\n", e.what());
      }
    }
  } else if (mode == "patch") {
    for (int i = 2; i < argc; ++i) {
      groove_welsh_params out;
      char err[256] = {0};
      const std::string text = slurp(argv[i]);
      if (gh_welsh_params_from_patch_json(text.c_str(), &out, err, sizeof(err)) == 0) std::printf("ok\n");
      else std::printf("error: %s\n", err);
    }
  } else if (mode == "wav") {
    for (int i = 2; i < argc; ++i) {
      std::vector<float> pcm;
      uint32_t rate = 0;
      std::string err;
      if (groove_host::read_wav_mono(argv[i], pcm, &rate, &err)) std::printf("ok %zu %u\n", pcm.size(), rate);
      else std::printf("error: %s\n", err.c_str());
    }
  } else return 2;
  return 0;
}
