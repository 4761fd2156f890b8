// groove-cli-hip — offline renderer, the GPU-path counterpart of the reference's `groove-cli`
// (/root/reference/src/bin/groove-cli.rs:24-53 args, :56-158 main; gated at the reference commit).
//
//   groove-cli-hip [--wav] [--assets DIR] [--device N] [--synthetic-kit] [--quiet] [--perf] [--debug] FILE...
//
// For each project file: SongSettings::new_from_project_file → instantiate → update_sample_rate(44100)
// → run_performance(buffer) → (with --wav) send_performance_to_file(<input with .json5/.json → .wav>).
#include "project.hpp"
#include <chrono>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

using namespace groove_host;

static std::string wav_name(const std::string& in) {
  // groove-cli.rs:144-148 rewrites ".json5" → ".wav"; the shipped demo is ".json" (SURVEY F6), so accept both
  for (const char* ext : {".json5", ".json"}) {
    const size_t n = std::strlen(ext);
    if (in.size() > n && in.compare(in.size() - n, n, ext) == 0) return in.substr(0, in.size() - n) + ".wav";
  }
  return in + ".wav";
}

int main(int argc, char** argv) {
  bool wav = false, quiet = false, perf = false, debug = false, synthetic = false;
  int device = 0;
  std::string assets = "assets";
  std::vector<std::string> inputs;
  for (int i = 1; i < argc; ++i) {
    const std::string a = argv[i];
    if (a == "--wav" || a == "-w") wav = true;
    else if (a == "--quiet" || a == "-q") quiet = true;
    else if (a == "--perf" || a == "-p") perf = true;
    else if (a == "--debug" || a == "-d") debug = true;
    else if (a == "--synthetic-kit") synthetic = true;
    else if (a == "--assets" && i + 1 < argc) assets = argv[++i];
    else if (a == "--device" && i + 1 < argc) device = std::atoi(argv[++i]);
    else if (a == "--version" || a == "-v") { std::printf("groove-cli-hip 0.1 (MI355X render path)\n"); return 0; }
    else if (a == "--help" || a == "-h") {
      std::printf("usage: groove-cli-hip [--wav] [--assets DIR] [--device N] [--synthetic-kit] [--quiet] [--perf] [--debug] FILE...\n");
      return 0;
    } else inputs.push_back(a);
  }
  if (inputs.empty()) { std::fprintf(stderr, "no input files\n"); return 2; }
  for (const std::string& in : inputs) {
    ProjectDesc desc;
    try { desc = parse_project_file(in, assets); }
    catch (const std::exception& e) { std::fprintf(stderr, "%s: %s\n", in.c_str(), e.what()); return 1; }
    for (const std::string& w : desc.warnings) std::fprintf(stderr, "Warning: %s\n", w.c_str());
    if (debug) std::printf("%s\n", describe(desc).c_str());
    Orchestrator o(device, GROOVE_DEFAULT_SAMPLE_RATE, desc.bpm);
    if (!o.ctx()) { std::fprintf(stderr, "no HIP device: %s (this renderer has no CPU path)\n", o.last_error().c_str()); return 1; }
    if (instantiate(o, desc, assets, synthetic)) { std::fprintf(stderr, "%s: %s\n", in.c_str(), o.last_error().c_str()); return 1; }
    if (!quiet) std::printf("Performing to queue %s\n", in.c_str());
    const auto t0 = std::chrono::steady_clock::now();
    Performance perf_out;
    if (o.run_performance(GROOVE_BLOCK_FRAMES, perf_out)) { std::fprintf(stderr, "%s: %s\n", in.c_str(), o.last_error().c_str()); return 1; }
    const double secs = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    const size_t frames = perf_out.worker.size();
    if (perf || !quiet) {
      // the same figures groove-cli --perf prints (groove-cli.rs:123-139)
      std::printf("Orchestrator performance:\n Elapsed    : %.6f s\n Frames     : %zu\n Samples per msec (goal > %.1f): %.1f\n usec per sample (goal < %.2f): %.4f\n x real time: %.1f\n",
                  secs, frames, perf_out.sample_rate / 1000.0, frames / (secs * 1000.0), 1e6 / perf_out.sample_rate,
                  secs * 1e6 / (frames ? frames : 1), frames / (double)perf_out.sample_rate / secs);
    }
    if (wav) {
      const std::string out = wav_name(in);
      if (o.send_performance_to_file(perf_out, out)) { std::fprintf(stderr, "%s\n", o.last_error().c_str()); return 1; }
      if (!quiet) std::printf("Rendered %s (%zu frames)\n", out.c_str(), frames);
    }
  }
  return 0;
}
