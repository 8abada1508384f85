// Minimal gflags-style command line for the host tools: --name=value, --name value,
// --flag / --noflag for booleans (the reference uses gflags, src/util/util.cpp InitApp).
#pragma once
#include <cstdio>
#include <cstdlib>
#include <map>
#include <string>

namespace super_resolution {
namespace app {

class Flags {
 public:
  Flags(int argc, char** argv, const char* usage) : usage_(usage) {
    for (int i = 1; i < argc; ++i) {
      std::string a = argv[i];
      if (a.rfind("--", 0) != 0) Die("unexpected argument '" + a + "'");
      a = a.substr(2);
      const auto eq = a.find('=');
      if (eq != std::string::npos) { values_[a.substr(0, eq)] = a.substr(eq + 1); continue; }
      if (i + 1 < argc && std::string(argv[i + 1]).rfind("--", 0) != 0) { values_[a] = argv[++i]; continue; }
      if (a.rfind("no", 0) == 0 && a.size() > 2) values_[a.substr(2)] = "false"; else values_[a] = "true";
    }
    if (values_.count("help")) { std::printf("%s\n", usage_); std::exit(0); }
  }
  std::string Str(const std::string& name, const std::string& def = "") { seen_[name] = true; return values_.count(name) ? values_[name] : def; }
  int Int(const std::string& name, int def) { const std::string s = Str(name); return s.empty() ? def : std::atoi(s.c_str()); }
  double Double(const std::string& name, double def) { const std::string s = Str(name); return s.empty() ? def : std::atof(s.c_str()); }
  bool Bool(const std::string& name, bool def) { const std::string s = Str(name); return s.empty() ? def : (s == "true" || s == "1"); }
  void Require(const std::string& name) { if (!values_.count(name) || values_[name].empty()) Die("Required argument '" + name + "' is missing"); }
  void RejectUnknown() {
    for (const auto& kv : values_) if (!seen_.count(kv.first)) Die("unknown flag --" + kv.first);
  }

 private:
  [[noreturn]] void Die(const std::string& m) { std::fprintf(stderr, "%s\n%s\n", m.c_str(), usage_); std::exit(2); }
  const char* usage_;
  std::map<std::string, std::string> values_;
  std::map<std::string, bool> seen_;
};

}  // namespace app
}  // namespace super_resolution
