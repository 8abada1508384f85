// Key/value configuration files (reference: src/util/config_reader.{h,cpp},
// src/util/string_util.cpp:11-62).  One pair per line, split at the FIRST
// delimiter that follows a non-empty key (runs of the delimiter before the key
// are skipped), both sides trimmed; lines that start with '#' or do not yield
// two pieces are ignored; later keys overwrite earlier ones.
#pragma once
#include <cctype>
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <map>
#include <string>

namespace super_resolution {
namespace util {

inline std::string TrimString(const std::string& s) {
  size_t a = 0, b = s.size();
  while (a < b && std::isspace(static_cast<unsigned char>(s[a]))) ++a;
  while (b > a && std::isspace(static_cast<unsigned char>(s[b - 1]))) --b;
  return s.substr(a, b - a);
}

class ConfigurationFileReader {
 public:
  void ReadFromFile(const std::string& file_path) {
    std::ifstream fin(file_path);
    if (!fin.is_open()) Fatal("Could not open file '" + file_path + "' for reading.");
    std::string line;
    while (std::getline(fin, line)) {
      if (line.rfind("#", 0) == 0) continue;  // comment
      // key = first non-empty piece, value = everything after the delimiter that ends it
      size_t pos = 0;
      std::string key;
      bool have_key = false;
      while (pos <= line.size()) {
        const size_t d = line.find(key_value_delimiter_, pos);
        if (d == std::string::npos) break;
        if (d > pos) { key = line.substr(pos, d - pos); have_key = true; pos = d + 1; break; }
        pos = d + 1;  // empty piece: skip
      }
      if (!have_key) continue;
      const std::string rest = line.substr(pos);
      if (rest.empty()) continue;  // fewer than two pieces
      config_map_[TrimString(key)] = TrimString(rest);
    }
  }
  void SetDelimiter(const char delimiter) { key_value_delimiter_ = delimiter; }
  void SetValue(const std::string& key, const std::string& value) { config_map_[key] = value; }
  bool HasValue(const std::string& key) const { return config_map_.find(key) != config_map_.end(); }
  std::string GetValue(const std::string& key) const {
    const auto it = config_map_.find(key);
    return it != config_map_.end() ? it->second : std::string();
  }
  int GetValueAsInt(const std::string& key) const {
    if (!HasValue(key)) return 0;
    return std::atoi(GetValue(key).c_str());  // 0 when not a number
  }
  std::string GetValueOrDie(const std::string& key) const {
    if (!HasValue(key)) Fatal("The map does not have a value for key '" + key + "'.");
    return GetValue(key);
  }

 private:
  [[noreturn]] static void Fatal(const std::string& message) {  // glog CHECK semantics
    std::fprintf(stderr, "Check failed: %s\n", message.c_str());
    std::abort();
  }
  char key_value_delimiter_ = ' ';
  std::map<std::string, std::string> config_map_;
};

}  // namespace util
}  // namespace super_resolution
