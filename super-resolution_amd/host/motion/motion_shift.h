// MotionShift / MotionShiftSequence (src/motion/motion_shift.h:14-57,
// motion_shift.cpp:17-53): per-frame (dx, dy), loadable from a text file of
// "dx dy" lines.
#pragma once
#include <fstream>
#include <string>
#include <vector>

#include "util/srmap_host.h"

namespace super_resolution {

struct MotionShift {
  MotionShift(const double dx, const double dy) : dx(dx), dy(dy) {}
  double dx, dy;
};

class MotionShiftSequence {
 public:
  MotionShiftSequence() {}
  explicit MotionShiftSequence(const std::vector<MotionShift>& shifts) : shifts_(shifts) {}
  void SetMotionSequence(const std::vector<MotionShift>& shifts) { shifts_ = shifts; }
  void LoadSequenceFromFile(const std::string& path) {
    std::ifstream fin(path);
    if (!fin.is_open()) srmap_host::Fail(("Could not open file " + path).c_str());
    shifts_.clear();
    double dx, dy;
    while (fin >> dx >> dy) shifts_.push_back(MotionShift(dx, dy));
  }
  int GetNumMotionShifts() const { return static_cast<int>(shifts_.size()); }
  const MotionShift& GetMotionShift(const int index) const {
    if (index < 0 || index >= GetNumMotionShifts()) srmap_host::Fail("motion shift index out of range");
    return shifts_[index];
  }
  const MotionShift& operator[](const int index) const { return GetMotionShift(index); }
  std::vector<double> Flat() const {
    std::vector<double> f;
    for (const auto& s : shifts_) { f.push_back(s.dx); f.push_back(s.dy); }
    return f;
  }

 private:
  std::vector<MotionShift> shifts_;
};

}  // namespace super_resolution
