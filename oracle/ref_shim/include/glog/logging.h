// Minimal glog stand-in: CHECK* abort with the streamed message, LOG prints
// (TEST INFRASTRUCTURE ONLY, see dgemu.h).  The reference's third_party/glog is an
// empty submodule.
#pragma once
#include <cstdlib>
#include <iostream>
#include <sstream>

namespace dgemu_glog {
class Sink {
 public:
  Sink(bool fatal, const char* file, int line, const char* what) : fatal_(fatal) {
    s_ << file << ":" << line << ": " << what << " ";
  }
  ~Sink() {
    std::cerr << s_.str() << std::endl;
    if (fatal_) abort();
  }
  template <typename T>
  Sink& operator<<(const T& v) {
    s_ << v;
    return *this;
  }
  Sink& operator<<(std::ostream& (*m)(std::ostream&)) {
    s_ << m;
    return *this;
  }

 private:
  bool fatal_;
  std::ostringstream s_;
};
struct Voidify {
  void operator&(const Sink&) {}
};
}  // namespace dgemu_glog

#define DGEMU_CHECK_IMPL(cond, text) \
  (cond) ? (void)0 : dgemu_glog::Voidify() & dgemu_glog::Sink(true, __FILE__, __LINE__, "Check failed: " text)
#define CHECK(c) DGEMU_CHECK_IMPL((c), #c)
#define CHECK_EQ(a, b) DGEMU_CHECK_IMPL((a) == (b), #a " == " #b)
#define CHECK_NE(a, b) DGEMU_CHECK_IMPL((a) != (b), #a " != " #b)
#define CHECK_LE(a, b) DGEMU_CHECK_IMPL((a) <= (b), #a " <= " #b)
#define CHECK_LT(a, b) DGEMU_CHECK_IMPL((a) < (b), #a " < " #b)
#define CHECK_GE(a, b) DGEMU_CHECK_IMPL((a) >= (b), #a " >= " #b)
#define CHECK_GT(a, b) DGEMU_CHECK_IMPL((a) > (b), #a " > " #b)
#define DCHECK(c) CHECK(c)
#define DCHECK_EQ(a, b) CHECK_EQ(a, b)
#define LOG(sev) dgemu_glog::Sink(false, __FILE__, __LINE__, #sev)
#define VLOG(n) dgemu_glog::Sink(false, __FILE__, __LINE__, "V")
