// <pcl/console/parse.h>: the one overload family the reference's drivers call
// (exec/test_icp.cc:24-31, exec/kitti_eval.cc): the value following a flag.
#ifndef SICP_COMPAT_INCLUDE_PCL_CONSOLE_PARSE_H_
#define SICP_COMPAT_INCLUDE_PCL_CONSOLE_PARSE_H_
#include <cstdlib>
#include <cstring>
#include <string>

namespace pcl {
namespace console {
// index of the flag's value in argv, or -1 (PCL returns the index too; the drivers test it as a bool)
inline int find_value(int argc, const char* const* argv, const char* flag) {
  for (int i = 1; i + 1 < argc; ++i)
    if (std::strcmp(argv[i], flag) == 0) return i + 1;
  return -1;
}
inline int parse_argument(int argc, const char* const* argv, const char* flag, std::string& val) {
  const int i = find_value(argc, argv, flag);
  if (i > 0) val = argv[i];
  return i > 0 ? i : 0;
}
inline int parse_argument(int argc, const char* const* argv, const char* flag, int& val) {
  const int i = find_value(argc, argv, flag);
  if (i > 0) val = std::atoi(argv[i]);
  return i > 0 ? i : 0;
}
inline int parse_argument(int argc, const char* const* argv, const char* flag, double& val) {
  const int i = find_value(argc, argv, flag);
  if (i > 0) val = std::atof(argv[i]);
  return i > 0 ? i : 0;
}
inline bool find_switch(int argc, const char* const* argv, const char* flag) {
  for (int i = 1; i < argc; ++i)
    if (std::strcmp(argv[i], flag) == 0) return true;
  return false;
}
}  // namespace console
}  // namespace pcl
#endif
