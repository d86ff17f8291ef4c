// <pcl/io/pcd_io.h>: pcl::io::loadPCDFile / savePCDFileASCII for `x y z [label]` clouds live in
// compat/pcl_lite.h; PCL_ERROR is PCL's printf-style console macro.
#ifndef SICP_COMPAT_INCLUDE_PCL_IO_PCD_IO_H_
#define SICP_COMPAT_INCLUDE_PCL_IO_PCD_IO_H_
#include <cstdio>

#include "pcl/point_types.h"
#ifndef PCL_ERROR
#define PCL_ERROR(...) std::fprintf(stderr, __VA_ARGS__)
#endif
#endif
