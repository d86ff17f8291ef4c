// <pcl/point_types.h> of the stand-in include tree: forwards to compat/pcl_lite.h (the slice of PCL the
// reference's class surface and drivers use).  Only for building WITHOUT PCL; with PCL installed put
// its include directory first and define SICP_HAVE_REAL_DEPS.
#ifndef SICP_COMPAT_INCLUDE_PCL_POINT_TYPES_H_
#define SICP_COMPAT_INCLUDE_PCL_POINT_TYPES_H_
#include "../../pcl_lite.h"
#endif
