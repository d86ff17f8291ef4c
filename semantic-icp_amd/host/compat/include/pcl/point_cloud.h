// forwards to compat/pcl_lite.h (see pcl/point_types.h in this tree)
#include "pcl/point_types.h"
