// <pcl/features/fpfh.h> of the stand-in include tree: the types exec/bootstrap.h names live in
// compat/pcl_bootstrap_standins.h (declared so the reference's drivers compile unchanged; never called).
#include "../../../pcl_bootstrap_standins.h"
