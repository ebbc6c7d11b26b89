/* placeholder translation unit: filter + fuse restatement lands here (SD.cpp:3006-3495) */
#include "hcmvs_oracle.h"
