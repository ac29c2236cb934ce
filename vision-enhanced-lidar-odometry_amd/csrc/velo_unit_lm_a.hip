// velo_unit_lm_a.hip -- the translation unit that DEFINES the kernels of the VELO_DEF_LMA family (velo_kernels.h, "translation units"): their
// device code is generated here and nowhere else; velo_hip.hip (the host side of the C-ABI) and the other units see declarations and launch
// through the host stubs this unit exports.  No host logic lives here.
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <stdint.h>

#include "../../include/velo_hip.h"

#define VELO_DEF_LMA 1
#include "velo_kernels.h"
