// gemm_wn_tiled.hip — large-M tiled MFMA GEMM for packed weights (placeholder until the tiled kernel lands;
// the dispatcher falls through to the streaming MFMA kernel when this planner declines).
#include "gl_common.h"

namespace gl {
bool plan_gemm_wn_tiled(const gemlite_hip_forward_args&, WnParams&, LaunchPlan&) { return false; }
}  // namespace gl
