// Uniform-grid exact 1-NN engine.  Not built yet: the entry
// points exist so that the ABI is stable; PCCM_ENGINE_GRID reports PCCM_E_ARG until it lands.
#include "pccm_internal.h"

namespace pccm {

int nn_grid(pccm_ctx *, int, const Cloud &, const Cloud &, bool, NNResult &)
{
    return fail(PCCM_E_ARG, "PCCM_ENGINE_GRID is not available in this build");
}

void grid_release(pccm_ctx *) {}

}  // namespace pccm
