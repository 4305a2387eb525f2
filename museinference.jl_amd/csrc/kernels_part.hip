// kernels_part.hip -- one group of the solver / loop kernel instantiations (kernels.hpp: MUSE_PART_n); compiled once per group,
// side by side with the others (build.py).
#include "kernels.hpp"
#ifndef MUSE_PART
#error "compile with -DMUSE_PART=n"
#endif
namespace muse {
#define MUSE_CAT_(a, b) a##b
#define MUSE_CAT(a, b) MUSE_CAT_(a, b)
MUSE_CAT(MUSE_PART_, MUSE_PART)()
}  // namespace muse
