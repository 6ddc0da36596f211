// kernels_mobi.hpp -- MOBI biogeochemistry source terms, one ocean column per thread.
// (placeholder: the column kernel is added in a later commit of this round)
#ifndef UVIC_KERNELS_MOBI_HPP
#define UVIC_KERNELS_MOBI_HPP
#include <string>
#include "kenv.hpp"
#include "uvic_ctx.h"

struct mobi_host { int unused; };
struct mobi_dev { int unused; };

namespace uvic {
UVIC_DEV void mobi_column_kernel(const uvic_ctx &, const mobi_dev &, int, int) {}
}
#if defined(__HIPCC__)
static inline int mobi_bind(int, int, int, const mobi_host *, mobi_dev *, void **, hipStream_t, bool *have, std::string &err) {
  *have = false;
  err = "uvic_gpu_set_mobi: MOBI source terms are not part of this build";
  return 3;
}
#endif
#endif
