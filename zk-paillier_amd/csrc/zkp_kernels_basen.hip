// zkp_kernels_basen.hip — the shared-key Paillier kernels in base-n form (kernels_basen.hpp) as a translation unit of their own:
// they are compiled under their own machine-scheduler strategy (__graft_entry__.build) and their device assembly is kept apart for
// tests/test_isa_quality.py.  zkp_api.hip declares the same instantiations `extern template` (ZKP_SPLIT_TU) and launches them.
#define ZKP_TEMPLATE_KERNELS_ONLY          /* (bigint29.hpp) */
#include "../../include/zkp_hip.h"
#include "kernels_basen.hpp"

namespace zkp {
template __global__ void k_enc_basen<2>(EncArgs, const uint32_t*, uint32_t*, uint32_t*);
template __global__ void k_enc_basen<4>(EncArgs, const uint32_t*, uint32_t*, uint32_t*);
template __global__ void k_enc_basen_keys<2>(EncArgs, const uint32_t*, const uint32_t*, uint32_t*, uint32_t*);
template __global__ void k_enc_basen_keys<4>(EncArgs, const uint32_t*, const uint32_t*, uint32_t*, uint32_t*);
template __global__ void k_basen_finish<2>(EncArgs, const uint32_t*, const uint32_t*, int, const uint32_t*, const uint32_t*);
template __global__ void k_basen_finish<4>(EncArgs, const uint32_t*, const uint32_t*, int, const uint32_t*, const uint32_t*);
template __global__ void k_setup_basen<2>(const uint32_t*, uint32_t*, uint64_t, uint32_t*);
template __global__ void k_setup_basen<4>(const uint32_t*, uint32_t*, uint64_t, uint32_t*);
template __global__ void k_expected<4>(EncArgs, uint32_t*, const uint32_t*);
template __global__ void k_expected<8>(EncArgs, uint32_t*, const uint32_t*);
template __global__ void k_diag_basen<2>(const uint32_t*, int, const uint32_t*, const uint32_t*, const uint32_t*, const uint32_t*, uint32_t*, uint32_t*);
template __global__ void k_diag_basen<4>(const uint32_t*, int, const uint32_t*, const uint32_t*, const uint32_t*, const uint32_t*, uint32_t*, uint32_t*);
}  // namespace zkp
