// zkp_kernels_basen.hip — the shared-key Paillier kernels in base-n form (kernels_basen.hpp) as a translation unit of their own:
// they are compiled under their own machine-scheduler strategy (__graft_entry__.build) and their device assembly is kept apart for
// tests/test_isa_quality.py.  zkp_api.hip declares the same instantiations `extern template` (ZKP_SPLIT_TU) and launches them.
#define ZKP_TEMPLATE_KERNELS_ONLY          /* (bigint29.hpp) */
#include "../../include/zkp_hip.h"
#include "kernels_basen.hpp"

namespace zkp {
constexpr int BN_GA = 72 / W, BN_GB = 144 / W;      // lanes per n-sized integer (2 / 4 at 36 limbs per lane)
template __global__ void k_enc_basen<BN_GA>(EncArgs, const uint32_t*, uint32_t*, uint32_t*);
template __global__ void k_enc_basen<BN_GB>(EncArgs, const uint32_t*, uint32_t*, uint32_t*);
template __global__ void k_enc_basen_keys<BN_GA>(EncArgs, const uint32_t*, const uint32_t*, uint32_t*, uint32_t*);
template __global__ void k_enc_basen_keys<BN_GB>(EncArgs, const uint32_t*, const uint32_t*, uint32_t*, uint32_t*);
template __global__ void k_basen_finish<BN_GA>(EncArgs, const uint32_t*, const uint32_t*, int, const uint32_t*, const uint32_t*, const uint32_t*);
template __global__ void k_basen_finish<BN_GB>(EncArgs, const uint32_t*, const uint32_t*, int, const uint32_t*, const uint32_t*, const uint32_t*);
template __global__ void k_setup_basen<BN_GA>(const uint32_t*, uint32_t*, uint64_t, uint32_t*, const uint32_t*, uint32_t*);
template __global__ void k_setup_basen<BN_GB>(const uint32_t*, uint32_t*, uint64_t, uint32_t*, const uint32_t*, uint32_t*);
template __global__ void k_expected<2 * BN_GA>(EncArgs, uint32_t*, const uint32_t*);
template __global__ void k_expected<2 * BN_GB>(EncArgs, uint32_t*, const uint32_t*);
template __global__ void k_diag_basen<BN_GA>(const uint32_t*, int, const uint32_t*, const uint32_t*, const uint32_t*, const uint32_t*, uint32_t*, uint32_t*);
template __global__ void k_diag_basen<BN_GB>(const uint32_t*, int, const uint32_t*, const uint32_t*, const uint32_t*, const uint32_t*, uint32_t*, uint32_t*);
}  // namespace zkp
