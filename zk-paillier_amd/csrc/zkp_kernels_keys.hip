// zkp_kernels_keys.hip — the throughput engine's kernels whose ladders walk PER-ITEM exponents (fixed windows), as translation units
// of their own: the machine-scheduler strategy is a per-file compiler option, and these kernels want another one than the shared-key
// kernels of zkp_api.hip (iterative-ilp there; A/B on one box, profiles/r03/ab_scheduler_strategies.txt: `max-ilp` +2.2 … +3.5 % on
// k_enc<4, false> verify / prove, the default strategy +2 % on k_ck_check<2>, both -3 % on the n = 4096 shared-key kernel).
// -DZKP_TU_KEYS_ENC: Paillier Enc under per-proof keys; -DZKP_TU_KEYS_CK: NiCorrectKeyProof's sigma^n mod n.  zkp_api.hip declares the
// same instantiations `extern template` (ZKP_SPLIT_TU) and launches them; __graft_entry__.build() compiles and links the three files.
#define ZKP_TEMPLATE_KERNELS_ONLY          /* (bigint29.hpp) */
#include "../../include/zkp_hip.h"
#include "kernels_modexp.hpp"
#include "kernels_proofs.hpp"

namespace zkp {
constexpr int GA = 72 / W, GB = 144 / W, GC = 288 / W;
#ifdef ZKP_TU_KEYS_ENC
template __global__ void k_enc<GA, false, false>(EncArgs);
template __global__ void k_enc<GB, false, false>(EncArgs);
template __global__ void k_enc<GC, false, false>(EncArgs);
#endif
#ifdef ZKP_TU_KEYS_CK
template __global__ void k_ck_check<GA, false>(CkCheckArgs);
template __global__ void k_ck_check<GB, false>(CkCheckArgs);
#endif
}  // namespace zkp
