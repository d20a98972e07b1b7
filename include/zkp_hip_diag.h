/*
 * zkp_hip_diag.h — diagnostics of libzkp_hip.so that are NOT part of the drop-in boundary.
 *
 * include/zkp_hip.h is the boundary: every entry point there replaces a call shape of the reference
 * (ZenGo-X/zk-paillier) and is what a Rust / C caller binds.  What is declared here serves this repo's own
 * measurement tooling (profiles/collect_pmc.sh, bench.py --pmc-shape) and has no counterpart in the reference;
 * bindings/rust/zkp-hip-sys does not bind it.
 */
#ifndef ZKP_HIP_DIAG_H
#define ZKP_HIP_DIAG_H

#include "zkp_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

/* Moves a KNOWN number of bytes in the access pattern of the ladders' window tables — every lane of the resident
 * grid reads (mode 0) or writes (mode 1) its block of each entry of its table slot, `passes` times — so that the HBM-side PMC
 * counters behind the roofline's `traffic` figure can be calibrated (profiles/collect_pmc.sh).  out_bytes = bytes moved. */
int32_t zkp_diag_table_traffic(zkp_ctx* ctx, int32_t mode, int32_t passes, uint64_t* out_bytes);

#ifdef __cplusplus
}
#endif
#endif /* ZKP_HIP_DIAG_H */
