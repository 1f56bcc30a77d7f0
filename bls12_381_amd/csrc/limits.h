// limits.h -- the few compile-time limits that both the kernels and the host-side translation units need (no device code here).
#pragma once
#include <stdint.h>
namespace bls {
constexpr int ITEM_CAP_MAX = 4096;                  // msm.hip.h: entries per work item of the bucket accumulation; the cap is chosen per call: max(128, ~4 x mean bucket load)
constexpr int MML_MAX_K = 8;                        // pairing.hip.h: terms of one shared-accumulator Miller loop (lane-pair layout)
constexpr uint32_t PREP_NONE = 0xffffffffu;         // prep.hip.h: per-term index "not prepared, Q comes from the g2 array"
constexpr int FR_COLS_LOG_MAX = 12;                 // fr.hip.h: largest column tile (log2 elements) of the transform, = FR_COLS_LOG there
constexpr int MMLP_MAX_K = 8;                       // prep.hip.h: terms that share one pass of the prepared loop; longer segments take several passes
}  // namespace bls
