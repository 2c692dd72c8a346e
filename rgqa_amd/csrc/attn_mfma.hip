// MFMA attention core (bf16) — placeholder wiring; replaced by the tiled MFMA kernels.
#include "kernels.h"
int k_attn_fwd_mfma(const AttnArgs& a, hipStream_t s) { return k_attn_fwd_ref<bf16_t>(a, s); }
int k_attn_bwd_mfma(const AttnArgs& a, hipStream_t s) { return k_attn_bwd_ref<bf16_t>(a, s); }
